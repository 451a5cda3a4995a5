cd $GRAFT_REPO_ROOT/scripts/ubench
hipcc -O3 --offload-arch=gfx950 -o /tmp/$1 $1.hip && timeout 300 /tmp/$1
