cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
NM=${1:-2048}
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_pk$NM -- python3 $GRAFT_REPO_ROOT/bench.py --workload pk --nmesh $NM --steps 3 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_pk$NM.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_pk$NM -name "*kernel_stats.csv" | head -1); cut -c1-200 "$f" | head -30
