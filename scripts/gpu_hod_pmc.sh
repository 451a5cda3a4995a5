#!/bin/bash
# PMC passes over the kernels of one HOD workload (separate passes, kernel-trace only): gpu_hod_pmc.sh [multi|c2]
cd "$GRAFT_REPO_ROOT" || exit 1
W=${1:-multi}
O=$GRAFT_REPO_ROOT/gpurun_out/hod_pmc
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR"; do
  i=$((i + 1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$O/p$i" -- python3 "$GRAFT_REPO_ROOT/scripts/hod_probe.py" "$W" 3 > "$O/p$i.log" 2>&1 || { tail -5 "$O/p$i.log"; }
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for d in sorted(glob.glob(O + '/p*/')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        rows = list(csv.DictReader(open(f)))
        # per dispatch: kernel name, counter -> value; keep the hod_* kernels of the LAST populate (warm)
        by = collections.OrderedDict()
        for r in rows:
            k = r['Kernel_Name']
            if 'hod_filter' in k or 'hod_exact' in k or 'hod_emit' in k:
                by.setdefault((int(r['Dispatch_Id']), k.split('(')[0][-40:]), {})[r['Counter_Name']] = float(r['Counter_Value'])
        keys = list(by)[-4:]
        for key in keys:
            print(key[1], {c: ('%.4g' % v) for c, v in by[key].items()})
PY
find "$O" \( -name "*kernel_trace.csv" -o -name "*.db" \) -delete
