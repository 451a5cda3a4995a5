"""per-kernel mean FETCH_SIZE / WRITE_SIZE per launch from rocprofv3 --pmc passes (gpurun_out/round/pmc_*_<COUNTER>).
Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB;
FETCH_SIZE tallies 128-B requests at 64 B for wide streaming reads, so it is doubled."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = {}
for d in sorted(glob.glob(os.path.join(root, 'pmc_*'))):
    if not os.path.isdir(d):
        continue
    tag = os.path.basename(d)[4:]
    wl, counter = tag.rsplit('_', 2)[0], '_'.join(tag.rsplit('_', 2)[1:])
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        acc = defaultdict(lambda: [0.0, 0])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                k = row['Kernel_Name'].replace('(anonymous namespace)::', '').replace('abacus::', '')
                k = k[5:] if k.startswith('void ') else k
                k = k.split('(')[0]
                a = acc[k]
                a[0] += float(row['Counter_Value'])
                a[1] += 1
        for k, (s, c) in acc.items():
            e = out.setdefault(wl, {}).setdefault(k, {})
            e[counter + '_KiB_per_launch_raw'] = s / c
            e['launches_' + counter] = c
for wl in out.values():
    for k, e in wl.items():
        f = e.get('FETCH_SIZE_KiB_per_launch_raw')
        w = e.get('WRITE_SIZE_KiB_per_launch_raw')
        if f is not None and w is not None:
            e['hbm_bytes_per_launch'] = (2.0 * f + w) * 1024.0
print(json.dumps(out, indent=1, sort_keys=True))
