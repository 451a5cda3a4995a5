"""copy the judged summaries of the last `scripts/gpu_round.sh` run from gpurun_out/round (scratch) into
profiles/<round>/ (tracked): rocprofv3 --kernel-trace --stats kernel tables, the bench JSON lines, test logs and the
PMC traffic table.

PMC correction (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB, collected in
separate passes; on gfx950 FETCH_SIZE tallies a 128-B request at 64 B, so wide streaming reads are doubled.  Kernels whose
reads are 64-B segments (the 2048-point column pass of the plain three-pass FFT: 8 complex columns per row) issue 64-B
requests that are tallied exactly -- calibrated against the known byte count (raw FETCH_SIZE == 4M bytes of the
half-spectrum to 0.5 %), so that kernel takes factor 1; the 16-column passes read 128-B runs and take factor 2 (their
raw FETCH_SIZE is half the known byte count)."""
import glob
import json
import os
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = 'gpurun_out/round'
dst = os.path.join('profiles', rnd)
os.makedirs(dst, exist_ok=True)
FETCH_FACTOR = {'fft_cols<2048, 8>': 1.0}   # 64-B row segments; every other kernel reads >= 128-B runs
# the fused last pass at 2048^3 (fft_x_bin<1024, 8, ...>) also reads 64-B row segments (8 complex columns): factor 1 like
# fft_cols<2048, 8>.  Its raw FETCH_SIZE is 0.78 of the known 4M-byte read (part of the 64-B segments pair up into 128-B
# requests tallied at 64 B): the guide calls such widths uncalibrated - the entry carries `note`.
FETCH_PREFIX = {'fft_x_bin<1024, 8': (1.0, 'uncalibrated width (64-B row segments): raw FETCH_SIZE, known read = 4 B per mesh cell'),
                'fft_x_bin2<1024, 8': (1.0, 'uncalibrated width (64-B row segments): raw FETCH_SIZE, known read = 4 B per mesh cell')}

for d in ('prof_hod', 'prof_pk1024', 'prof_pk1536', 'prof_pk2048'):
    # gpurun MERGES its output into gpurun_out/: summaries of earlier calls are still there - take the newest
    found = sorted(glob.glob(os.path.join(src, d, '**', '*kernel_stats.csv'), recursive=True), key=os.path.getmtime)
    if found:
        shutil.copy(found[-1], os.path.join(dst, f'{d[5:]}_kernel_stats.csv'))
    log = os.path.join(src, d + '.log')
    if os.path.exists(log):   # the bench JSON line printed under the profiler
        lines = [ln for ln in open(log) if ln.startswith('{"metric"')]
        if lines:
            open(os.path.join(dst, f'{d[5:]}_bench_under_rocprof.json'), 'w').write(lines[-1])
for f in ('bench_default.json', 'gpu_tests.log', 'smoke.log'):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
p = os.path.join(src, 'pmc_summary.json')
if os.path.exists(p):
    raw = json.load(open(p))
    out = {}
    for wl, ks in raw.items():
        for k, e in ks.items():
            if 'FETCH_SIZE_KiB_per_launch_raw' not in e or 'WRITE_SIZE_KiB_per_launch_raw' not in e:
                continue
            fac, note = FETCH_FACTOR.get(k, 2.0), None
            for pre, (f, nt) in FETCH_PREFIX.items():
                if k.startswith(pre):
                    fac, note = f, nt
            rd = e['FETCH_SIZE_KiB_per_launch_raw'] * 1024 * fac
            wr = e['WRITE_SIZE_KiB_per_launch_raw'] * 1024
            out.setdefault(wl, {})[k] = {'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr,
                                         'hbm_bytes_per_launch': rd + wr, 'fetch_factor': fac,
                                         'launches': e.get('launches_FETCH_SIZE')}
            if note:
                out[wl][k]['note'] = note
    json.dump(out, open(os.path.join(dst, 'pmc_traffic.json'), 'w'), indent=1, sort_keys=True)
print(sorted(os.listdir(dst)))
