"""One process per GPU: start the rank processes of a multi-GPU run and collect their results.

Used by bench.py (`python bench.py --gpus N`) and usable for any driver script: the PARENT never touches the GPU (no
HIP call, no library load) - it only starts `world` fresh child processes with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT / ABACUS_RDZV_KEY set, waits for them with a timeout, and parses the line a rank prints as
`<TAG> <json>`.  A child that is stuck (a collective that never completes) or has crashed is ended by its exact PID
(its own session / process group); the survivors get a short grace period, then the same.  Nothing is re-executed in a
process that has initialised the GPU.

The same function serves a launcher that already started one process per rank (torchrun): pass `ranks=[RANK]` and each
of those processes starts only its own child.
"""
import json
import os
import signal
import socket
import subprocess
import tempfile
import time


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _end(proc):
    """terminate one child we started, by PID (and its process group: it was started in its own session)"""
    if proc.poll() is not None:
        return
    try:
        os.killpg(proc.pid, signal.SIGTERM)
    except (ProcessLookupError, PermissionError):
        pass
    try:
        proc.wait(timeout=5)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
        proc.wait()


def launch_ranks(argv, world, ranks=None, timeout=600.0, tag='RESULT', key=None, env=None, port=None, grace=15.0):
    """Start `argv` once per rank in `ranks` (default: all of range(world)) and wait.

    Returns {'results': {rank: parsed json of the rank's `<tag> ...` line or None}, 'returncodes': {rank: int or None},
    'stderr': {rank: tail of its stderr}, 'timed_out': bool, 'seconds': wall time}.
    key: rendezvous key shared by the ranks of this launch (abacusutils_amd.comm reads ABACUS_RDZV_KEY)."""
    ranks = list(range(world)) if ranks is None else list(ranks)
    base = dict(os.environ if env is None else env)
    base.setdefault('MASTER_ADDR', '127.0.0.1')
    if port is not None:
        base['MASTER_PORT'] = str(port)
    base.setdefault('MASTER_PORT', str(free_port()))
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL across processes needs it on this driver
    base['ABACUS_RDZV_KEY'] = key or f'{os.getpid()}_{int(time.time() * 1e3)}'
    base['WORLD_SIZE'] = str(world)
    base.setdefault('ABACUS_RDZV_T0', repr(time.time()))   # rendezvous files older than this launch are leftovers (comm.py)
    procs, files = {}, {}
    t0 = time.time()
    for r in ranks:
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        out = tempfile.TemporaryFile(mode='w+')
        err = tempfile.TemporaryFile(mode='w+')
        files[r] = (out, err)
        procs[r] = subprocess.Popen(list(argv), env=e, stdout=out, stderr=err, start_new_session=True)
    timed_out = False
    first_failure = None
    try:
        while True:
            codes = {r: p.poll() for r, p in procs.items()}
            if all(c is not None for c in codes.values()):
                break
            now = time.time()
            if first_failure is None and any(c not in (None, 0) for c in codes.values()):
                first_failure = now           # a rank died: its peers will wait for it in the next collective
            if now - t0 > timeout or (first_failure is not None and now - first_failure > grace):
                timed_out = now - t0 > timeout
                break
            time.sleep(0.05)
    finally:      # also when the caller is being terminated (an exception raised from its signal handler): no orphans
        for p in procs.values():
            _end(p)
    res = {'results': {}, 'returncodes': {}, 'stderr': {}, 'timed_out': timed_out, 'seconds': time.time() - t0}
    for r, p in procs.items():
        out, err = files[r]
        out.seek(0)
        err.seek(0)
        parsed = None
        for line in out.read().splitlines():
            if line.startswith(tag + ' '):
                try:
                    parsed = json.loads(line[len(tag) + 1:])
                except ValueError:
                    pass
        res['results'][r] = parsed
        res['returncodes'][r] = p.returncode
        res['stderr'][r] = err.read()[-1500:]
        out.close()
        err.close()
    return res


def failure_summary(res):
    """one string describing which ranks failed and why (last stderr line of each)"""
    parts = []
    if res['timed_out']:
        parts.append(f"abandoned after {res['seconds']:.0f} s")
    for r, code in sorted(res['returncodes'].items()):
        if code != 0:
            lines = [ln for ln in res['stderr'][r].strip().splitlines() if ln.strip()]
            parts.append(f"rank {r}: exit {code}: {lines[-1] if lines else 'no message'}")
    return '; '.join(parts)
