"""Multi-GPU communicator: RCCL through the C ABI (`abacus_comm_*`, csrc/comm.hip), one process per GPU.

No Python framework in between: the 128-byte RCCL id travels through a FILE rendezvous (rank 0 writes it atomically,
the other ranks poll), every rank then calls `ncclCommInitRank` inside the library on the device it was bound to.
Under a launcher that exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT (`python -m torch.distributed.run`, bench.py's
own launcher, or anything else) `RcclComm.from_env()` is all a rank has to call.

The reference has no distributed layer (its unit of decomposition is the slab chunk of one process,
abacusnbody/hod/abacus_hod.py:301-312); this is the transport of the slab P(k) (analysis/slab_power.py), the sharded HOD
(hod/shard.py) and the slab pair counts (analysis/slab_pairs.py).  The CPU tests use a host-staged stand-in with the same
methods (tests/gloo_comm.py, torch.distributed gloo) - test infrastructure, not part of this package.
"""
import ctypes as C
import os
import tempfile
import time

import numpy as np

from . import _lib

ID_BYTES = 128
_DT = {np.dtype(np.int64): 0, np.dtype(np.float64): 1, np.dtype(np.float32): 2, np.dtype(np.uint64): 3}
_seq = 0   # communicators created by this process (every rank creates them in the same order)


def _rendezvous_path(key=None):
    global _seq
    if key is None:
        key = os.environ.get('ABACUS_RDZV_KEY')
    if key is None:   # ranks of one launch share their parent (the launcher), the rendezvous port and the restart attempt
        key = (f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_"
               f"{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}_{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}")
    d = os.environ.get('ABACUS_RDZV_DIR', tempfile.gettempdir())
    path = os.path.join(d, f'abacus_rdzv_{key}_{_seq}')
    _seq += 1
    return path


def _launch_time():
    """when this launch began, if the launcher said so (abacusutils_amd/launch.py exports ABACUS_RDZV_T0): a rendezvous file
    older than that belongs to an earlier launch with the same key"""
    try:
        return float(os.environ['ABACUS_RDZV_T0'])
    except (KeyError, ValueError):
        return None


def exchange_id(rank, world, make_id, path, timeout=180.0):
    """rank 0 publishes `make_id()` (bytes) at `path`; the others wait for it.  Rank 0 first removes whatever an earlier
    launch left at `path`, writes under a temporary name and renames, so a reader never sees a stale or partial id; it
    removes the file again once every rank holds a communicator."""
    if rank == 0:
        try:
            os.unlink(path)
        except OSError:
            pass
        blob = make_id()
        tmp = f'{path}.tmp{os.getpid()}'
        with open(tmp, 'wb') as f:
            f.write(blob)
        os.replace(tmp, path)
        return blob
    t0 = time.time()
    launched = _launch_time()
    # rank 0 may have written before this rank started, so "older than me" is no criterion; "older than the launch" is
    # (when the launcher exported its start time), else anything older than ten minutes is a leftover
    oldest = launched - 30.0 if launched is not None else t0 - 600.0
    while True:
        try:
            if os.path.getmtime(path) >= oldest:
                with open(path, 'rb') as f:
                    blob = f.read()
                if len(blob) == ID_BYTES:
                    return blob
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError(f'rank {rank}: no RCCL id at {path} after {timeout:.0f} s (is rank 0 running?)')
        time.sleep(0.02)


class RcclJoinTimeout(TimeoutError):
    """ncclCommInitRank / the first barrier did not return: the helper thread is still inside the call (the first barrier
    holds the library's API lock while it waits), so this process cannot be trusted with further library calls - it must
    end and let the launcher start a fresh one.  `Dist.from_env` never turns this into a file-barrier fallback."""


def _run_with_deadline(fn, timeout, what):
    """fn() in a daemon thread; its exception is re-raised here, RcclJoinTimeout(what) if it is still running after `timeout` s"""
    import threading
    box = {}

    def body():
        try:
            fn()
        except BaseException as e:   # noqa: BLE001 - handed to the caller
            box['error'] = e

    t = threading.Thread(target=body, daemon=True)
    t.start()
    t.join(timeout)
    if t.is_alive():
        raise RcclJoinTimeout(what)
    if 'error' in box:
        raise box['error']


class RcclComm:
    """ncclComm of `world` ranks on the library's device and stream.

    Device-buffer collectives are enqueued on the library stream (no host synchronisation); the small host-side
    messages (histograms, counts, timings) are staged through device scratch by the library.
    Method names are those of the slab estimator's transport (analysis/slab_power.py)."""

    device = True   # mesh-sized exchanges stay on the device

    def __init__(self, rank, world, key=None, timeout=180.0):
        self.rank, self.world = int(rank), int(world)
        self._h = C.c_void_p()
        L = _lib.lib()

        def make_id():
            buf = C.create_string_buffer(ID_BYTES)
            _lib.check(L.abacus_comm_unique_id(buf, ID_BYTES))
            return buf.raw

        path = _rendezvous_path(key)
        blob = exchange_id(self.rank, self.world, make_id, path, timeout)
        # ncclCommInitRank and the first barrier block until every rank has joined and have no timeout of their own: a rank
        # that died between the rendezvous and here would hang the others for good.  They run in a helper thread (ctypes
        # releases the GIL) and the caller gives up after `timeout` - the stuck call still holds the library lock, so
        # the only sound reaction to this error is to end the process (abacusutils_amd/launch.py does, by PID).
        def join_ranks():
            _lib.check(L.abacus_comm_init(self.rank, self.world, blob, ID_BYTES, C.byref(self._h)))
            self.collective = True      # the collectives run (and are exercised) for a single rank as well
            self.barrier()

        _run_with_deadline(join_ranks, timeout,
                           f'rank {self.rank}/{self.world}: ncclCommInitRank / first barrier did not return within {timeout:.0f} s '
                           '(a peer rank is missing or the links are down); this process must exit')
        if self.rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass

    @classmethod
    def from_env(cls, key=None, timeout=180.0):
        """RANK / WORLD_SIZE / LOCAL_RANK of the launcher; binds this process to GPU LOCAL_RANK before the first HIP call"""
        rank = int(os.environ.get('RANK', '0'))
        world = int(os.environ.get('WORLD_SIZE', '1'))
        local = int(os.environ.get('LOCAL_RANK', str(rank)))
        ndev = _lib.device_count()
        if ndev < 1:
            raise _lib.AbacusHipError(f'rank {rank}: no HIP device available (libabacus_hip.so has no CPU fallback)')
        _lib.set_device(local % ndev)
        return cls(rank, world, key=key, timeout=timeout)

    # ---- bookkeeping -------------------------------------------------------------------------------------
    def info(self):
        r, w, v, b = C.c_int(0), C.c_int(0), C.c_int(0), C.c_uint64(0)
        _lib.check(_lib.lib().abacus_comm_info(self._h, C.byref(r), C.byref(w), C.byref(v), C.byref(b)))
        return dict(rank=r.value, world=w.value, rccl_version=v.value, bytes_sent=b.value)

    def free(self):
        if self._h:
            _lib.lib().abacus_comm_free(self._h)
            self._h = C.c_void_p()

    def abort(self):
        if self._h:
            _lib.lib().abacus_comm_abort(self._h)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    # ---- host-side small messages ------------------------------------------------------------------------
    def barrier(self):
        _lib.check(_lib.lib().abacus_comm_barrier(self._h))

    def all_reduce_array(self, a, op='sum'):
        """in-place all-reduce of a host int64 / uint64 / float64 / float32 array; returns it"""
        a = np.ascontiguousarray(a)
        _lib.check(_lib.lib().abacus_comm_allreduce_host(self._h, _lib.ptr(a), C.c_int64(a.size), _DT[a.dtype],
                                                         {'sum': 0, 'max': 1}[op]))
        return a

    def all_reduce_int(self, v):
        return int(self.all_reduce_array(np.array([int(v)], dtype=np.int64))[0])

    def all_reduce_float(self, v, op='sum'):
        return float(self.all_reduce_array(np.array([float(v)], dtype=np.float64), op)[0])

    def all_gather_array(self, a):
        """[world, ...] stack of every rank's (equal-shape) host array"""
        a = np.ascontiguousarray(a)
        out = np.empty((self.world,) + a.shape, dtype=a.dtype)
        _lib.check(_lib.lib().abacus_comm_allgather_host(self._h, _lib.ptr(a), _lib.ptr(out), C.c_uint64(a.nbytes)))
        return out

    def all_reduce_raw(self, raw, n_u64):
        """sum the raw (k, mu) histogram over ranks: the first n_u64 entries are uint64 counts, the rest float64"""
        out = np.array(raw, dtype=np.uint8, copy=True)
        cnt = out[: n_u64 * 8].view(np.uint64)
        val = out[n_u64 * 8:].view(np.float64)
        if cnt.size:
            cnt[:] = self.all_reduce_array(cnt.copy())
        if val.size:
            val[:] = self.all_reduce_array(val.copy())
        return out

    # ---- mesh-sized exchanges (device buffers of analysis.slab_power.HipBuf; float32 element counts) -----
    def ring_exchange(self, backend, buf, left_off, right_off, recv, n, recv_off=0):
        """buf[left_off:+n] -> rank-1, buf[right_off:+n] -> rank+1; recv[recv_off:+n] <- from rank+1, the next n <- from rank-1"""
        _lib.check(_lib.lib().abacus_comm_ring_exchange(self._h, buf.ptr(left_off), buf.ptr(right_off), recv.ptr(recv_off),
                                                        recv.ptr(recv_off + n), C.c_uint64(4 * int(n))))

    def all_to_all(self, backend, send, recv, n_total):
        _lib.check(_lib.lib().abacus_comm_all_to_all(self._h, send.ptr(0), recv.ptr(0),
                                                     C.c_uint64(4 * (int(n_total) // self.world)), 0))

    def all_to_all_piece(self, backend, send, recv, peer_stride, offset, n, overlap=True):
        """the piece [offset, offset + n) of every peer block (blocks `peer_stride` floats apart); `overlap`: on the
        communicator's stream, behind the kernels enqueued so far - `join()` before the received data is used"""
        _lib.check(_lib.lib().abacus_comm_all_to_all_strided(self._h, send.ptr(0), recv.ptr(0), C.c_uint64(4 * int(peer_stride)),
                                                             C.c_uint64(4 * int(offset)), C.c_uint64(4 * int(n)),
                                                             int(bool(overlap))))

    def all_to_all_piece_v(self, backend, send, recv, send_off, send_n, recv_off, recv_n, overlap=True):
        """a piece of a transpose whose peer blocks differ in size (the compact transpose of slab_power.py): per peer the
        float offset / count inside the send and the receive buffer; one grouped send / recv, on the communicator's stream
        behind an event fork when `overlap`"""
        a = [np.ascontiguousarray(4 * np.asarray(v, dtype=np.int64)).astype(np.uint64) for v in (send_n, send_off, recv_n, recv_off)]
        _lib.check(_lib.lib().abacus_comm_all_to_all_v_async(self._h, send.ptr(0), _lib.ptr(a[0]), _lib.ptr(a[1]), recv.ptr(0),
                                                             _lib.ptr(a[2]), _lib.ptr(a[3]), int(bool(overlap))))

    def join(self):
        _lib.check(_lib.lib().abacus_comm_join(self._h))

    def all_to_all_v_dev(self, send_ptr, send_bytes, send_off, recv_ptr, recv_bytes, recv_off):
        u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint64)   # noqa: E731
        sb, so, rb, ro = u8(send_bytes), u8(send_off), u8(recv_bytes), u8(recv_off)
        _lib.check(_lib.lib().abacus_comm_all_to_all_v(self._h, send_ptr, _lib.ptr(sb), _lib.ptr(so), recv_ptr, _lib.ptr(rb),
                                                       _lib.ptr(ro)))

    def all_to_all_host(self, arrays):
        """variable-size all-to-all of host float32 arrays (ghost points of the slab pair counts): arrays[p] goes to
        rank p; returns the W flat arrays received.  Staged through HBM, one grouped send/recv."""
        flat = [np.ascontiguousarray(a, dtype=np.float32).ravel() for a in arrays]
        counts = np.array([a.size for a in flat], dtype=np.int64)
        rcounts = self.all_gather_array(counts)[:, self.rank].copy()
        soff = np.concatenate(([0], np.cumsum(counts)[:-1]))
        roff = np.concatenate(([0], np.cumsum(rcounts)[:-1]))
        send = np.concatenate(flat) if counts.sum() else np.zeros(1, dtype=np.float32)
        dsend = _lib.DeviceArray(send)
        nrecv = int(rcounts.sum())
        drecv = _lib.DeviceArray(nbytes=max(nrecv, 1) * 4, dtype=np.float32, shape=(max(nrecv, 1),))
        self.all_to_all_v_dev(dsend.ptr, counts * 4, soff * 4, drecv.ptr, rcounts * 4, roff * 4)
        got = drecv.get()       # synchronises
        dsend.free()
        drecv.free()
        return [got[roff[p]:roff[p] + rcounts[p]].copy() for p in range(self.world)]

    def all_gather_object(self, obj):
        """every rank's picklable object, in rank order (control plane: merged mock catalogues, result dicts)"""
        import pickle
        blob = np.frombuffer(pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL), dtype=np.uint8)
        sizes = self.all_gather_array(np.array([blob.size], dtype=np.int64))[:, 0]
        pad = np.zeros(int(sizes.max()), dtype=np.uint8)
        pad[:blob.size] = blob
        allb = self.all_gather_array(pad)
        return [pickle.loads(allb[r, :sizes[r]].tobytes()) for r in range(self.world)]

    def transpose_chunks(self, npair):
        """pieces the pencil transpose is cut into so that the links work while the next plane pairs are transformed"""
        for c in (4, 2):
            if self.world > 1 and npair % c == 0 and npair // c >= 8:
                return c
        return 1

    # ---- particle routing on the device ------------------------------------------------------------------
    def route_particles(self, dpos, dw, Lbox, fold=False):
        """every particle to the rank that owns its x-slab (fold: its folded slab pair, analysis/slab_power.py), without
        leaving HBM: stable bucket sort by owner (abacus_slab_route_dev), counts all-gathered, ONE grouped send/recv of the
        variable blocks.
        dpos: DeviceArray (n, 3) float32; dw: DeviceArray (n,) float32 or None.  Returns new DeviceArrays."""
        L = _lib.lib()
        W = self.world
        n = dpos.shape[0]
        spos = _lib.DeviceArray(nbytes=max(n, 1) * 12, dtype=np.float32, shape=(n, 3))
        sw = None if dw is None else _lib.DeviceArray(nbytes=max(n, 1) * 4, dtype=np.float32, shape=(n,))
        counts = np.zeros(W, dtype=np.int64)
        _lib.check(L.abacus_slab_route_dev(dpos.ptr, C.c_int64(n), None if dw is None else dw.ptr, C.c_double(Lbox), W,
                                           int(bool(fold)), spos.ptr, None if sw is None else sw.ptr, _lib.ptr(counts)))
        allc = self.all_gather_array(counts)            # allc[r, p] = particles rank r holds for rank p
        rcounts = allc[:, self.rank].copy()
        nrecv = int(rcounts.sum())
        soff = np.concatenate(([0], np.cumsum(counts)[:-1]))
        roff = np.concatenate(([0], np.cumsum(rcounts)[:-1]))
        rpos = _lib.DeviceArray(nbytes=max(nrecv, 1) * 12, dtype=np.float32, shape=(nrecv, 3))
        self.all_to_all_v_dev(spos.ptr, counts * 12, soff * 12, rpos.ptr, rcounts * 12, roff * 12)
        rw = None
        if dw is not None:
            rw = _lib.DeviceArray(nbytes=max(nrecv, 1) * 4, dtype=np.float32, shape=(nrecv,))
            self.all_to_all_v_dev(sw.ptr, counts * 4, soff * 4, rw.ptr, rcounts * 4, roff * 4)
        _lib.sync()     # the sorted copies are freed here
        spos.free()
        if sw is not None:
            sw.free()
        return rpos, rw


class LocalComm:
    """one process, no transport: what the slab estimator / slab pair counter / bench use when WORLD_SIZE is 1.  The
    periodic neighbours of the only slab are the slab itself - callers handle `collective == False` in place."""

    device = False
    collective = False
    rank, world = 0, 1

    def barrier(self):
        pass

    def join(self):
        pass

    def transpose_chunks(self, nxl):
        return 1

    def all_reduce_raw(self, raw, n_u64):
        return raw

    def all_reduce_int(self, v):
        return int(v)

    def all_reduce_float(self, v, op='sum'):
        return float(v)

    def all_to_all_host(self, arrays):
        return [arrays[0]]

    def info(self):
        return dict(rank=0, world=1, transport='none (single process)')

    def free(self):
        pass


_default = None


def default_comm():
    """the communicator of this process: an RcclComm from the launcher's environment when WORLD_SIZE > 1 (created once,
    reused by every later call), else a LocalComm"""
    global _default
    if _default is None:
        _default = RcclComm.from_env() if int(os.environ.get('WORLD_SIZE', '1')) > 1 else LocalComm()
    return _default


class FileComm:
    """barrier and scalar all-reduce through files of the rendezvous directory: what a leg WITHOUT a data-path collective
    (the sharded HOD: every rank populates its own shard) falls back to when the RCCL communicator cannot be created -
    its throughput does not depend on the transport, only the start barrier and the max over the ranks' timings do.
    Each call is one round: every rank writes `<key>.<round>.<rank>`, polls for the others, and reads their values.
    File lifetime: a rank that has completed round r knows every rank has written round r, hence finished READING round
    r - 1 - so it removes its own files up to r - 1 and nothing later.  `free()` is an acknowledged last round: every rank
    writes a `done` marker; rank 0 waits for all markers and removes whatever is left (a peer still polling the last
    round therefore always finds its files)."""

    device = False

    def __init__(self, rank, world, key=None, timeout=180.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self._base = _rendezvous_path(key) + '.file'
        self._round = 0

    def _name(self, rnd, rank):
        return f'{self._base}.{rnd}.{rank}'

    @staticmethod
    def _publish(path, text):
        tmp = path + '.tmp'
        with open(tmp, 'w') as f:
            f.write(text)
        os.replace(tmp, path)       # atomic: a reader sees the whole value or no file

    def _read(self, path, what, t0):
        while True:
            try:
                with open(path) as f:
                    return f.read()
            except OSError:
                pass
            if time.time() - t0 > self.timeout:
                raise TimeoutError(f'rank {self.rank}: {what}')
            time.sleep(0.0005)

    def _exchange(self, value):
        self._round += 1
        self._publish(self._name(self._round, self.rank), repr(float(value)))
        t0 = time.time()
        vals = [float(self._read(self._name(self._round, r),
                                 f'rank {r} did not reach round {self._round} of the file barrier', t0))
                for r in range(self.world)]
        if self._round > 1:          # every rank has left round - 1 behind (see the class comment)
            try:
                os.unlink(self._name(self._round - 1, self.rank))
            except OSError:
                pass
        return vals

    def barrier(self):
        self._exchange(0.0)

    def all_reduce_float(self, x, op='sum'):
        vals = self._exchange(x)
        return max(vals) if op == 'max' else sum(vals)

    def info(self):
        return dict(rank=self.rank, world=self.world, transport='file barrier (no RCCL communicator)')

    def free(self):
        import glob
        if self._round < 0:
            return
        self._publish(f'{self._base}.done.{self.rank}', str(self._round))
        if self.rank == 0:           # the last one out removes the files: every marker means "I read my last round"
            t0 = time.time()
            try:
                for r in range(self.world):
                    self._read(f'{self._base}.done.{r}', f'rank {r} did not finish (file barrier clean-up)', t0)
            except TimeoutError:
                pass                 # a crashed peer: clean up anyway
            for f in glob.glob(f'{self._base}.*'):
                try:
                    os.unlink(f)
                except OSError:
                    pass
        self._round = -1


class Dist:
    """what bench.py and the sharded drivers need from a process group: rank / world, barrier, max / sum of a scalar.
    world == 1: no communicator at all (no RCCL, no rendezvous)."""

    rccl_error = None

    def __init__(self, comm=None):
        self.comm = comm
        self.rank = comm.rank if comm else 0
        self.world = comm.world if comm else 1
        self.local_rank = int(os.environ.get('LOCAL_RANK', str(self.rank)))

    @classmethod
    def from_env(cls, allow_file_fallback=False, **kw):
        """`allow_file_fallback`: a leg without a data-path collective keeps going on a FileComm when RCCL cannot be
        initialised (the error is kept in `self.rccl_error` and reported by the caller)"""
        if int(os.environ.get('WORLD_SIZE', '1')) > 1:
            try:
                return cls(RcclComm.from_env(**kw))
            except RcclJoinTimeout:  # a join that never returned: a thread of this process is still inside the library
                raise
            except Exception as e:   # noqa: BLE001 - whatever RCCL / the rendezvous raised
                if not allow_file_fallback:
                    raise
                d = cls(FileComm(int(os.environ.get('RANK', '0')), int(os.environ['WORLD_SIZE']), key=kw.get('key')))
                d.rccl_error = f'{type(e).__name__}: {e}'
                return d
        return cls(None)

    def barrier(self):
        if self.comm:
            self.comm.barrier()

    def max(self, x):
        return self.comm.all_reduce_float(x, 'max') if self.comm else x

    def sum(self, x):
        return self.comm.all_reduce_float(x, 'sum') if self.comm else x

    def finish(self):
        if self.comm:
            self.comm.free()
            self.comm = None
