"""TEST infrastructure: the slab estimator's transport between THREADS of one process - W ranks of calc_power_slab on the one
GPU of the test box without W processes (the box allows six GPU processes; 8 ranks are the north-star configuration).  Same
methods as abacusutils_amd.comm.RcclComm; mesh-sized exchanges are staged through host arrays, the library serialises its
entry points.  Nothing in the product imports this."""
import threading

import numpy as np


class ThreadWorld:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world

    def comm(self, rank):
        return ThreadComm(self, rank)


class ThreadComm:
    collective, device = True, False

    def __init__(self, tw, rank):
        self.tw, self.rank, self.world = tw, rank, tw.world

    def _exchange(self, mine):
        """every rank deposits an object, all see the list"""
        self.tw.barrier.wait()
        self.tw.slots[self.rank] = mine
        self.tw.barrier.wait()
        return list(self.tw.slots)

    def ring_exchange(self, backend, buf, left_off, right_off, recv, n, recv_off=0):
        backend.sync()
        allb = self._exchange((buf.get(left_off, n), buf.get(right_off, n)))
        left, right = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        recv.set(recv_off, allb[right][0])         # what rank + 1 sent to its left
        recv.set(recv_off + n, allb[left][1])      # what rank - 1 sent to its right

    def all_to_all_piece(self, backend, send, recv, peer_stride, offset, n, overlap=False):
        backend.sync()
        allb = self._exchange([send.get(p * peer_stride + offset, n) for p in range(self.world)])
        self.floats_sent = getattr(self, 'floats_sent', 0) + int(n) * (self.world - 1)
        for p in range(self.world):
            recv.set(p * peer_stride + offset, allb[p][self.rank])

    def all_to_all_piece_v(self, backend, send, recv, send_off, send_n, recv_off, recv_n, overlap=False):
        """peer blocks of different sizes (the compact transpose): float offsets / counts per peer on either side"""
        backend.sync()
        allb = self._exchange([send.get(int(send_off[p]), int(send_n[p])) for p in range(self.world)])
        self.floats_sent = getattr(self, 'floats_sent', 0) + int(sum(int(send_n[p]) for p in range(self.world) if p != self.rank))
        for p in range(self.world):
            assert len(allb[p][self.rank]) == int(recv_n[p])
            recv.set(int(recv_off[p]), allb[p][self.rank])

    def join(self):
        pass

    def transpose_chunks(self, npair):
        return 2 if npair % 2 == 0 and npair >= 4 else 1

    def all_reduce_raw(self, raw, n_u64):
        allr = self._exchange(raw.copy())
        out = np.empty_like(raw)
        out[:n_u64 * 8] = np.sum([a[:n_u64 * 8].view(np.uint64) for a in allr], axis=0, dtype=np.uint64).view(np.uint8)
        out[n_u64 * 8:] = np.sum([a[n_u64 * 8:].view(np.float64) for a in allr], axis=0).view(np.uint8)
        return out

    def all_reduce_int(self, v):
        return int(sum(self._exchange(int(v))))

    # what hod.shard.HodComm needs from a transport
    def all_reduce_array(self, a):
        return np.sum(self._exchange(np.asarray(a).copy()), axis=0)

    def all_gather_object(self, obj):
        return self._exchange(obj)

    def all_to_all_host(self, arrays):
        allb = self._exchange([np.ascontiguousarray(a, dtype=np.float32).ravel() for a in arrays])
        return [allb[p][self.rank] for p in range(self.world)]


def run_ranks(world, fn):
    """fn(comm) on `world` threads; returns the list of results, re-raises the first exception"""
    tw = ThreadWorld(world)
    out, err = [None] * world, []

    def body(r):
        try:
            out[r] = fn(tw.comm(r))
        except BaseException as e:   # noqa: BLE001
            err.append(e)
            tw.barrier.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if err:
        real = [e for e in err if not isinstance(e, threading.BrokenBarrierError)]
        raise (real or err)[0]
    return out
