"""TEST infrastructure: host-staged stand-ins for `abacusutils_amd.comm.RcclComm` over torch.distributed (gloo) - the
transport of the CPU tests (world 2 / 4 with the NumPy device stand-in) and of several ranks sharing ONE GPU.  Same
methods as the product communicator; mesh-sized exchanges are copied to the host, exchanged and copied back.  Nothing in
the product package imports torch."""
import numpy as np


class GlooSlabComm:
    """the slab estimator's / slab pair counter's transport over an initialised gloo process group (or none: one rank)"""

    device = False

    def __init__(self, group=None, force_collectives=False):
        self.dist = None
        self.rank, self.world = 0, 1
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.dist = dist
                self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        except ImportError:
            pass
        self.group = group
        self.collective = self.dist is not None and (self.world > 1 or bool(force_collectives))

    def _pairwise(self, ins, outs):
        """ins[p] -> rank p, outs[p] <- rank p (host tensors; gloo has no all_to_all)"""
        reqs = []
        for peer in range(self.world):
            if peer == self.rank:
                outs[peer].copy_(ins[peer])
            else:
                if ins[peer].numel():
                    reqs.append(self.dist.isend(ins[peer].contiguous(), peer, group=self.group))
                if outs[peer].numel():
                    reqs.append(self.dist.irecv(outs[peer], peer, group=self.group))
        for q in reqs:
            q.wait()

    def ring_exchange(self, backend, buf, left_off, right_off, recv, n, recv_off=0):
        """send buf[left_off:+n] to rank-1 and buf[right_off:+n] to rank+1;
        recv[recv_off:+n] <- what rank+1 sent left, the next n <- what rank-1 sent right"""
        if not self.collective:   # the ring neighbour is this rank (periodic box): callers add the ghosts in place
            raise RuntimeError('ring_exchange needs an initialised process group')
        import torch
        backend.sync()
        left, right = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        s_l, s_r = torch.from_numpy(buf.get(left_off, n)), torch.from_numpy(buf.get(right_off, n))
        r_from_right = torch.empty(n, dtype=torch.float32)
        r_from_left = torch.empty(n, dtype=torch.float32)
        if left == self.rank:   # one rank: its own ghosts come back
            r_from_right.copy_(s_l)
            r_from_left.copy_(s_r)
        else:
            ops = [self.dist.P2POp(self.dist.isend, s_l, left, self.group),
                   self.dist.P2POp(self.dist.isend, s_r, right, self.group),
                   self.dist.P2POp(self.dist.irecv, r_from_right, right, self.group),
                   self.dist.P2POp(self.dist.irecv, r_from_left, left, self.group)]
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        recv.set(recv_off, r_from_right.numpy())
        recv.set(recv_off + n, r_from_left.numpy())

    def all_to_all(self, backend, send, recv, n_total):
        self.all_to_all_piece(backend, send, recv, n_total // self.world, 0, n_total // self.world)

    def all_to_all_piece(self, backend, send, recv, peer_stride, offset, n, overlap=False):
        if not self.collective:   # callers unpack straight from the send buffer
            raise RuntimeError('all_to_all needs an initialised process group')
        import torch
        backend.sync()
        ins = [torch.from_numpy(send.get(p * peer_stride + offset, n)) for p in range(self.world)]
        outs = [torch.empty(n, dtype=torch.float32) for _ in range(self.world)]
        self._pairwise(ins, outs)
        for p in range(self.world):
            recv.set(p * peer_stride + offset, outs[p].numpy())

    def all_to_all_piece_v(self, backend, send, recv, send_off, send_n, recv_off, recv_n, overlap=False):
        """peer blocks of different sizes (the compact transpose of slab_power.py)"""
        if not self.collective:
            raise RuntimeError('all_to_all needs an initialised process group')
        import torch
        backend.sync()
        ins = [torch.from_numpy(send.get(int(send_off[p]), int(send_n[p]))) for p in range(self.world)]
        outs = [torch.empty(int(recv_n[p]), dtype=torch.float32) for p in range(self.world)]
        self._pairwise(ins, outs)
        for p in range(self.world):
            recv.set(int(recv_off[p]), outs[p].numpy())

    def join(self):
        pass

    def transpose_chunks(self, npair):
        return 2 if (self.collective and npair % 2 == 0 and npair >= 4) else 1   # the chunked code path, in the CPU tests too

    def all_reduce_raw(self, raw, n_u64):
        """sum the raw histogram over ranks: first n_u64 entries are uint64 counts, the rest float64"""
        if not self.collective:
            return raw
        import torch
        cnt = torch.from_numpy(raw[: n_u64 * 8].view(np.int64).copy())
        val = torch.from_numpy(raw[n_u64 * 8:].view(np.float64).copy())
        self.dist.all_reduce(cnt, group=self.group)
        if val.numel():
            self.dist.all_reduce(val, group=self.group)
        out = np.empty_like(raw)
        out[: n_u64 * 8] = cnt.numpy().view(np.uint8)
        out[n_u64 * 8:] = val.numpy().view(np.uint8)
        return out

    def all_reduce_int(self, v):
        if not self.collective:
            return int(v)
        import torch
        t = torch.tensor([int(v)], dtype=torch.int64)
        self.dist.all_reduce(t, group=self.group)
        return int(t[0])

    def all_reduce_float(self, v, op='sum'):
        if not self.collective:
            return float(v)
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == 'max' else self.dist.ReduceOp.SUM, group=self.group)
        return float(t[0])

    def barrier(self):
        if self.collective:
            self.dist.barrier(group=self.group)

    def all_to_all_host(self, arrays):
        """variable-size host all-to-all of float32 arrays (particle routing): arrays[p] goes to rank p"""
        if not self.collective:
            return [arrays[0]]
        import torch
        sizes = torch.tensor([a.size for a in arrays], dtype=torch.int64)
        all_sizes = [torch.empty(self.world, dtype=torch.int64) for _ in range(self.world)]
        self.dist.all_gather(all_sizes, sizes, group=self.group)
        outs = [np.empty(int(all_sizes[p][self.rank]), dtype=np.float32) for p in range(self.world)]
        self._pairwise([torch.from_numpy(np.ascontiguousarray(a.ravel())) for a in arrays], [torch.from_numpy(o) for o in outs])
        return outs


class GlooHodTransport:
    """what hod.shard.HodComm needs from a transport, over the default gloo process group"""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def all_reduce_array(self, a):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a).copy())
        self.dist.all_reduce(t, group=self.group)
        return t.numpy()

    def all_gather_object(self, obj):
        objs = [None] * self.world
        self.dist.all_gather_object(objs, obj, group=self.group)
        return objs
