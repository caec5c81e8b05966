"""Skirt-cell halo exchange for one-partition-per-GPU runs.

The reference has no communication backend: every ``dom(f, args...)`` call gathers each
partition's ``domain`` rows from one global host array and scatters its ``image`` rows back
(/root/reference/src/ImmersedBoundary.jl:836-859).  With fields resident per GPU in
partition-local order that gather/scatter becomes: before a sweep, every skirt cell
(``domain`` minus ``image``) receives the current value from the rank whose ``image`` owns it.

Owner of global cell g = g // max_partition_size (contiguous ranges, :594).  For rank r and peer
q: recv list = domain_r ∩ image_q, send list = image_r ∩ domain_q, both ordered by global id so
the two sides agree without exchanging indices.  Transport = grouped point-to-point send/recv
(``torch.distributed.batch_isend_irecv``: on RCCL one ncclGroupStart/End of ncclSend/ncclRecv,
each pair on its own xGMI link); pack/unpack are row gather/scatter kernels of libibhip.
CPU tensors are accepted for the transport only (gloo tests): pack/unpack there is plain
indexing, no arithmetic.
"""
from __future__ import annotations

import numpy as np
import torch


class HaloPlan:
    """Send/recv lists of partition ``pid`` (1-based) against every other partition of ``dom``."""

    def __init__(self, dom, pid):
        part = dom.partitions[pid]
        self.pid = pid
        self.nc = part.spacing.shape[0]
        lo, hi = dom.images[pid]
        g2l = {}
        domain = np.asarray(part.domain)
        self.send, self.recv = {}, {}
        for q, (qlo, qhi) in dom.images.items():
            if q == pid:
                continue
            # cells of my domain owned by q (sorted by global id since domain is sorted)
            sel = np.nonzero((domain >= qlo) & (domain < qhi))[0]
            if sel.size:
                self.recv[q] = sel.astype(np.int32)
            dq = np.asarray(dom.domains[q])
            mine = dq[(dq >= lo) & (dq < hi)]
            if mine.size:
                self.send[q] = np.searchsorted(domain, mine).astype(np.int32)
        self.peers = sorted(set(self.send) | set(self.recv))
        self.n_send = sum(v.size for v in self.send.values())
        self.n_recv = sum(v.size for v in self.recv.values())


class HaloExchange:
    """Executes a HaloPlan over a torch.distributed process group (rank = pid - 1)."""

    def __init__(self, plan: HaloPlan, device, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.plan = plan
        self.group = group
        self.device = torch.device(device)
        self.send_idx = {q: torch.from_numpy(v).to(self.device) for q, v in plan.send.items()}
        self.recv_idx = {q: torch.from_numpy(v).to(self.device) for q, v in plan.recv.items()}
        self._bufs = {}

    def _buffers(self, nv):
        if nv not in self._bufs:
            sb = {q: torch.empty((nv, i.numel()), dtype=torch.float32, device=self.device)
                  for q, i in self.send_idx.items()}
            rb = {q: torch.empty((nv, i.numel()), dtype=torch.float32, device=self.device)
                  for q, i in self.recv_idx.items()}
            self._bufs[nv] = (sb, rb)
        return self._bufs[nv]

    def _pack(self, field, idx, buf):
        if field.is_cuda:
            from . import backend as B
            f, nv, ld = B._field(field)
            B._stream()
            B.call("ibh_gather_rows", B._ptr(idx), idx.numel(), B._ptr(f), nv, ld, B._ptr(buf), idx.numel())
        else:
            src = field if field.ndim == 2 else field[:, None]
            buf.copy_(src[idx.long()].T)

    def _unpack(self, field, idx, buf):
        if field.is_cuda:
            from . import backend as B
            f, nv, ld = B._field(field)
            B._stream()
            B.call("ibh_scatter_rows", B._ptr(idx), idx.numel(), B._ptr(buf), nv, idx.numel(), B._ptr(f), ld)
        else:
            dst = field if field.ndim == 2 else field[:, None]
            dst[idx.long()] = buf.T

    def start(self, field):
        """Pack and post the sends/receives of ``field`` (local ``(nc,)`` or ``(nc, nv)``); returns a handle."""
        nv = 1 if field.ndim == 1 else field.shape[1]
        sb, rb = self._buffers(nv)
        ops = []
        for q in self.plan.peers:
            if q in self.send_idx:
                self._pack(field, self.send_idx[q], sb[q])
        for q in self.plan.peers:
            if q in self.recv_idx:
                ops.append(self.dist.P2POp(self.dist.irecv, rb[q], q - 1, self.group))
            if q in self.send_idx:
                ops.append(self.dist.P2POp(self.dist.isend, sb[q], q - 1, self.group))
        reqs = self.dist.batch_isend_irecv(ops) if ops else []
        return (field, rb, reqs)

    def finish(self, handle):
        field, rb, reqs = handle
        for r in reqs:
            r.wait()
        for q in self.plan.peers:
            if q in self.recv_idx:
                self._unpack(field, self.recv_idx[q], rb[q])

    def exchange(self, field):
        self.finish(self.start(field))
        return field
