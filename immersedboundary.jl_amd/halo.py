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
        # gloo cannot move device buffers point-to-point: stage through pinned host memory (rehearsal only;
        # the production transport is RCCL, which takes the device buffers directly)
        self.staged = self.device.type == "cuda" and dist.get_backend(group) == "gloo"

    def _buffers(self, nv):
        if nv not in self._bufs:
            if nv == 1:
                # scalar fields: one contiguous buffer per direction, peers are slices -> ONE pack and ONE
                # unpack kernel per exchange whatever the number of peers
                sq, rq = sorted(self.send_idx), sorted(self.recv_idx)
                self.send_all = torch.cat([self.send_idx[q] for q in sq]) if sq else None
                self.recv_all = torch.cat([self.recv_idx[q] for q in rq]) if rq else None
                sball = torch.empty((1, sum(self.send_idx[q].numel() for q in sq)), dtype=torch.float32, device=self.device)
                rball = torch.empty((1, sum(self.recv_idx[q].numel() for q in rq)), dtype=torch.float32, device=self.device)
                self.sb_all, self.rb_all = sball, rball
                sb, rb, o = {}, {}, 0
                for q in sq:
                    n = self.send_idx[q].numel()
                    sb[q] = sball[:, o:o + n]
                    o += n
                o = 0
                for q in rq:
                    n = self.recv_idx[q].numel()
                    rb[q] = rball[:, o:o + n]
                    o += n
            else:
                sb = {q: torch.empty((nv, i.numel()), dtype=torch.float32, device=self.device)
                      for q, i in self.send_idx.items()}
                rb = {q: torch.empty((nv, i.numel()), dtype=torch.float32, device=self.device)
                      for q, i in self.recv_idx.items()}
            hs = hr = None
            if self.staged:
                hs = {q: torch.empty(b.shape, dtype=b.dtype).pin_memory() for q, b in sb.items()}
                hr = {q: torch.empty(b.shape, dtype=b.dtype).pin_memory() for q, b in rb.items()}
            self._bufs[nv] = (sb, rb, hs, hr)
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
        sb, rb, hs, hr = self._buffers(nv)
        ops = []
        if nv == 1:
            if self.send_all is not None:
                self._pack(field, self.send_all, self.sb_all)
        else:
            for q in self.plan.peers:
                if q in self.send_idx:
                    self._pack(field, self.send_idx[q], sb[q])
        if self.staged:
            for q in hs:
                hs[q].copy_(sb[q], non_blocking=True)
            torch.cuda.current_stream().synchronize()
        src, dst = (hs, hr) if self.staged else (sb, rb)
        for q in self.plan.peers:
            if q in self.recv_idx:
                ops.append(self.dist.P2POp(self.dist.irecv, dst[q], q - 1, self.group))
            if q in self.send_idx:
                ops.append(self.dist.P2POp(self.dist.isend, src[q], q - 1, self.group))
        reqs = self.dist.batch_isend_irecv(ops) if ops else []
        return (field, rb, hr, reqs, nv)

    def finish(self, handle):
        field, rb, hr, reqs, nv = handle
        for r in reqs:
            r.wait()
        if self.staged:
            for q in hr:
                rb[q].copy_(hr[q], non_blocking=True)
        if nv == 1:
            if self.recv_all is not None:
                self._unpack(field, self.recv_all, self.rb_all)
            return
        for q in self.plan.peers:
            if q in self.recv_idx:
                self._unpack(field, self.recv_idx[q], rb[q])

    def exchange(self, field):
        self.finish(self.start(field))
        return field


def sweep_overlapped(hx, dpart, u, C, ud, comm_stream, flags=0):
    """One advection residual sweep with the skirt exchange overlapped with interior compute.

    The exchange (pack, grouped send/recv, unpack) runs on ``comm_stream``; meanwhile the compute stream
    runs both passes on the blocks that do not depend on skirt cells (``IBH_PHASE_INTERIOR``: the block table
    is ordered interior-first by the library's analysis); the remaining blocks and the face-list cells run
    after the exchange has landed (``IBH_PHASE_BOUNDARY``).  Interior blocks never read a skirt cell, so the
    unpack kernel and the interior kernels touch disjoint rows of ``u``.
    """
    import torch
    from . import backend as B
    cur = torch.cuda.current_stream()
    comm_stream.wait_stream(cur)
    with torch.cuda.stream(comm_stream):
        hx.finish(hx.start(u))
    B.residual_advection(dpart, u, C, out=ud, flags=flags | B.IBH_PHASE_INTERIOR)
    cur.wait_stream(comm_stream)
    B.residual_advection(dpart, u, C, out=ud, flags=flags | B.IBH_PHASE_BOUNDARY)
    return ud
