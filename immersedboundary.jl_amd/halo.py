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

    def __init__(self, dom, pid, extra=None):
        """``extra``: table ``{partition: sorted global ids}`` of cells each partition needs beyond its ``domain`` --
        the ``impose_bc!`` donor cells of distributed.bc_donor_extras.  They live in rows
        ``nc .. nc + len(extra[pid])`` of the extended local arrays and are received from their owners with the skirt
        cells in the same exchange."""
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
        self.n_extra = 0
        if extra is not None:
            mine = np.asarray(extra[pid], dtype=np.int64)
            self.n_extra = int(mine.size)
            for q, (qlo, qhi) in dom.images.items():
                if q == pid:
                    continue
                sel = np.nonzero((mine >= qlo) & (mine < qhi))[0]          # my extras owned by q
                if sel.size:
                    add = (self.nc + sel).astype(np.int32)
                    self.recv[q] = np.concatenate([self.recv[q], add]) if q in self.recv else add
                want = np.asarray(extra[q], dtype=np.int64)
                want = want[(want >= lo) & (want < hi)]                    # q's extras owned by me
                if want.size:
                    add = np.searchsorted(domain, want).astype(np.int32)
                    self.send[q] = np.concatenate([self.send[q], add]) if q in self.send else add
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
            f, nv, ld = B._field_inplace(field, what="halo field")
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
    from . import backend as B
    return overlapped(hx, u, comm_stream,
                      lambda ph: B.residual_advection(dpart, u, C, out=ud, flags=flags | ph))


def euler_sweep_overlapped(hx, dpart, P, R, comm_stream, flags=0, fluid=None):
    """The same for the Euler sweep (``hx`` built for nv = nd + 2)."""
    from . import backend as B
    return overlapped(hx, P, comm_stream,
                      lambda ph: B.residual_euler_hll(dpart, P, out=R, flags=flags | ph, fluid=fluid))


def overlapped(hx, field, comm_stream, sweep_phase):
    """Exchange ``field`` on ``comm_stream`` while ``sweep_phase(IBH_PHASE_INTERIOR)`` runs, then
    ``sweep_phase(IBH_PHASE_BOUNDARY)``."""
    import torch
    from . import backend as B
    cur = torch.cuda.current_stream()
    comm_stream.wait_stream(cur)
    with torch.cuda.stream(comm_stream):
        hx.finish(hx.start(field))
    out = sweep_phase(B.IBH_PHASE_INTERIOR)
    cur.wait_stream(comm_stream)
    sweep_phase(B.IBH_PHASE_BOUNDARY)
    return out


class XgmiHalo:
    """Direct peer-write halo exchange over xGMI (same start/finish/exchange interface as HaloExchange).

    Every rank owns a fine-grained receive buffer (two parities) and a flag word per peer, exported through
    HIP IPC and mapped by the peers.  The exchange is ONE kernel (``ibh_halo_exchange``): it stores the skirt values
    of all peers straight into their buffers, its last workgroup to finish bumps the sequence number in the peers'
    flag words, then every workgroup waits (bounded spin) for the local flag words and unpacks its share.  (A
    strong-scaled step is a handful of ~3 us launches, so every launch counts; ``ibh_halo_push`` / ``ibh_halo_pull``
    are the same two halves as separate launches.)  No host
    synchronisation and no library call besides kernel launches, so a whole sweep (exchange included) can be
    captured in a HIP graph.  Double buffering makes the protocol race-free: a rank can only overwrite
    parity p of a peer after that peer signalled step n+1, which it does after unpacking step n.
    ``healthy()`` reports spin time-outs; callers verify against the RCCL path once and fall back.
    """

    def __init__(self, plan: HaloPlan, dom, device, group=None, nv=1, max_spins=4_000_000):
        import ctypes as C
        import torch.distributed as dist
        from . import backend as B
        self.B, self.C, self.dist, self.group = B, C, dist, group
        self.plan, self.nv, self.max_spins = plan, nv, int(max_spins)
        self.device = torch.device(device)
        self.world = dist.get_world_size(group)
        self.me = plan.pid - 1
        self.peers_recv, self.peers_send = sorted(plan.recv), sorted(plan.send)
        self._recv = self._flags = None
        self._opened = []
        # Set-up is collective: EVERY step that can fail on one rank alone (no device, allocation, IPC export / import,
        # device tables) runs under `_agree`, which gathers the outcome of all ranks -- either every rank gets an
        # exchanger or all of them raise the same error here; nobody is left waiting in a collective of a later step.
        self._agree(self._setup_device, "device set-up")
        # The double-buffer argument (class docstring) needs every peer I write to to be a peer I wait on: only then
        # am I at most one step ahead of it.  Skirt dependencies are symmetric for face-connected partitions; a
        # one-sided plan (e.g. donor-extended lists) is rejected here, collectively, instead of racing silently.
        sym = self._agree(lambda: self.peers_recv == self.peers_send, "peer lists", gather=True)
        if not all(sym):
            raise ValueError("XgmiHalo needs symmetric peers (every send peer is also a receive peer) on every rank")
        self.recv_off, o = {}, 0
        for q in self.peers_recv:
            self.recv_off[q] = o
            o += nv * int(plan.recv[q].size)
        self.n_recv_f = o
        every = self._agree(self._ipc_alloc_export, "allocation / IPC export", gather=True)
        self.remote = {}       # q -> (recv base address, floats per parity, my offset, flag slot address)
        self._agree(lambda: self._ipc_import(every), "mapping a peer buffer")
        self._agree(self._device_tables, "device tables")
        self.step = 0
        dist.barrier(group=group)

    def _agree(self, fn, what, gather=False):
        """Run ``fn`` on this rank, then gather (value, error) of every rank: raise on ALL ranks if any failed.  Returns
        this rank's value, or the list of every rank's value (``gather``)."""
        val, err = None, None
        try:
            val = fn()
        except Exception as e:  # noqa: BLE001 -- any local failure must become everybody's failure
            err = f"rank {self.me}: {e!r}"
        every = [None] * self.world
        self.dist.all_gather_object(every, (val if gather else None, err), group=self.group)
        errs = [x[1] for x in every if x[1]]
        if errs:
            self._release_local()
            raise RuntimeError(f"XgmiHalo: {what} failed on a rank: " + "; ".join(errs))
        return [x[0] for x in every] if gather else val

    def _release_local(self):
        """Free what this rank has allocated or mapped so far (no collective inside)."""
        for p in self._opened:
            try:
                self.B.call("ibh_ipc_close", p)
            except Exception:  # noqa: BLE001
                pass
        self._opened = []
        for name in ("_recv", "_flags"):
            p = getattr(self, name, None)
            if p is not None and getattr(p, "value", None):
                try:
                    self.B.call("ibh_ipc_free", p)
                except Exception:  # noqa: BLE001
                    pass
            setattr(self, name, None)

    # ---- the steps of the set-up that touch the device (overridden by the CPU tests of the agreement logic)
    def _setup_device(self):
        self.B._dev()

    def _ipc_alloc_export(self):
        B, C = self.B, self.C
        self._recv, self._flags = B.c_vp(), B.c_vp()
        B.call("ibh_ipc_alloc", C.byref(self._recv), 2 * max(self.n_recv_f, 1) * 4, 1)
        B.call("ibh_ipc_alloc", C.byref(self._flags), self.world * 4, 1)
        hr, hf = (C.c_ubyte * 64)(), (C.c_ubyte * 64)()
        B.call("ibh_ipc_export", self._recv, C.cast(hr, B.c_vp))
        B.call("ibh_ipc_export", self._flags, C.cast(hf, B.c_vp))
        return dict(recv=bytes(hr), flags=bytes(hf), off={int(q): int(v) for q, v in self.recv_off.items()},
                    n=int(self.n_recv_f))

    def _ipc_import(self, every):
        B, C = self.B, self.C
        for q in self.peers_send:
            info = every[q - 1]
            pr, pf = B.c_vp(), B.c_vp()
            B.call("ibh_ipc_import", C.cast(C.create_string_buffer(info["recv"], 64), B.c_vp), C.byref(pr))
            self._opened.append(pr)
            B.call("ibh_ipc_import", C.cast(C.create_string_buffer(info["flags"], 64), B.c_vp), C.byref(pf))
            self._opened.append(pf)
            self.remote[q] = (pr.value, info["n"], info["off"][self.plan.pid], pf.value + 4 * self.me)

    def _device_tables(self):
        C, plan = self.C, self.plan
        self.send_idx = {q: torch.from_numpy(plan.send[q]).to(self.device) for q in self.peers_send}
        self.recv_idx = {q: torch.from_numpy(plan.recv[q]).to(self.device) for q in self.peers_recv}
        self.recv_all = torch.cat([self.recv_idx[q] for q in self.peers_recv]) if self.peers_recv else None
        self.send_all = torch.cat([self.send_idx[q] for q in self.peers_send]) if self.peers_send else None
        self.state = torch.zeros(8, dtype=torch.int32, device=self.device)  # signal seq, wait seq, status, done x2
        # host-side launch tables of the two-kernel exchange (ibh_halo_push / ibh_halo_pull)
        ns, nr = len(self.peers_send), len(self.peers_recv)
        if max(ns, nr) > 16:
            raise RuntimeError("XgmiHalo: more than 16 peers")
        sseg = np.concatenate([[0], np.cumsum([plan.send[q].size for q in self.peers_send])]).astype(np.int32)
        rseg = np.concatenate([[0], np.cumsum([plan.recv[q].size for q in self.peers_recv])]).astype(np.int32)
        self._sseg = (C.c_int32 * (ns + 1))(*sseg.tolist())
        self._rseg = (C.c_int32 * (nr + 1))(*rseg.tolist())
        self._dst = []
        for par in (0, 1):
            self._dst.append((C.c_void_p * max(ns, 1))(*[self.remote[q][0] + 4 * (par * self.remote[q][1] + self.remote[q][2])
                                                        for q in self.peers_send]))
        self._sflags = (C.c_void_p * max(ns, 1))(*[self.remote[q][3] for q in self.peers_send])
        self._rflags = (C.c_void_p * max(nr, 1))(*[self._flags.value + 4 * (q - 1) for q in self.peers_recv])
        if self.device.type == "cuda":
            torch.cuda.synchronize()

    def start(self, field):
        """The whole exchange, one launch (``ibh_halo_exchange``); ``finish`` has nothing left to do."""
        B, C, nv = self.B, self.C, self.nv
        f, fnv, ld = B._field_inplace(field, what="halo field")  # skirt cells are written in place
        if fnv != nv:
            raise ValueError(f"XgmiHalo was built for nv={nv}")
        self.step += 1
        par = 0  # (the kernel picks the buffer parity from its own sequence number)
        B._stream()
        if self.peers_send or self.peers_recv:
            B.call("ibh_halo_exchange", B._ptr(f), nv, ld, B._ptr(self.send_all), len(self.peers_send),
                   C.cast(self._sseg, B.c_vp), C.cast(self._dst[0], B.c_vp), C.cast(self._dst[1], B.c_vp),
                   C.cast(self._sflags, B.c_vp), B._ptr(self.recv_all), B.c_vp(self._recv.value),
                   B.c_vp(self._recv.value + 4 * self.n_recv_f), len(self.peers_recv),
                   C.cast(self._rseg, B.c_vp), C.cast(self._rflags, B.c_vp), B.c_vp(self.state.data_ptr()),
                   self.max_spins)
        return (f, ld, par)

    def finish(self, handle):
        return None

    def exchange(self, field):
        self.finish(self.start(field))
        return field

    def can_fuse(self, dpart):
        """Whether ``fused_step`` covers this partition (what ``ibh_step_advection_xgmi`` requires: a 2-D partition with
        skirt fragments whose image blocks are all eligible for the quad sweep, a scalar exchanger with peers) -- a
        rank-local predicate: callers that run collectively all-reduce it before any rank launches."""
        info = dpart.info
        has_fragments = info.get("irregular_cells", 0) > 0 or info.get("fusable_blocks") != info.get("full_blocks")
        return bool(self.nv == 1 and dpart.nd == 2 and info.get("image_blocks_all_eligible")
                    and info.get("image_quads", 0) > 0 and has_fragments and (self.peers_send or self.peers_recv))

    def fused_step(self, dpart, u, C, ud):
        """Exchange of the scalar field ``u`` + the image-only quad sweep ``ud = residual_advection(u, C)`` in ONE launch
        (``ibh_step_advection_xgmi``): the exchange workgroups run beside the interior quads, boundary waves wait for the
        unpacked skirt rows.  Raises ``IbhError`` on partitions the fused kernel does not cover (callers fall back to
        ``exchange`` + ``residual_advection(IBH_IMAGE_ONLY)``: same result)."""
        B, Cc = self.B, self.C
        if self.nv != 1:
            raise ValueError("fused_step exchanges a scalar field (XgmiHalo(nv=1))")
        f, _, _ = B._field_inplace(u, what="halo field")
        Cf, nvc, ldc = B._field(C, dpart.nc)
        o, _, _ = B._field_inplace(ud, dpart.nc, "out")
        # the ticket arithmetic of the kernel assumes ONE grid per state word pair: one pair per partition
        if getattr(self, "_fstates", None) is None:
            self._fstates = {}
        if id(dpart) not in self._fstates:
            self._fstates[id(dpart)] = torch.zeros(2, dtype=torch.int64, device=self.device)
        self._fstate = self._fstates[id(dpart)]
        self.step += 1
        B._stream()
        B.call("ibh_step_advection_xgmi", dpart.handle, B._ptr(f), B._ptr(Cf), ldc, B._ptr(o), B._ptr(self.send_all),
               len(self.peers_send), Cc.cast(self._sseg, B.c_vp), Cc.cast(self._dst[0], B.c_vp),
               Cc.cast(self._dst[1], B.c_vp), Cc.cast(self._sflags, B.c_vp), B._ptr(self.recv_all),
               B.c_vp(self._recv.value), B.c_vp(self._recv.value + 4 * self.n_recv_f), len(self.peers_recv),
               Cc.cast(self._rseg, B.c_vp), Cc.cast(self._rflags, B.c_vp), B.c_vp(self.state.data_ptr()),
               self.max_spins, B.c_vp(self._fstate.data_ptr()))
        return ud

    def healthy(self):
        """False if any wait kernel hit its spin bound (collective: every rank gets the same answer)."""
        ok = torch.tensor([1 if int(self.state[2].item()) == 0 else 0], dtype=torch.int32,
                          device=self.device if self.dist.get_backend(self.group) == "nccl" else "cpu")
        self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN, group=self.group)
        return bool(ok.item())

    def reset_health(self):
        """Clear the time-out flag (after a caller has dealt with a failed ``healthy()``)."""
        self.state[2] = 0

    def close(self):
        """Collective: peers unmap before anybody frees."""
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        self.dist.barrier(group=self.group)
        for p in self._opened:
            self.B.call("ibh_ipc_close", p)
        self._opened = []
        self.dist.barrier(group=self.group)
        self._release_local()


def verify_exchangers(a, b, nc, nv, rounds=3):
    """Run both exchangers on the same random fields; True iff every rank got identical skirt values.  Collective: a rank
    on which an exchange raises still takes part in the all-reduce (everybody then gets False)."""
    import torch.distributed as dist
    dev = a.device
    ok = True
    try:
        for r in range(rounds):
            g = torch.Generator(device="cpu").manual_seed(1234 + 17 * r + a.plan.pid)
            base = torch.rand((nv, nc), generator=g) if nv > 1 else torch.rand(nc, generator=g)
            fa, fb = base.to(dev).clone(), base.to(dev).clone()
            if nv > 1:
                fa, fb = fa.T, fb.T  # (nc, nv) column-major views
            a.exchange(fa)
            b.exchange(fb)
            if dev.type == "cuda":
                torch.cuda.synchronize()
            ok = ok and bool(torch.equal(fa, fb))
    except Exception as e:  # noqa: BLE001
        import sys
        print(f"[halo] rank {a.plan.pid - 1}: exchanger verification raised {e!r}", file=sys.stderr)
        ok = False
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if dist.get_backend(a.group) == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=a.group)
    return bool(t.item())
