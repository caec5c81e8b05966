"""First contact of the N > 1 path, rehearsed on CPU with gloo (world 2): a failure of ONE rank in the set-up of the
optional direct transport (XgmiHalo: device, allocation, IPC export / import, device tables) must become the SAME error
on every rank -- nobody left waiting in a collective -- after which both ranks finish on the reference exchange with
rc 0; and a rank on which the verification of an exchanger raises still takes part in its all-reduce."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ibamd
from conftest import ADV_FAMILIES, advection_mesh, seeded_field
from ibamd.halo import HaloExchange, HaloPlan, XgmiHalo, verify_exchangers


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeXgmi(XgmiHalo):
    """The agreement logic of XgmiHalo with the device steps faked; `fail_at` names the step that raises on rank 1."""
    fail_at = None

    def _step(self, name, value=None):
        if self.fail_at == name and self.me == 1:
            raise RuntimeError(f"injected failure in {name}")
        return value

    def _setup_device(self):
        return self._step("device")

    def _ipc_alloc_export(self):
        return self._step("export", dict(recv=b"r" * 64, flags=b"f" * 64,
                                          off={int(q): int(v) for q, v in self.recv_off.items()}, n=int(self.n_recv_f)))

    def _ipc_import(self, every):
        assert all(isinstance(e, dict) and e["recv"] == b"r" * 64 for e in every)
        return self._step("import")

    def _device_tables(self):
        return self._step("tables")

    def exchange(self, field):
        raise RuntimeError("the fake transport moves nothing")


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        msh = advection_mesh(2e-2)
        ncells = len(msh)
        mps = -(-(-(-ncells // world)) // 64) * 64
        dom = ibamd.Domain(msh, max_partition_size=mps, hypercube_families=ADV_FAMILIES, boundaries=False,
                           only=[rank + 1])
        part = dom.partitions[rank + 1]
        plan = HaloPlan(dom, rank + 1)
        seen = []
        for step in ("device", "export", "import", "tables"):
            _FakeXgmi.fail_at = step
            try:
                _FakeXgmi(plan, dom, "cpu")
                seen.append("built")
            except RuntimeError as e:
                seen.append(str(e))
        # every rank saw the failure of rank 1, at the step it was injected in
        ok_err = all("rank 1" in m and f"injected failure in {s}" in m
                     for m, s in zip(seen, ("device", "export", "import", "tables")))
        _FakeXgmi.fail_at = None
        xg = _FakeXgmi(plan, dom, "cpu")      # no failure: both ranks get the object
        hx = HaloExchange(plan, "cpu")
        # its verification raises on every rank here (the fake moves nothing): collective, returns False, no hang
        ok_ver = verify_exchangers(xg, hx, part.domain.size, 1, rounds=1) is False
        # ... and the run goes on with the reference exchange
        g = seeded_field(dom.global_centers())
        local = np.array(g[part.domain])
        skirt = np.ones(local.shape[0], dtype=bool)
        skirt[part.image_in_domain] = False
        local[skirt] = np.nan
        t = torch.from_numpy(local)
        hx.exchange(t)
        ok_x = np.array_equal(t.numpy(), g[part.domain])
        res = torch.tensor([int(ok_err and ok_ver and ok_x)], dtype=torch.int32)
        dist.all_reduce(res, op=dist.ReduceOp.MIN)
        if rank == 0:
            out.put((int(res.item()), seen))
    finally:
        dist.destroy_process_group()


def test_one_rank_failure_becomes_everybodys_fallback():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    ok, seen = q.get(timeout=10)
    assert ok == 1, seen
