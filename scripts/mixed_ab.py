"""Single-process timing of the sweep on partition 1 of N of the benchmark mesh (no exchange): mixed launch
(single kernel on the eligible blocks + two-kernel form on the rest) vs the pure two-kernel form vs the image-only
single-kernel sweep (IBH_IMAGE_ONLY), whole and in phases.
   python scripts/mixed_ab.py [nparts]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ibamd
import bench
nparts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
msh = bench.build_mesh("rae2822_0.87M")
n = len(msh)
mps = -(-(-(-n // nparts)) // 64) * 64
dom = ibamd.Domain(msh, max_partition_size=mps, boundaries=False, only=[1])
part = dom.partitions[1]
dpart = ibamd.to_backend(part, ibamd.hip)
nc = part.centers.shape[0]
rng = np.random.default_rng(1)
u = ibamd.hip(rng.uniform(-1, 1, nc).astype(np.float32))
C = ibamd.hip(np.ones((nc, 2), dtype=np.float32))
ud = torch.zeros(nc, dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()
for phases, nofuse in ((False, False), (True, False), (False, True), (True, True), (False, 'image-only'), (True, 'image-only')):
    fl = ibamd.IBH_IMAGE_ONLY if nofuse == 'image-only' else ibamd.IBH_NO_FUSE if nofuse else ibamd.IBH_FORCE_MIXED

    def step():
        if phases:
            ibamd.residual_advection(dpart, u, C, out=ud, flags=fl | ibamd.IBH_PHASE_INTERIOR)
            ibamd.residual_advection(dpart, u, C, out=ud, flags=fl | ibamd.IBH_PHASE_BOUNDARY)
        else:
            ibamd.residual_advection(dpart, u, C, out=ud, flags=fl)
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(20):
                step()
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 1000
    print(f"nparts={nparts} two_kernel_form={nofuse} phases={phases}: {dt * 1e6:.2f} us per sweep", dpart.info["fusable_blocks"], dpart.info["full_blocks"], dpart.info["irregular_cells"])

# ---- what the cross-stream fork/join of the overlapped step costs inside a graph: a stand-in "exchange" (one small
#      kernel: copy of the skirt rows) on a second stream beside the interior phase, vs everything on one stream
skirt = np.ones(nc, bool); skirt[part.image_in_domain] = False
sidx = torch.from_numpy(np.nonzero(skirt)[0].astype(np.int64)).cuda()
buf = torch.zeros(sidx.numel(), dtype=torch.float32, device="cuda")
comm = torch.cuda.Stream()
IO = ibamd.IBH_IMAGE_ONLY


def fake_exchange():
    buf.copy_(u[sidx])          # one gather kernel standing in for ibh_halo_exchange


def step_overlap():
    cur = torch.cuda.current_stream()
    comm.wait_stream(cur)
    with torch.cuda.stream(comm):
        fake_exchange()
    ibamd.residual_advection(dpart, u, C, out=ud, flags=IO | ibamd.IBH_PHASE_INTERIOR)
    cur.wait_stream(comm)
    ibamd.residual_advection(dpart, u, C, out=ud, flags=IO | ibamd.IBH_PHASE_BOUNDARY)


def step_serial():
    fake_exchange()
    ibamd.residual_advection(dpart, u, C, out=ud, flags=IO)


for name, fn in (("exchange || interior, then boundary (two streams)", step_overlap), ("exchange, then whole sweep (one stream)", step_serial)):
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(20):
                fn()
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 1000
    print(f"nparts={nparts} step with a stand-in exchange, {name}: {dt * 1e6:.2f} us")
