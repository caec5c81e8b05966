#!/usr/bin/env python3
"""Residual-sweep benchmark: Mcells*iters/s of the fused advection-JST-MUSCL residual (R1 of
SURVEY.md 8d = the closure of /root/reference/test/advection.jl:67-83) on the 2-D RAE2822
block-quadtree mesh at ~1 M cells (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, or plainly as
   `python bench.py --gpus N ...`: without WORLD_SIZE in the environment the script starts the N ranks itself)

One step = one residual sweep over the rank's partition with all fields resident in HBM; for
N > 1 the ~1 M-cell domain is cut into N contiguous partitions (one per GPU, strong scaling)
and every step first refreshes the skirt cells from their owners (RCCL grouped send/recv).
Rank 0 prints ONE JSON line.  The oracle is used only in the `cpu_baseline` leg.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
B_ALG_2D = 16.0        # algorithmic bytes per cell and sweep: u, Cx, Cy in, ud out

WORKLOADS = {
    # name: (h_wall, h_feature)  -- reference mesher semantics on test/rae2822.dat, domain [-25,25]^2
    "rae2822_0.87M": (2.5e-4, 1.25e-4),
    "rae2822_3.47M": (6e-5, 3e-5),
    "rae2822_37k": (1e-2, 5e-3),
    # secondary: working set (16 B/cell + tables) past the 256 MiB Infinity Cache: the kernel against real HBM
    "rae2822_28M": (1e-5, 5e-6),
    # 3-D (secondary): unit sphere (icosphere STL) in a [-8,8]^3 box, 8^3 blocks (BASELINE.json configs[3] shape)
    "sphere3d_1.6M": 0.06,
    "sphere3d_4.6M": 0.03,
    # BASELINE.json configs[3] shape at ~8 M cells: sphere of radius 1.4, immersed-boundary ghosts kept (--step config4)
    "sphere3d_8M": (0.03, 1.4),
    # BASELINE.json configs[4] size (the surface spacing snaps to octree levels: 4.6 M -> 15 M; the radius does the rest)
    "sphere3d_15M": 0.015,
    "sphere3d_33M": (0.015, 1.5),
}


def icosphere(radius=1.0, subdiv=3):
    """Triangulated sphere as a Stereolitography (the reference reads such surfaces from STL files)."""
    from ibamd.mesher import Stereolitography
    t = (1.0 + 5 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]])
    for _ in range(subdiv):
        nv, nf, cache = list(v), [], {}

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                nv.append((nv[a] + nv[b]) / 2)
                cache[k] = len(nv) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.array(nv), np.array(nf)
    v = v / np.linalg.norm(v, axis=1, keepdims=True) * radius
    return Stereolitography(v.T.astype(np.float32), (f.T + 1).astype(np.int64))


# Hutchinson samples per variable of the point-implicit lines (the reference's default is 30, point_implicit.jl:185-209).
# With ONE sample the block estimate of a few cells is so poor that the preconditioned direction drives their temperature
# negative in the finite-difference product and the solve returns NaN (rounds 2-3 reported that NaN at 7.9 M cells).
PI_SAMPLES = 8


def build_mesh(name):
    import ibamd
    from ibamd.mesher import DistanceField, Mesh, Stereolitography, feature_regions, merge_points
    f32 = np.float32
    if name.startswith("sphere3d"):
        w = WORKLOADS[name]
        h, radius = w if isinstance(w, tuple) else (w, 1.0)
        # the distance field of the sphere stays on the mesh: `--step config4` builds its ghost cells
        # (Domain(..., boundaries=False) -- the plain sweep benchmark -- ignores it)
        return Mesh(f32([-8, -8, -8]), f32([16, 16, 16]), ("sphere", icosphere(radius=radius), f32(h)))
    hw, hf = WORKLOADS[name]
    stl = merge_points(Stereolitography(os.path.join(ROOT, "tests", "golden", "rae2822.dat")))
    features = DistanceField(feature_regions(stl, radius=0.05))
    return Mesh(f32([-25.0, -25.0]), f32([50.0, 50.0]), ("wall", stl, f32(hw)),
                refinement_regions=[(features, f32(hf))])


def synthetic_fields(centers, seed=12345):
    """u = sin(2 pi x) cos(2 pi y) [+ 0.3 z] + 0.1 noise, C = (1, ..., 1)  (SURVEY.md 8d)."""
    rng = np.random.default_rng(seed)
    x, y = centers[:, 0].astype(np.float64), centers[:, 1].astype(np.float64)
    u = np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y) + 0.1 * rng.uniform(-1, 1, x.size)
    if centers.shape[1] == 3:
        u = u + 0.3 * centers[:, 2]
    C = np.ones((x.size, centers.shape[1]), dtype=np.float32)
    return u.astype(np.float32), C


def cpu_limits():
    """CPUs this job may really use: affinity mask and cgroup CPU quota (a box can show 256 cores and give less)."""
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                tok = f.read().split()
            if path.endswith("cpu.max"):
                quota = None if tok[0] == "max" else round(int(tok[0]) / int(tok[1]), 2)
            else:
                q = int(tok[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    quota = None if q < 0 else round(q / int(f.read().split()[0]), 2)
            break
        except (OSError, ValueError, IndexError):
            continue
    aff = len(os.sched_getaffinity(0))
    usable = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return aff, quota, usable


def cpu_baseline(part, u, C, budget_s=10.0):
    """The oracle timed on this host: the C restatement (oracle/csrc/residual.c) in both of its forms -- one array
    pass per reference broadcast like the Julia code ("faithful"), and cell-fused -- each at its best OpenMP thread
    count.  The thread counts scanned go up to min(affinity mask, cgroup quota) and are set explicitly
    (ibo_set_threads): OMP_NUM_THREADS of the box does not cap the scan.  The reported value is the faster."""
    from oracle import residual_c as rc
    cp = rc.CPart(part)
    aff, quota, usable = cpu_limits()

    def rate(fused, threads, reps):
        cp.residual_advection(u, C, fused=fused, threads=threads)
        t0 = time.perf_counter()
        for _ in range(reps):
            cp.residual_advection(u, C, fused=fused)
        return reps / (time.perf_counter() - t0)
    scan = sorted({t for t in (1, 2, 4, 8, 16, 32, 64, 128, usable) if t <= usable})
    best = {}
    for fused in (False, True):
        best[fused] = max((rate(fused, t, 3), t) for t in scan)
    fused = best[True][0] > best[False][0]
    threads = best[fused][1]
    cp.residual_advection(u, C, fused=fused, threads=threads)
    n, t0 = 0, time.perf_counter()
    while True:
        cp.residual_advection(u, C, fused=fused)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 2000:
            break
    mc = u.shape[0] / 1e6
    other = best[not fused]
    return dict(value=mc * n / dt, n=n, secs=dt, threads=threads, form="cell-fused" if fused else "faithful",
                other_form="faithful" if fused else "cell-fused", other_value=mc * other[0], other_threads=other[1],
                one_thread=mc * rate(False, 1, 2), affinity=aff, cgroup_cpus=quota, scanned=scan,
                omp_num_threads_env=os.environ.get("OMP_NUM_THREADS"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="rae2822_0.87M", choices=sorted(WORKLOADS))
    ap.add_argument("--residual", default="advection", choices=["advection", "euler"],
                    help="advection = R1 (headline, 16 B/cell); euler = R2 HLL residual (32 B/cell), secondary")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--general", action="store_true", help="force the face-list kernels (no block fast path)")
    ap.add_argument("--exact", action="store_true", help="block path with the literal IEEE arithmetic")
    ap.add_argument("--all-cells", action="store_true",
                    help="N > 1: compute the residual on skirt cells too (default: image cells only)")
    ap.add_argument("--no-fuse", action="store_true", help="keep the two-kernel sweep where one kernel could do it")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--dt-every", type=int, default=1,
                    help="--step march: evaluate the time step every that many steps (1 = every step, like march! of "
                         "test/advection.jl:65; C is constant in the script, so 10 gives the same march)")
    ap.add_argument("--dt-separate", action="store_true",
                    help="--step march: the time step as its own two launches in front of every sweep (A/B of the default, "
                         "which evaluates the next step's dt beside the boundary conditions of the step in hand)")
    ap.add_argument("--overlap", action="store_true",
                    help="N > 1: exchange on a second stream beside the interior blocks, boundary blocks after it. "
                         "Off by default: inside a HIP graph the cross-stream fork/join costs ~15 us per step "
                         "(scripts/mixed_ab.py), far more than the ~3 us it hides on partitions of this size")
    ap.add_argument("--fused-step", action="store_true",
                    help="N > 1: exchange + image-only sweep as ONE launch (XgmiHalo.fused_step: exchange workgroups "
                         "beside the interior quads, boundary waves wait for the unpacked skirt), used where it "
                         "reproduces the two-launch step bit for bit.  Opt-in: rehearsed for correctness with all "
                         "ranks on one GPU, where its timing means nothing; not yet timed on separate GPUs")
    ap.add_argument("--no-overlap", action="store_true", help="(default since the measurement above; kept for old command lines)")
    ap.add_argument("--halo", default="auto", choices=["auto", "rccl", "xgmi"],
                    help="N > 1 skirt exchange: rccl = grouped send/recv (eager launches); xgmi = direct peer writes "
                         "into IPC-mapped buffers + device flags, captured in HIP graphs; auto = xgmi if it "
                         "reproduces the rccl exchange bit for bit at start-up, else rccl")
    ap.add_argument("--graph-batch", type=int, default=20,
                    help="sweeps captured per HIP graph (launch-bound loop; 0 = eager launches)")
    ap.add_argument("--launch", default="auto", choices=["auto", "graph", "c-loop"],
                    help="how the K sweeps of a timed block are launched (one GPU, scalar sweep): HIP graphs of --graph-batch "
                         "sweeps, or one C call that launches them back to back (ibh_residual_advection_n: the step loop of "
                         "a compiled host); auto = both timed on a few blocks, the faster one is used and both are recorded")
    ap.add_argument("--step", default="sweep", choices=["sweep", "march", "config4", "config5"],
                    help="config4 (3-D workloads, --residual euler): a step = impose_bc! with FlowBC closures on the "
                         "immersed sphere and the far field (ghost-layer interpolation) + the Euler residual sweep; one "
                         "point-implicit linearise + relaxation is timed beside it; config5: a step = one FAS! V-cycle "
                         "(3 levels, 2 smoothing iterations each) of the Euler + Wray-Agarwal residual")
    ap.add_argument("--repeats", type=int, default=20,
                    help="the timed block of --steps sweeps is repeated this many times; the median is reported")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="check the launch path only: ranks meet (gloo), rank 0 prints one JSON line; no GPU is touched")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly: become the launcher.  Nothing in this process has touched the GPU (torch is not even
        # imported yet); the ranks are children of torch.distributed.run and rank 0's JSON line is relayed.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
        lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
        if lines:
            print(lines[-1])
        else:
            sys.stderr.write(proc.stdout)
        raise SystemExit(proc.returncode if proc.returncode else (0 if lines else 1))

    if args.rendezvous_only:
        import torch
        import torch.distributed as dist
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.ones(1, dtype=torch.int64)
            dist.all_reduce(t)
            ok = int(t.item()) == world
            rank = dist.get_rank()
            dist.destroy_process_group()
        else:
            ok, rank = True, 0
        if rank == 0:
            print(json.dumps({"n_gpus": world, "rendezvous": "ok" if ok else "failed"}))
        raise SystemExit(0 if ok else 1)

    import torch
    import torch.distributed as dist
    import ibamd
    from ibamd import _lib
    _lib.load()  # fail loudly if the HIP library is missing

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import datetime
        pg_timeout = datetime.timedelta(seconds=120)   # a rank that dies in set-up must not hold the others for 10 min
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
        else:
            dist.init_process_group("gloo", timeout=pg_timeout)

    msh = build_mesh(args.workload)
    ncells = len(msh)
    npb = msh.block_size ** msh.ndims
    mps = -(-ncells // world)
    mps = -(-mps // npb) * npb  # block-aligned partitions
    config4 = args.step == "config4"
    config5 = args.step == "config5"
    march = args.step == "march"
    if march and (msh.ndims != 2 or args.residual != "advection" or world != 1):
        raise SystemExit("--step march: the explicit step of test/advection.jl on a 2-D workload, one GPU")
    if (config4 or config5) and (msh.ndims != 3 or args.residual != "euler"):
        raise SystemExit("--step config4 / config5 need a sphere3d workload and --residual euler")
    fam4 = [("farfield", [(d, s_) for d in (1, 2, 3) for s_ in (False, True)])]
    fam2 = [("farfield", [(1, False), (1, True), (2, False), (2, True)])]
    bc5 = config5                      # configs[4] with its boundary conditions on every level
    dom = ibamd.Domain(msh, max_partition_size=mps, boundaries=config4 or march or bc5, only=[rank + 1],
                       hypercube_families=fam4 if (config4 or bc5) else fam2 if march else ())
    part = dom.partitions[rank + 1]
    n_image = int(part.image.size)
    u_h, C_h = synthetic_fields(part.centers)
    dpart = ibamd.to_backend(part, ibamd.hip)
    u, C = ibamd.hip(u_h), ibamd.hip(C_h)
    ud = torch.zeros(dpart.nc, dtype=torch.float32, device=u.device)
    euler = args.residual == "euler"
    if euler:
        rng = np.random.default_rng(12345)
        n = part.centers.shape[0]
        nvp = msh.ndims + 2
        P_h = np.empty((n, nvp), dtype=np.float32)   # P = [p T u v (w)], SURVEY.md 8d
        P_h[:, 0] = 1e5 * (1 + 0.05 * rng.uniform(-1, 1, n))
        P_h[:, 1] = 288.15 * (1 + 0.05 * rng.uniform(-1, 1, n))
        for k in range(2, nvp):
            P_h[:, k] = 100.0 * (1 + 0.1 * rng.uniform(-1, 1, n))
        if config4:   # a flow past the sphere: free stream along x with small perturbations (tests/test_config4.py)
            P_h[:, 0] = 1e5 * (1 + 0.02 * rng.uniform(-1, 1, n))
            P_h[:, 1] = 288.15 * (1 + 0.02 * rng.uniform(-1, 1, n))
            P_h[:, 3:] = 10.0 * rng.uniform(-1, 1, (n, nvp - 3))
        P = ibamd.hip(P_h)
        Rres = torch.zeros((nvp, dpart.nc), dtype=torch.float32, device=P.device).T
    flags = (ibamd.IBH_FORCE_GENERAL if args.general else 0) | (ibamd.IBH_EXACT if args.exact else 0)
    flags |= ibamd.IBH_NO_FUSE if args.no_fuse else 0
    # a rank of a multi-GPU run needs the residual on its image cells only (what dom(f, args...) scatters back,
    # ImmersedBoundary.jl:842-845): with every image block eligible that is one launch per overlap phase
    image_only = (world > 1 and flags == 0 and dpart.info["image_blocks_all_eligible"]
                  and not args.all_cells)
    if image_only:
        flags |= ibamd.IBH_IMAGE_ONLY

    hx = None
    auto_forms = False
    comm_stream = None
    halo_kind = None
    fused_step = False
    if world > 1 and (config4 or config5):
        from ibamd.halo import HaloExchange, HaloPlan   # (these steps bring their own exchangers, below)
        halo_kind = "rccl" if args.backend == "nccl" else "gloo-staged"
    if world > 1 and not (config4 or config5):
        from ibamd.halo import (HaloExchange, HaloPlan, XgmiHalo, euler_sweep_overlapped, sweep_overlapped,
                                verify_exchangers)
        plan = HaloPlan(dom, rank + 1)
        hx = HaloExchange(plan, u.device)
        halo_kind = "rccl" if args.backend == "nccl" else "gloo-staged"
        if args.halo in ("auto", "xgmi"):
            try:
                nvx = (msh.ndims + 2) if euler else 1
                xg = XgmiHalo(plan, dom, u.device, nv=nvx)
                good = verify_exchangers(xg, hx, dpart.nc, nvx, rounds=3) and xg.healthy()
                if good:
                    hx, halo_kind = xg, "xgmi-direct"
                elif args.halo == "xgmi":
                    raise SystemExit("xgmi halo exchange failed verification against the reference exchange")
                else:
                    xg.close()
            except Exception as e:  # noqa: BLE001 -- any failure of the optional transport falls back to RCCL
                if args.halo == "xgmi":
                    raise
                if rank == 0:
                    print(f"[bench] xgmi halo exchange unavailable ({e}); using {halo_kind}", file=sys.stderr)
        # (the Euler sweep has overlap phases only where the image blocks all take the single kernel)
        auto_forms = args.halo == "auto" and not (args.overlap or args.fused_step or args.no_overlap)
        # (3-D image-only sweeps have no overlap phases: one launch over the image blocks)
        can_overlap = (not args.general and dpart.info["interior_blocks"] > 0 and (not euler or image_only)
                       and not (msh.ndims == 3 and image_only))
        overlap = args.overlap and not args.no_overlap and can_overlap
        comm_stream = torch.cuda.Stream() if overlap else None
        # exchange + image-only quad sweep as ONE launch (XgmiHalo.fused_step: the exchange workgroups run beside the
        # interior quads), taken when it reproduces exchange-then-sweep bit for bit on every rank
        # Every rank must take the same branch (the trial holds collectives): the rank-local predicate is all-reduced.
        red0 = u.device if args.backend == "nccl" else "cpu"
        elig = int(bool((args.fused_step or auto_forms) and halo_kind == "xgmi-direct" and image_only and not euler
                        and comm_stream is None and flags == ibamd.IBH_IMAGE_ONLY))
        te = torch.tensor([elig], dtype=torch.int32, device=red0)
        dist.all_reduce(te, op=dist.ReduceOp.MIN)
        if int(te.item()):
            ok_f = 0
            try:
                ref_f = torch.zeros_like(ud)
                hx.exchange(u)
                ibamd.residual_advection(dpart, u, C, out=ref_f, flags=flags)
                got_f = torch.zeros_like(ud)
                torch.cuda.synchronize()
                ok_f = 1
            except Exception as e:  # noqa: BLE001
                print(f"[bench] rank {rank}: reference step of the fused trial failed ({e})", file=sys.stderr)
            # a rank that cannot launch the fused kernel would leave its peers' sequence numbers one step ahead: the
            # decision to launch is collective too
            tf = torch.tensor([ok_f and int(hx.can_fuse(dpart))], dtype=torch.int32, device=red0)
            dist.all_reduce(tf, op=dist.ReduceOp.MIN)
            launched = bool(int(tf.item()))
            ok_f = 0
            if launched:
                try:
                    hx.fused_step(dpart, u, C, got_f)
                    torch.cuda.synchronize()
                    ok_f = int(bool(torch.equal(got_f, ref_f)))
                except Exception as e:  # noqa: BLE001
                    print(f"[bench] rank {rank}: fused exchange + sweep step failed ({e})", file=sys.stderr)
            elif rank == 0:
                print("[bench] fused exchange + sweep step: not available on every rank's partition", file=sys.stderr)
            tf = torch.tensor([ok_f and int(hx.healthy())], dtype=torch.int32, device=red0)
            dist.all_reduce(tf, op=dist.ReduceOp.MIN)
            fused_step = bool(tf.item())
            if fused_step:
                halo_kind = "xgmi-direct, fused with the sweep (one launch per step)"
            elif launched:
                # the exchanger's sequence numbers may be out of step after a failed trial: do not keep it
                if args.halo == "xgmi":
                    raise SystemExit("fused exchange + sweep trial failed on a rank; rerun without --fused-step")
                if rank == 0:
                    print("[bench] fused-step trial failed on a rank: closing the xgmi exchanger, using the "
                          "reference exchange", file=sys.stderr)
                hx.close()
                hx = HaloExchange(plan, u.device)
                halo_kind = "rccl" if args.backend == "nccl" else "gloo-staged"

    def sweep(extra=0):
        if euler:
            ibamd.residual_euler_hll(dpart, P, out=Rres, flags=flags | extra)
        else:
            ibamd.residual_advection(dpart, u, C, out=ud, flags=flags | extra)

    dist4 = dist5 = None
    if config4:
        from ibamd import cfd as gcfd
        far_bc = gcfd.FlowBC(gcfd.Fluid(), [1.0e5, 288.15, 100.0, 0.0, 0.0])
        wall_bc = gcfd.FlowBC(gcfd.Fluid(), [1.0e5, 288.15, 0.0], normal_flow=True)
        bdom4 = dom
        if world > 1:
            # one partition per rank (distributed.py): the ghosts this rank owns, re-indexed to its local rows, their donor
            # cells beyond the skirt appended to the local arrays and to the halo lists
            from ibamd.distributed import LocalDomain, bc_donor_extras
            extras4 = bc_donor_extras(dom)
            bdom4 = LocalDomain(dom, rank + 1, extras4)
            hx4 = HaloExchange(HaloPlan(dom, rank + 1, extra=extras4), u.device)
            P4 = ibamd.colmajor_empty(bdom4.n_rows, nvp)
            P4[:] = P[0]
            P4[:dpart.nc] = P
            dist4 = {"extra_rows": int(bdom4.n_rows - dpart.nc), "recv_cells": hx4.plan.n_recv}
        for bname in bdom4.boundaries:                # device-resident Boundary structs, built before the timed region
            for b in bdom4.boundaries[bname].values():
                ibamd.to_backend(b, ibamd.hip)

    if config5 and world > 1:
        # FAS! across ranks (distributed.RankLevels): every rank holds its partition of every level, the transfer operators
        # restricted to its rows, a halo exchange per level; norms all-reduced over the owned cells
        from ibamd.closures import config5_boundary_conditions, navier_stokes_wray_agarwal_residual
        from ibamd.distributed import RankLevels, Reductions
        lv5 = RankLevels(msh, rank + 1, world, 2, domain_kwargs=dict(hypercube_families=fam4))
        levels5 = [dpart] + [ibamd.to_backend(p_, ibamd.hip) for p_ in lv5.parts[1:]]
        hx5 = [HaloExchange(pl, u.device) for pl in lv5.plans]
        red5 = [Reductions(p_.image_in_domain, device=u.device) for p_ in lv5.parts]
        coar5, prol5 = lv5.coarseners, lv5.prolongators
        for a5 in list(prol5) + list(coar5):
            ibamd.to_backend(a5)
        Q5 = ibamd.colmajor_empty(lv5.nrows[0], nvp + 1)
        Q5[:] = 0.0
        Q5[:dpart.nc, :nvp] = P
        Q5[:, nvp] = 4.5e-5
        hx5[0].exchange(Q5)
        Q5_0 = Q5.clone()
        ncs5 = [int(l_.nc) for l_ in levels5]

        FAR5 = [1.0e5, 288.15, 100.0, 0.0, 0.0]
        ghosts5 = []
        for ld5 in lv5.local_doms:                     # device-resident Boundary structs, built before the timed region
            ghosts5.append({k: int(sum(b.ghost_indices.size for b in v.values())) for k, v in ld5.boundaries.items()})
            for v5 in ld5.boundaries.values():
                for b5 in v5.values():
                    ibamd.to_backend(b5, ibamd.hip)

        def f5(level, Q):
            # (FAS! has refreshed the skirt and donor rows): impose_bc! on the ghosts this rank owns, on the level's own
            # boundaries; a second exchange brings the ghosts the peers own; then the residual on the local partition
            nc_l = ncs5[level]
            config5_boundary_conditions(lv5.local_doms[level], Q, FAR5)
            hx5[level].exchange(Q)
            if Q.shape[0] == nc_l:
                return navier_stokes_wray_agarwal_residual(levels5[level], Q), 2e-7
            r = ibamd.colmajor_empty(Q.shape[0], nvp + 1)
            r[nc_l:] = 0.0
            r[:nc_l] = navier_stokes_wray_agarwal_residual(levels5[level], Q[:nc_l])
            return r, 2e-7
        dist5 = {"levels_rows": [int(n_) for n_ in lv5.nrows], "levels_cells": ncs5,
                 "extra_rows": [int(lv5.nrows[l_] - ncs5[l_]) for l_ in range(3)],
                 "recv_cells": [int(pl.n_recv) for pl in lv5.plans], "owned_ghost_cells": ghosts5}
    elif config5:
        # one GPU: the level closure of configs[4] as a solver script would write it -- impose_bc! on the level's own
        # Boundary structs (FlowBC free stream; slip wall with wall_function), then Euler HLL + viscous_fluxes(mu + mu_t) +
        # Wray-Agarwal transport (closures.py; tests/test_config5.py runs the same closure against the oracle)
        from ibamd.closures import config5_boundary_conditions, navier_stokes_wray_agarwal_residual
        cds5, prol5, coar5 = ibamd.multigrid(dom, max_levels=2)
        doms5 = [dom] + list(cds5)
        levels5 = [dpart] + [ibamd.to_backend(d.partitions[1], ibamd.hip) for d in cds5]
        for a5 in list(prol5) + list(coar5):
            ibamd.to_backend(a5)                       # transfer operators on the device before the timed region
        ghosts5 = []
        for d5 in doms5:                               # device-resident Boundary structs, built before the timed region
            ghosts5.append({k: int(sum(b.ghost_indices.size for b in v.values())) for k, v in d5.boundaries.items()})
            for v5 in d5.boundaries.values():
                for b5 in v5.values():
                    ibamd.to_backend(b5, ibamd.hip)
        Q5 = ibamd.colmajor_empty(dpart.nc, nvp + 1)
        Q5[:, :nvp] = P
        Q5[:, nvp] = 4.5e-5
        Q5_0 = Q5.clone()
        FAR5 = [1.0e5, 288.15, 100.0, 0.0, 0.0]

        def f5(level, Q):
            config5_boundary_conditions(doms5[level], Q, FAR5)
            return navier_stokes_wray_agarwal_residual(levels5[level], Q), 2e-7

    step_form = {"fused": fused_step, "overlap": comm_stream is not None}

    if march:
        # march! of test/advection.jl:61-89, device resident: dt by a device reduction (every step by default, --dt-every),
        # sweep + u .+= ud .* dt in one launch, the impose_bc! calls as one BC set (wall: value 0, far field: copy(u))
        bcs_m = ibamd.BCSet(dom, [(name, "copy" if name == "farfield" else 0.0) for name in dom.boundaries], ipart=rank + 1)
        dt_m = ibamd.timestep_advection(dpart, C, scale=0.75)
        um = [u.clone(), torch.empty_like(u)]
        mstate = {"k": 0}

    def step():
        if march:
            k = mstate["k"]
            if args.dt_every <= 1 and not args.dt_separate:
                # dt of the NEXT step evaluated beside this step's boundary conditions (it depends on C alone): extra
                # workgroups of the BC set's two launches instead of two launches in front of the next sweep
                ibamd.step_advection(dpart, um[k & 1], C, dt_m, bcs_m, out=um[(k + 1) & 1], next_dt=dt_m, scale=0.75)
            else:
                if k % max(1, args.dt_every) == 0:
                    ibamd.timestep_advection(dpart, C, scale=0.75, out=dt_m)
                ibamd.step_advection(dpart, um[k & 1], C, dt_m, bcs_m, out=um[(k + 1) & 1])
            mstate["k"] = k + 1
            return
        if config5 and world > 1:
            Q5.copy_(Q5_0)
            ibamd.FAS(f5, Q5, coarseners=coar5, prolongators=prol5, n_iter=2, rtol=1e-9,
                      exchange=lambda l_, Q_: hx5[l_].exchange(Q_),
                      level_norm=lambda l_, r_: red5[l_].norm(r_[:ncs5[l_]]))
        elif config5:
            # (no reset of Q between the steps: a solver's V-cycles follow one another; n_iter is fixed, rtol unreachable)
            ibamd.FAS(f5, Q5, coarseners=coar5, prolongators=prol5, n_iter=2, rtol=1e-9)
        elif config4 and world > 1:
            hx4.exchange(P4)                              # skirt and donor rows as the owners have them
            ibamd.impose_bc(lambda b, ia: far_bc(ia, b.normals), bdom4, "farfield", P4)
            ibamd.impose_bc(lambda b, ia: wall_bc(ia, b.normals), bdom4, "sphere", P4)
            hx4.exchange(P4)                              # the ghosts the peers own, for the sweep
            ibamd.residual_euler_hll(dpart, P4[:dpart.nc], out=Rres, flags=flags)
        elif config4:
            ibamd.impose_bc(lambda b, ia: far_bc(ia, b.normals), dom, "farfield", P)
            ibamd.impose_bc(lambda b, ia: wall_bc(ia, b.normals), dom, "sphere", P)
            sweep()
        elif hx is None:
            sweep()
        elif step_form["fused"]:
            hx.fused_step(dpart, u, C, ud)
        elif step_form["overlap"] and euler:
            euler_sweep_overlapped(hx, dpart, P, Rres, comm_stream, flags=flags)
        elif step_form["overlap"]:
            sweep_overlapped(hx, dpart, u, C, ud, comm_stream, flags=flags)
        else:
            hx.exchange(P if euler else u)
            sweep()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # --halo auto on real peers: the step forms that are available here -- exchange-then-sweep, the fused launch (if its
    # trial reproduced exchange-then-sweep bit for bit on every rank), the overlapped phases (if they do) -- are timed for
    # 50 steps each and the fastest is taken; every decision is collective (all-reduced), the three times are reported.
    auto_step_us = None
    if world > 1 and auto_forms and not (config4 or config5):
        red0 = u.device if args.backend == "nccl" else "cpu"

        def agree(flag):
            t = torch.tensor([int(bool(flag))], dtype=torch.int32, device=red0)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())
        base_kind = "xgmi-direct" if (halo_kind or "").startswith("xgmi-direct") else halo_kind
        forms = [("exchange-then-sweep", False, False)]
        if fused_step:
            forms.append(("fused", True, False))
        if agree(can_overlap):
            ok_o = False
            try:   # the overlapped phases must reproduce exchange-then-sweep bit for bit before they are timed
                comm_stream = torch.cuda.Stream()
                ref_o = (Rres if euler else ud).clone()
                step_form.update(fused=False, overlap=False)
                step()
                torch.cuda.synchronize()
                ref_o.copy_(Rres if euler else ud)
                step_form.update(overlap=True)
                step()
                torch.cuda.synchronize()
                ok_o = bool(torch.equal(ref_o, Rres if euler else ud))
            except Exception as e:  # noqa: BLE001
                print(f"[bench] rank {rank}: overlapped step failed ({e})", file=sys.stderr)
            if agree(ok_o):
                forms.append(("overlap", False, True))
        can_graph = (halo_kind or "").startswith("xgmi-direct")
        auto_step_us = {}
        for name, fz, ov in forms:
            step_form.update(fused=fz, overlap=ov)
            nrep, gsz = 5, 10
            torch.cuda.synchronize()
            dist.barrier()
            sg = torch.cuda.Stream()
            g = None
            with torch.cuda.stream(sg):
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                if can_graph:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=sg):
                        for _ in range(gsz):
                            step()
                torch.cuda.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                for _ in range(nrep):
                    if g is not None:
                        g.replay()
                    else:
                        for _ in range(gsz):
                            step()
                torch.cuda.synchronize()
            dist.barrier()
            tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red0)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            auto_step_us[name] = round(float(tt.item()) / (nrep * gsz) * 1e6, 3)
            del g
        best = min(auto_step_us, key=auto_step_us.get)     # (identical on every rank: the times were all-reduced)
        fused_step = best == "fused"
        step_form.update(fused=fused_step, overlap=best == "overlap")
        if best != "overlap":
            comm_stream = None
        halo_kind = base_kind if best == "exchange-then-sweep" else (
            "xgmi-direct, fused with the sweep (one launch per step)" if fused_step else (base_kind or "") + ", overlapped phases")

    # The sweep is ~10 us of GPU work: a Python/ctypes launch per step would be host-bound, so on one GPU
    # the step loop is captured into HIP graphs of `graph_batch` sweeps each (every sweep still runs in full).
    # (N > 1: only with the xgmi exchange, which is kernels only; RCCL calls are launched eagerly.)
    graphable = (world == 1 or (halo_kind or "").startswith("xgmi-direct")) and not (config4 or config5)   # (closures launch eagerly)
    batch = args.graph_batch if (graphable and args.graph_batch > 0) else 0
    graph = None
    side = torch.cuda.Stream()
    if batch:
        batch = min(batch, args.steps)
        with torch.cuda.stream(side):
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(batch):
                    step()
        torch.cuda.synchronize()

    def run(nsteps):
        if graph is None:
            for _ in range(nsteps):
                step()
            return
        with torch.cuda.stream(side):
            for _ in range(nsteps // batch):
                graph.replay()
            for _ in range(nsteps % batch):  # (the exchange kernel keeps its buffer parity on the device: any mix works)
                step()

    # the plain scalar sweep on one GPU can also be looped by ONE C call (no interpreter between the launches, no graph)
    plain = world == 1 and hx is None and not (euler or march or config4 or config5)
    launch_forms = None
    c_loop = False
    if plain and args.launch != "graph":
        def run_c(nsteps):
            with torch.cuda.stream(side):
                ibamd.residual_advection_repeat(dpart, u, C, ud, nsteps, flags=flags)

        def block_us(fn, blocks=30):
            ts = []
            for _ in range(blocks):
                barrier()
                t0 = time.perf_counter()
                fn(args.steps)
                barrier()
                ts.append(time.perf_counter() - t0)
            ts.sort()
            return ts[len(ts) // 2] / args.steps * 1e6
        run(args.warmup)
        run_c(args.warmup)
        launch_forms = {"hip-graph": round(block_us(run), 3), "c-loop": round(block_us(run_c), 3)}
        c_loop = args.launch == "c-loop" or launch_forms["c-loop"] < launch_forms["hip-graph"]
        if c_loop:
            run = run_c
    run(args.warmup)
    # EXACTLY `steps` sweeps between barrier + synchronize on both sides, max over ranks -- and that block `repeats`
    # times: a block is a fraction of a millisecond, its wall time moves with launch jitter; the median is reported
    dts, gts = [], []
    ev_stream = side if graph is not None else torch.cuda.current_stream()
    # short blocks (a graph of 20 sweeps is ~0.13 ms): 20 blocks are over before the device clocks have settled (the
    # median drifts from 7.3 to 7.0 us per sweep between 20 and >= 200 blocks) -- keep going until the blocks add up to 0.2 s
    # or 400 of them have run; every block is exactly `steps` steps between barrier + synchronize
    n_rep = max(1, args.repeats)
    n_goal = None                      # total number of blocks, decided (by all ranks alike) after the first n_rep
    # configs[3] imposes its boundary conditions on a FROZEN field every step (nothing updates the cells): the ghost values
    # feed their own image-point interpolation, and over hundreds of steps the temperature of a few ghost cells drifts
    # linearly (-1.7 K per step at 7.9 M cells) until the sound speed is NaN.  Every timed block starts from the initial
    # state (restored outside the timed region), as a solver's step would start from a state its update produced.
    state4 = None
    if config4:
        state4 = (P4, P4.clone()) if world > 1 else (P, P.clone())
    while len(dts) < n_rep or len(dts) < n_goal:
        if state4 is not None:
            state4[0].copy_(state4[1])
            torch.cuda.synchronize()
        barrier()
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        g0.record(ev_stream)
        run(args.steps)
        g1.record(ev_stream)
        barrier()
        dts.append(time.perf_counter() - t0)
        gts.append(g0.elapsed_time(g1) * 1e-3)   # GPU side of the same block (events on the launch stream)
        if len(dts) == n_rep and n_goal is None:
            tot = sum(dts)
            if world > 1:                # the same count on every rank: the slowest rank's total decides
                tt = torch.tensor([tot], dtype=torch.float64, device=u.device if args.backend == "nccl" else "cpu")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                tot = float(tt.item())
            n_goal = n_rep if tot >= 0.2 else min(max(n_rep, 400), int(n_rep * 0.2 / max(tot, 1e-9)) + 1)
    if world > 1:
        red_dev = u.device if args.backend == "nccl" else "cpu"
        tmax = torch.tensor(dts, dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dts = [float(x) for x in tmax.tolist()]
    dts.sort()
    dt = dts[len(dts) // 2]
    gts.sort()
    gpu_dt = gts[len(gts) // 2]
    halo_timeouts = None
    if world > 1:
        if (halo_kind or "").startswith("xgmi-direct"):
            # a wait kernel that hit its spin bound unpacked stale skirt values: the figure would be for a wrong
            # residual.  Checked after the timed region (collective); such a run does not publish a number.
            halo_timeouts = 0 if hx.healthy() else 1
            if halo_timeouts:
                raise SystemExit("xgmi halo exchange: a wait timed out during the run (stale skirt values); "
                                 "rerun with --halo rccl")
        tot = torch.tensor([n_image], dtype=torch.int64, device=red_dev)
        dist.all_reduce(tot)
        total_cells = int(tot.item())
    else:
        total_cells = n_image
    ms_per_step = dt / args.steps * 1e3
    value = total_cells * args.steps / dt / 1e6

    # --- what this box's memory system delivers (report the fraction against the measured device
    #     bandwidth as well): 1 GiB device-to-device copy and a triad a = b + s*c over 3 x 1 GiB, HIP events
    def measured_bandwidth():
        n = 1 << 28
        a = torch.empty(n, dtype=torch.float32, device=u.device)
        b = torch.ones(n, dtype=torch.float32, device=u.device)
        c = torch.ones(n, dtype=torch.float32, device=u.device)
        res = {}
        for name, fn, nbytes in (("copy", lambda: a.copy_(b), 8 * n), ("triad", lambda: torch.add(b, c, alpha=0.5, out=a), 12 * n)):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[name] = nbytes * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b, c
        return res
    bw = measured_bandwidth() if world == 1 else None

    # --- roofline of the dominant kernel (pass B: MUSCL + flux + Green-Gauss), HIP events on the launch stream
    def time_pass(f, reps):
        # `reps` launches of one kernel captured in a graph, timed with events on the launch stream
        with torch.cuda.stream(side):
            sweep()  # valid workspace
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(reps):
                    sweep(f)
            g.replay()
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(side)
            for _ in range(5):
                g.replay()
            ev1.record(side)
            torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / (5 * reps) * 1e-3
    reps = 50
    cells_launch = dpart.nc
    is3d = msh.ndims == 3
    # single-kernel sweep: every block eligible, no face-list cells (2-D scalar sweep on one partition)
    inf = dpart.info
    base_flags = flags & ~ibamd.IBH_IMAGE_ONLY
    fused = (not euler and not is3d and base_flags == 0 and inf["irregular_cells"] == 0
             and inf["fusable_blocks"] == inf["full_blocks"] > 0)
    # partitions with skirt fragments: image-only single-kernel sweeps, or eligible blocks in the single kernel and
    # the rest in the two-kernel form (only where that saves more than the extra launch)
    mixed = (not euler and not is3d and base_flags == 0 and not fused and not image_only
             and inf["fusable_blocks"] >= 12000 and 4 * inf["fusable_blocks"] >= inf["full_blocks"])
    fused_e = euler and not is3d and flags == 0 and inf["irregular_cells"] == 0 and inf["fusable_blocks"] == inf["full_blocks"] > 0
    # 3-D scalar sweep: single kernel over the eligible blocks (+ the two-kernel form over the rest, same sweep)
    fused3 = (is3d and flags == 0 and inf["irregular_cells"] == 0 and inf["fusable_blocks"] == inf["full_blocks"] > 0)
    if image_only:
        tB, tA = time_pass(0, reps), None
        cells_launch = n_image
    elif fused or fused_e or fused3:
        tB, tA = time_pass(0, reps), None
    elif mixed:
        tB, tA = time_pass(ibamd.IBH_SWEEP_ONLY, reps), None
        cells_launch = 64 * inf["fusable_blocks"]
    else:
        tB = time_pass(ibamd.IBH_PASS_B_ONLY, reps)
        tA = time_pass(ibamd.IBH_PASS_A_ONLY, reps)
    # R1 = 4*(1 + nd + 1) B/cell, R2 = 2 * 4 * (nd + 2) B/cell
    b_alg = (40.0 if is3d else 32.0) if euler else (20.0 if is3d else B_ALG_2D)
    # two-kernel workloads: the algorithmic bytes are priced against pass A + pass B (the sweep), not pass B alone
    t_kern = tB + (tA or 0.0)
    achieved = b_alg * cells_launch / t_kern / 1e9
    # HBM-side traffic of one launch of the dominant kernel from the committed PMC passes of this build (separate
    # `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` runs of this same command; FETCH_SIZE x2 on gfx950,
    # calibrated against the kernel's known tile loads); null if not profiled.
    kernel = ("k_sweep3_euler_cols" if (euler and (fused3 or (image_only and is3d))) else "k_passB3e_blk" if (euler and is3d) else ("k_sweep_quad_euler" if inf.get("image_quads" if image_only else "quads", 0) > 0 and os.environ.get("IBH_QUAD", "1") != "0"
               else "k_sweep_euler") if (fused_e or (image_only and euler)) else "k_passB_euler_blk" if euler else
              "k_sweep3_cols" if (fused3 or (image_only and is3d)) else "k_passB3_adv_blk" if is3d else
              "k_sweep_quad" if ((fused and inf.get("quads", 0) > 0) or
                                 (image_only and not euler and inf.get("image_quads", 0) > 0)) else
              "k_sweep_adv" if (fused or mixed or (image_only and not euler)) else "k_passB_adv<2,false>")
    traffic = None
    pmc_extra = {}
    try:
        with open(os.path.join(ROOT, "profiles", "current_pmc.json")) as f:
            pm = json.load(f)
        pm = next((e_ for e_ in pm.get("entries", [pm]) if e_.get("workload") == args.workload and e_.get("kernel") == kernel), {})
        if pm and world == 1:
            traffic = round((2.0 * pm["fetch_kb"] + pm["write_kb"]) * 1024.0)
            pmc_extra = {k: pm[k] for k in ("valu_insts_per_wave", "valu_busy_frac", "lds_insts_per_wave",
                                            "wave_wait_frac") if k in pm}
    except (OSError, KeyError, ValueError):
        pass
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": kernel,
                "kernel_us": round(tB * 1e6, 3), "passA_us": None if tA is None else round(tA * 1e6, 3),
                "frac_priced_on": ("pass A + pass B" if tA is not None else
                                   "the whole sweep: single kernel over %d blocks + two-kernel form over %d" %
                                   (inf["fusable_blocks"], inf["full_blocks"] - inf["fusable_blocks"])
                                   if fused3 and inf["fusable_blocks"] < inf["full_blocks"] else "the one kernel of the sweep"),
                "alg_bytes_per_cell": b_alg, "cells_per_launch": cells_launch,
                "sweep_frac": round(b_alg * cells_launch / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}
    if pmc_extra:
        roofline["binding_limit"] = dict(pmc_extra, what="launch ramp + two dependent load round trips, then VALU issue "
                                                         "(profiles/current_pmc.json, DESIGN.md section 3)")
    if kernel == "k_sweep_quad" and world == 1:
        # what the launch itself costs on this box: the same grid with the work stripped down (ibh_probe_sweep)
        from ibamd import backend as _bk

        def probe(mode):
            def f(_=0):
                _bk._stream()
                _lib.call("ibh_probe_sweep", dpart.handle, u.data_ptr(), C.data_ptr(), C.stride(1), ud.data_ptr(), mode)
            return f
        keep = sweep
        try:
            sweep = probe(0)
            t_disp = time_pass(0, reps)
            sweep = probe(1)
            t_stream = time_pass(0, reps)
        finally:
            sweep = keep
        roofline["ceiling"] = {"dispatch_only_us": round(t_disp * 1e6, 3), "stream_16B_per_cell_us": round(t_stream * 1e6, 3),
                               "frac_of_a_pure_stream_kernel": round(b_alg * cells_launch / t_stream / 1e9 / HBM_PEAK_GBS, 4),
                               "what": "same grid: every wave returns at once / loads its cells of u, Cx, Cy and stores ud"}
    if bw:
        roofline.update(measured_copy_gbs=round(bw["copy"], 1), measured_triad_gbs=round(bw["triad"], 1),
                        frac_of_measured_triad=round(achieved / bw["triad"], 4))

    out = {
        "metric": "Mcells*iters/s residual sweep (%s), %s" % ("Euler HLL-JST-MUSCL" if euler else "advection-JST-MUSCL",
                                                              "3D sphere" if is3d else "2D RAE2822"),
        "value": round(value, 2), "unit": "Mcells*iters/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "repeats": len(dts), "ms_per_step": round(ms_per_step, 5),
        "gpu_ms_per_step": round(gpu_dt / args.steps * 1e3, 5),
        "ms_per_step_min_max": [round(dts[0] / args.steps * 1e3, 5), round(dts[-1] / args.steps * 1e3, 5)],
        "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {'3D sphere block octree' if is3d else '2D RAE2822 block quadtree'}, "
                               f"{ncells} cells, {msh.nblocks} {'8x8x8' if is3d else '8x8'} blocks, "
                               f"{world} partition(s), skirt depth 2, "
                               f"{'R2 Euler HLL' if euler else 'R1 advection'}-JST-MUSCL residual, fields resident in HBM",
                   "cells_total": ncells, "cells_per_rank_with_skirt": int(dpart.nc),
                   "path": "face-list" if args.general else ("block-fast-path-literal" if args.exact else
                            "block-fast-path, single kernel on the image blocks" if image_only else
                            "block-fast-path, single kernel" if (fused or fused_e or fused3) else
                            "block-fast-path, single kernel on %d of %d blocks" % (inf["fusable_blocks"], inf["full_blocks"])
                            if mixed else "block-fast-path, two kernels"),
                   "launch": (f"c-loop x{args.steps} (one C call launches the sweeps of a block back to back)" if c_loop else
                              f"hip-graph x{batch}" if batch else "eager"),
                   "launch_forms_us_per_step": launch_forms,
                   "halo": None if hx is None else {"backend": args.backend, "exchange": halo_kind,
                                                    "overlap": bool(step_form["overlap"]), "timeouts": halo_timeouts,
                                                    "auto_step_us": auto_step_us,
                                                    "send_cells": hx.plan.n_send, "recv_cells": hx.plan.n_recv,
                                                    "peers": len(hx.plan.peers),
                                                    "interior_blocks": dpart.info["interior_blocks"]},
                   "block_analysis": dpart.info},
        "roofline": roofline,
    }
    if config4 and world > 1:
        out["metric"] = ("Mcells*iters/s, config-4 step across ranks (exchange, impose_bc! FlowBC on the owned ghosts, "
                         "exchange, Euler HLL residual sweep on the image blocks), 3D sphere")
        out["config"]["step"] = dict(dist4, what="distributed.LocalDomain: local boundary chunks, donor cells beyond the "
                                                 "skirt as extra rows; two exchanges per step")
        # the point-implicit smoother across the ranks (point_implicit.py with distributed.RankOps): every residual sweep of
        # the Hutchinson estimate and of the Jacobian-vector products behind a skirt exchange, dots / norms / max all-reduced
        from ibamd import point_implicit as pi
        from ibamd.distributed import RankOps
        ops4 = RankOps(part.image_in_domain, bdom4.n_rows, hx4.exchange, device=u.device)
        Xc4 = torch.as_tensor(part.centers, device=P4.device)     # (a smooth state: see the one-rank branch below)
        wv4 = 1.0 + 1e-3 * torch.sin(Xc4[:, 0]) * torch.cos(Xc4[:, 1])
        P4[:dpart.nc, 0] = 1.0e5 * wv4
        P4[:dpart.nc, 1] = 288.15 * wv4
        P4[:dpart.nc, 2] = 100.0 * wv4
        P4[:dpart.nc, 3:] = 0.0
        hx4.exchange(P4)
        P40 = P4.clone()
        dtp = 1e-5

        def f_pi_local(X):
            r_ = ibamd.colmajor_empty(X.shape[0], nvp)
            r_[dpart.nc:] = 0.0
            H = ibamd.HipArray
            r_[:dpart.nc] = ((H(X[:dpart.nc]) - H(P40[:dpart.nc])) / dtp
                             - H(ibamd.residual_euler_hll(dpart, X[:dpart.nc], flags=flags))).t
            return r_
        f_pi = ops4.closure(f_pi_local)
        pi.linearize(f_pi, P4, 1, h=1e-2, seed=1 + rank)   # (untimed: the first call sizes the allocator's pools)
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        lin, bb, prec = pi.linearize(f_pi, P4, PI_SAMPLES, h=1e-2, seed=1 + rank)
        barrier()
        t1 = time.perf_counter()
        _, ratio = pi.solve(lin, bb, prec, n_iter=1, rtol=1e-9, reduce=ops4)
        barrier()
        t2 = time.perf_counter()
        out["config"]["step"]["point_implicit"] = {
            "linearize_ms": round((t1 - t0) * 1e3, 2), "solve_1_iteration_ms": round((t2 - t1) * 1e3, 2),
            "residual_ratio": round(float(ratio), 4),
            "what": f"pseudo-time step (P - P0)/dt - R(P) across the ranks: Hutchinson block estimate with {PI_SAMPLES} samples per "
                    f"variable ({5 * PI_SAMPLES + 1} exchanges + sweeps), one two-direction relaxation (2 exchanges + sweeps, 2 x 2 all-reduced "
                    "dot products, max |r| and the norm all-reduced)"}
    if config4 and world == 1:
        from ibamd import point_implicit as pi
        # the smoother is timed on a SMOOTH state (free stream + 1e-3 waves, no wall jump): on the 2 % white noise of the sweep
        # benchmark -- or behind an impulsively imposed slip wall, where the residual of the first ghost layer is ~1e8 -- the
        # Hutchinson blocks of a few of the 7.9 M cells are near-singular whatever the sample count, the preconditioned
        # direction drives their temperature negative in the finite-difference product and the relaxation returns NaN
        # (rounds 2-3 printed that NaN; scripts/diag_pi_nan.py).  The reference's smoother has no safeguard either.
        Xc = torch.as_tensor(part.centers, device=P.device)
        wv = 1.0 + 1e-3 * torch.sin(Xc[:, 0]) * torch.cos(Xc[:, 1])
        P[:, 0] = 1.0e5 * wv
        P[:, 1] = 288.15 * wv
        P[:, 2] = 100.0 * wv
        P[:, 3:] = 0.0
        P0 = P.clone()
        dtp = 1e-5

        def f_pi(X):   # (X - P0) / dt - R(X) as ONE broadcast launch (ibh_ew_eval), like a Julia `@.` line
            H = ibamd.HipArray
            return ((H(X) - H(P0)) / dtp - H(ibamd.residual_euler_hll(dpart, X))).t
        pi.linearize(f_pi, P, 1, h=1e-2, seed=1)            # (untimed: the first call sizes the allocator's pools)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lin, bb, prec = pi.linearize(f_pi, P, PI_SAMPLES, h=1e-2, seed=1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _, ratio = pi.solve(lin, bb, prec, n_iter=1, rtol=1e-9)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out["metric"] = "Mcells*iters/s, config-4 step (impose_bc! FlowBC ghosts + Euler HLL residual sweep), 3D sphere"
        out["config"]["step"] = {"ghost_cells": {k: int(sum(b.ghost_indices.size for b in v.values()))
                                                 for k, v in dom.boundaries.items()},
                                 "point_implicit": {"linearize_ms": round((t1 - t0) * 1e3, 2),
                                                    "solve_1_iteration_ms": round((t2 - t1) * 1e3, 2),
                                                    "residual_ratio": round(float(ratio), 4),
                                                    "residual_finite": bool(torch.isfinite(Rres).all().item()),
                                                    "what": "pseudo-time step (P - P0)/dt - R(P), Hutchinson block "
                                                            f"estimate with {PI_SAMPLES} samples per variable "
                                                            f"({5 * PI_SAMPLES + 1} sweeps), one two-direction relaxation "
                                                            "(2 sweeps)"}}
    if march:
        out["metric"] = "Mcells*steps/s, explicit march (device dt + sweep and update in one launch + BC set), 2D RAE2822"
        out["config"]["step"] = {"ghost_cells": int(bcs_m.n_ghost), "bc_set_levels": int(bcs_m.n_levels), "bc_set_levels_in_one_launch": int(bcs_m.n_direct_levels),
                                 "dt_reduction_every": max(1, args.dt_every),
                                 "dt_beside_the_bc_set": bool(args.dt_every <= 1 and not args.dt_separate),
                                 "finite": bool(torch.isfinite(um[0]).all().item()),
                                 "what": "test/advection.jl:61-89 without the host in the loop: ibh_timestep_advection every "
                                         f"{max(1, args.dt_every)} step(s), ibh_step_advection (k_sweep_quad storing u + dt ud, then the BC set) "
                                         "ping-pong between two arrays; boundaries: wall = 0, far field = copy(u)"}
    if config5:
        out["metric"] = ("Mcells*V-cycles/s, config-5 step (FAS! V-cycle, 3 levels x 2 smoothing iterations, impose_bc! on "
                         "every level + Euler HLL + viscous fluxes with eddy viscosity + Wray-Agarwal scalar), 3D sphere")
        if dist5 is not None:
            out["config"]["distributed"] = dist5
        out["config"]["step"] = {"levels_cells": [int(l.nc) for l in levels5],
                                 "residual_evaluations_per_step": 9,
                                 "what": "solver.jl:39-91 over multigrid() (ImmersedBoundary.jl:1355-1407); level closure = "
                                         "impose_bc! on the level's own boundaries (FlowBC free stream, slip wall with "
                                         "wall_function: cfd.jl:243-300, turbulence.jl:27-98), fused 3-D Euler sweep "
                                         "(face-list kernels on the coarse levels), viscous_fluxes(mu + mu_t) at operator "
                                         "granularity (cfd.jl:664-736), Wray-Agarwal transport (closures.py); one host sync "
                                         "per iteration for the convergence test" +
                                         ("" if world == 1 else "; across ranks: distributed.RankLevels (the rank's partition, "
                                          "local boundaries and transfer operators of every level), two exchanges per "
                                          "residual evaluation (before and after impose_bc!), all-reduced norms")}
        if world == 1:
            out["config"]["step"]["ghost_cells"] = ghosts5
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not euler and not is3d:
        cb = cpu_baseline(part, u_h, C_h)
        out["cpu_baseline"] = {"value": round(cb["value"], 3), "unit": "Mcells*iters/s", "cores": cb["threads"],
                               "kind": "port", "host_cores": os.cpu_count(), "affinity_cpus": cb["affinity"],
                               "cgroup_cpu_quota": cb["cgroup_cpus"], "threads_scanned": cb["scanned"],
                               "sample": f"{cb['n']} sweeps of the same {u_h.shape[0]}-cell partition in {cb['secs']:.1f} s: "
                                         f"C restatement of the Julia closure (oracle/csrc/residual.c), {cb['form']} form, "
                                         f"OpenMP on {cb['threads']} threads of a {os.cpu_count()}-core host "
                                         f"({'best of the thread counts ' + str(cb['scanned']) if len(cb['scanned']) > 1 else 'the only thread count this job is allowed'}; "
                                         f"OMP_NUM_THREADS={cb['omp_num_threads_env']}); "
                                         f"{cb['other_form']} form {cb['other_value']:.1f} on {cb['other_threads']} threads, "
                                         f"faithful form on 1 thread {cb['one_thread']:.1f} Mcells*iters/s"}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
