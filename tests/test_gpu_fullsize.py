"""Parity at the benchmark's full size (BASELINE.json configs[1]: 2-D RAE2822, 867 904 cells, one partition).

The oracle here is the C restatement (oracle/csrc/residual.c: bit-identical to the numpy restatement on the small
cases, tests/test_oracle_c.py), which does the whole mesh in a fraction of a second; plus size-independent
properties of the closure: constant fields give exactly zero, the residual is positively homogeneous of degree one
(minmod and the JST ratio are scale-invariant up to the 1e-7 regularisation), and all arithmetic variants of the
library agree with each other.  Tolerance 1e-5 norm-wise (north_star)."""
import os
import sys

import numpy as np
import pytest

import ibamd
from conftest import rel_inf

pytestmark = pytest.mark.gpu
f32 = np.float32
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.fixture(scope="module")
def full():
    import bench
    msh = bench.build_mesh("rae2822_0.87M")
    assert len(msh) == 867904
    dom = ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False)
    (part,) = dom.partitions.values()
    u, C = bench.synthetic_fields(part.centers)
    return part, ibamd.to_backend(part, ibamd.hip), u, C


def test_full_size_advection_sweep_against_the_c_oracle(full):
    from oracle import residual_c as rc
    part, dpart, u, C = full
    assert dpart.info["fusable_blocks"] == dpart.info["full_blocks"] == 13561
    Cv = np.stack([C[:, 0], f32(0.5) * C[:, 1] + f32(0.2) * np.sin(part.centers[:, 0]).astype(f32)], axis=1)
    exp = rc.CPart(part).residual_advection(u, Cv)
    du, dC = ibamd.hip(u), ibamd.hip(Cv)
    one = ibamd.to_host(ibamd.residual_advection(dpart, du, dC))                              # single kernel
    two = ibamd.to_host(ibamd.residual_advection(dpart, du, dC, flags=ibamd.IBH_NO_FUSE))     # two kernels
    lit = ibamd.to_host(ibamd.residual_advection(dpart, du, dC, flags=ibamd.IBH_EXACT))       # literal block path
    gen = ibamd.to_host(ibamd.residual_advection(dpart, du, dC, flags=ibamd.IBH_FORCE_GENERAL))  # face lists
    assert np.array_equal(gen, exp)            # the literal arithmetic reproduces the oracle bit for bit
    assert np.array_equal(lit, gen)
    assert rel_inf(two, exp) <= 1e-5
    assert rel_inf(one, exp) <= 1e-5
    assert rel_inf(one, two) <= 2e-6


def test_full_size_properties(full):
    part, dpart, u, C = full
    dC = ibamd.hip(C)
    n = u.shape[0]
    const = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(np.full(n, 3.25, dtype=f32)), dC))
    assert np.array_equal(const, np.zeros(n, dtype=f32))                  # constant field: exactly zero
    r1 = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), dC))
    r4 = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(f32(4.0) * u), dC))
    assert rel_inf(r4, f32(4.0) * r1) <= 1e-5                             # positively homogeneous, degree one
    rm = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(f32(-1.0) * C)))
    rf = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(-u), ibamd.hip(f32(-1.0) * C)))
    assert rel_inf(rf, -rm) <= 1e-5                                       # odd in u for a fixed velocity field


def test_full_size_euler_sweep_variants_agree(full):
    import bench
    part, dpart, _, _ = full
    rng = np.random.default_rng(12345)
    n = part.centers.shape[0]
    P = np.empty((n, 4), dtype=f32)
    P[:, 0] = 1e5 * (1 + 0.05 * rng.uniform(-1, 1, n))
    P[:, 1] = 288.15 * (1 + 0.05 * rng.uniform(-1, 1, n))
    P[:, 2] = 100.0 * (1 + 0.1 * rng.uniform(-1, 1, n))
    P[:, 3] = 100.0 * (1 + 0.1 * rng.uniform(-1, 1, n))
    dP = ibamd.hip(P)
    one = ibamd.to_host(ibamd.residual_euler_hll(dpart, dP))
    two = ibamd.to_host(ibamd.residual_euler_hll(dpart, dP, flags=ibamd.IBH_NO_FUSE))
    blk = ibamd.to_host(ibamd.residual_euler_hll(dpart, dP, flags=ibamd.IBH_NO_QUAD))      # per-block single kernel
    gen = ibamd.to_host(ibamd.residual_euler_hll(dpart, dP, flags=ibamd.IBH_FORCE_GENERAL))  # literal, Float64 HLL
    from oracle import residual_c as rc
    exp = rc.CPart(part).residual_euler(P)          # the C restatement on the whole mesh
    assert np.array_equal(gen, exp)                 # literal arithmetic: bit for bit, Float64 HLL combine included
    for v in range(4):
        assert rel_inf(two[:, v], exp[:, v]) <= 1e-5, v
        assert rel_inf(one[:, v], exp[:, v]) <= 1e-5, v
        assert rel_inf(blk[:, v], exp[:, v]) <= 1e-5, v


def test_config3_partitions_image_only_sweep():
    """BASELINE.json configs[2] on one GPU: the 3.47 M-cell RAE2822 mesh cut in 8 block-aligned partitions
    (ImmersedBoundary.jl:594-621); the image-only sweep (what a rank of the 8-GPU run computes) of the first and of an
    interior partition against the C restatement of the closure on `part.domain`."""
    import torch
    import bench
    from oracle import residual_c as rc
    msh = bench.build_mesh("rae2822_3.47M")
    ncells, world = len(msh), 8
    assert ncells == 3469888
    mps = -(-(-(-ncells // world)) // 64) * 64
    dom = ibamd.Domain(msh, max_partition_size=mps, boundaries=False, only=[1, 4])
    for k in (1, 4):
        part = dom.partitions[k]
        img = part.image_in_domain
        assert img.size == mps and part.spacing.shape[0] > img.size     # skirt cells are there
        dpart = ibamd.to_backend(part, ibamd.hip)
        assert dpart.info["image_blocks_all_eligible"] and dpart.info["image_blocks"] * 64 == img.size
        assert dpart.info["image_quads"] > 0
        u, C = bench.synthetic_fields(part.centers)
        exp = rc.CPart(part).residual_advection(u, C)
        out = torch.full((u.shape[0],), float("nan"), dtype=torch.float32, device="cuda")
        ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), out=out, flags=ibamd.IBH_IMAGE_ONLY)
        got = ibamd.to_host(out)
        assert np.isnan(got).sum() == u.shape[0] - img.size              # nothing written outside the image
        assert rel_inf(got[img], exp[img]) <= 1e-5
        per_block = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C),
                                                           flags=ibamd.IBH_IMAGE_ONLY | ibamd.IBH_NO_QUAD))
        assert rel_inf(got[img], per_block[img]) <= 2e-6
        if k == 4:   # the Euler sweep of the same rank: quad form (quad2::sweep_quad_euler) over the image quads
            rng = np.random.default_rng(7)
            n = u.shape[0]
            P = np.stack([1e5 * (1 + 0.05 * rng.uniform(-1, 1, n)), 288.15 * (1 + 0.05 * rng.uniform(-1, 1, n)),
                          100.0 * (1 + 0.1 * rng.uniform(-1, 1, n)), 100.0 * (1 + 0.1 * rng.uniform(-1, 1, n))],
                         axis=1).astype(f32)
            expE = rc.CPart(part).residual_euler(P)
            outE = torch.full((4, n), float("nan"), dtype=torch.float32, device="cuda").T          # column-major
            ibamd.residual_euler_hll(dpart, ibamd.hip(P), out=outE, flags=ibamd.IBH_IMAGE_ONLY)
            gotE = ibamd.to_host(outE)
            blkE = ibamd.to_host(ibamd.residual_euler_hll(dpart, ibamd.hip(P),
                                                          flags=ibamd.IBH_IMAGE_ONLY | ibamd.IBH_NO_QUAD))
            assert np.isnan(gotE[:, 0]).sum() == n - img.size
            for v in range(4):
                assert rel_inf(gotE[img, v], expE[img, v]) <= 1e-5, v
                assert rel_inf(gotE[img, v], blkE[img, v]) <= 5e-6, v


def test_full_size_per_cell_error_percentiles(full, capsys):
    """The 1e-5 tolerance above is norm-wise (max |d| / max |ref|).  Once, at full size: the PER-CELL picture of the
    tuned sweep against the literal arithmetic -- error relative to the local scale |ref| + |u| / h (the size of the
    terms the residual is a difference of), percentiles over the 867 904 cells -- and the worst cells in absolute terms.
    (Printed with -s; the figures of the committed build are in DESIGN.md section 5.)"""
    import json
    from oracle import residual_c as rc
    part, dpart, u, C = full
    exp = rc.CPart(part).residual_advection(u, C).astype(np.float64)
    got = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C))).astype(np.float64)
    from conftest import stencil_scale
    rel = np.abs(got - exp) / stencil_scale(part, u, exp)
    pct = {f"p{p}": float(np.percentile(rel, p)) for p in (50, 90, 99, 99.9)}
    pct["max"] = float(rel.max())
    pct["norm_wise"] = float(np.abs(got - exp).max() / np.abs(exp).max())
    with capsys.disabled():
        print("\nper-cell relative error of the tuned sweep (0.87 M cells):", json.dumps(pct))
    # (measured: max 5.9e-7, p99.9 3.4e-7; the same bound as the 3-D sweeps below)
    assert pct["p99.9"] <= 1e-6 and pct["max"] <= 2e-6 and pct["norm_wise"] <= 1e-5


def test_3d_single_kernel_sweeps_at_bench_size():
    """The 3-D workloads of bench.py at full size (sphere-in-box octree, 1 667 072 cells, every kind of block side): the
    single-kernel sweeps -- column forms (scalar and Euler: one wavefront per block) -- against the C restatement on the whole mesh,
    against the thread-per-cell and the two-kernel forms, and the size-independent properties of the closure."""
    import bench
    from ibamd import _lib
    from oracle import residual_c as rc
    msh = bench.build_mesh("sphere3d_1.6M")
    assert len(msh) == 1667072
    dom = ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False)
    (part,) = dom.partitions.values()
    dpart = ibamd.to_backend(part, ibamd.hip)
    info = dpart.info
    assert info["fusable_blocks"] == info["full_blocks"] == 3256 and info["sides_fine"] > 0 and info["rim4_rows"] > 0
    u, C = bench.synthetic_fields(part.centers)
    n = u.shape[0]
    cpart = rc.CPart(part)
    exp = cpart.residual_advection(u, C)
    du, dC = ibamd.hip(u), ibamd.hip(C)
    strip = ibamd.to_host(ibamd.residual_advection(dpart, du, dC))
    _lib.call("ibh_set_tuning", b"quad_variant", 512)            # thread-per-cell single kernel
    try:
        cellk = ibamd.to_host(ibamd.residual_advection(dpart, du, dC))
    finally:
        _lib.call("ibh_set_tuning", b"quad_variant", 0)
    two = ibamd.to_host(ibamd.residual_advection(dpart, du, dC, flags=ibamd.IBH_NO_FUSE))
    gen = ibamd.to_host(ibamd.residual_advection(dpart, du, dC, flags=ibamd.IBH_FORCE_GENERAL))
    assert np.array_equal(gen, exp)                      # literal arithmetic: bit for bit
    for got in (strip, cellk, two):
        assert rel_inf(got, exp) <= 1e-5
    assert rel_inf(strip, cellk) <= 2e-6 and rel_inf(strip, two) <= 2e-6
    # per cell (as in the 2-D test above): error relative to the local scale |ref| + |u| / h
    from conftest import stencil_scale
    e64, s64 = exp.astype(np.float64), strip.astype(np.float64)
    rel = np.abs(s64 - e64) / stencil_scale(part, u, e64)
    pct = {f"p{q}": float(np.percentile(rel, q)) for q in (50, 99, 99.9)}
    pct["max"] = float(rel.max())
    print("\nper-cell relative error of the 3-D column sweep (1.67 M cells):", pct)
    # (measured: max 7.7e-7, p99.9 3.7e-7.  Round 2 divided by |ref| + |u| / h of the cell alone and needed 5e-5: its worst
    # cells are zero crossings of u, not sensor effects -- conftest.stencil_scale)
    assert pct["p99.9"] <= 1e-6 and pct["max"] <= 2e-6
    const = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(np.full(n, 3.25, dtype=f32)), dC))
    assert np.array_equal(const, np.zeros(n, dtype=f32))            # constant field: exactly zero
    r4 = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(f32(4.0) * u), dC))
    assert rel_inf(r4, f32(4.0) * strip) <= 1e-5                    # positively homogeneous, degree one
    rng = np.random.default_rng(5)
    P = np.stack([1e5 * (1 + 0.05 * rng.uniform(-1, 1, n)), 288.15 * (1 + 0.05 * rng.uniform(-1, 1, n)),
                  100.0 * (1 + 0.1 * rng.uniform(-1, 1, n)), 60.0 * (1 + 0.1 * rng.uniform(-1, 1, n)),
                  -40.0 * (1 + 0.1 * rng.uniform(-1, 1, n))], axis=1).astype(f32)
    expE = cpart.residual_euler(P)
    one = ibamd.to_host(ibamd.residual_euler_hll(dpart, ibamd.hip(P)))
    twoE = ibamd.to_host(ibamd.residual_euler_hll(dpart, ibamd.hip(P), flags=ibamd.IBH_NO_FUSE))
    for v in range(5):
        assert rel_inf(one[:, v], expE[:, v]) <= 1e-5, v
        assert rel_inf(twoE[:, v], expE[:, v]) <= 1e-5, v
        assert rel_inf(one[:, v], twoE[:, v]) <= 5e-6, v
