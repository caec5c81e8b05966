"""The product's mesher (explicit stack, vectorised cell enumeration) against the independent recursive restatement of
``refine_octree`` and ``get_cells`` in oracle/mesher.py (/root/reference/src/mesher.jl:811-862, 1064-1112): block origins,
block widths (= levels) and the cell order, bit for bit, on the advection mesh, the RAE2822 mesh of the tests and a small
3-D sphere octree; plus analytic criteria that involve no code of the product at all."""
import numpy as np

import ibamd
from ibamd import mesher as pm
from oracle import mesher as om

f32 = np.float32


def _compare(msh):
    ows = om.refine_octree(msh.ref_regions, msh.origin, msh.widths, msh.growth_ratio)
    bo = np.stack([t[0] for t in ows], axis=1)
    bw = np.stack([t[1] for t in ows], axis=1)
    assert bo.dtype == np.float32 and np.array_equal(bo, msh.block_origins)      # same leaves, same order
    assert np.array_equal(bw, msh.block_widths)                                  # same levels
    c_o, w_o = om.get_cells(bo, bw, msh.block_size)
    c_p, w_p = pm.get_cells(msh)
    assert np.array_equal(c_o, c_p) and np.array_equal(w_o, w_p)                 # same cells in the same order
    return bo.shape[1]


def test_advection_mesh(adv_mesh):
    assert _compare(adv_mesh) == adv_mesh.nblocks > 100


def test_rae2822_mesh(rae_mesh_small):
    assert _compare(rae_mesh_small) == rae_mesh_small.nblocks > 400


def test_sphere_octree():
    import bench
    msh = pm.Mesh(f32([-4, -4, -4]), f32([8, 8, 8]), ("sphere", bench.icosphere(subdiv=1), f32(0.4)), block_size=8)
    assert msh.ndims == 3 and _compare(msh) == msh.nblocks > 30


def test_analytic_criteria_and_anisotropic_root():
    """Criteria written here (a ball, a half-plane), a 2:1 root box (split sizes 3 x 2, mesher.jl:838-842) and a growth
    ratio other than 2: nothing of the product but refine_octree itself is involved."""
    def ball(c):
        return f32(max(np.sqrt(float((c[0] - f32(0.3)) ** 2 + (c[1] - f32(0.1)) ** 2)) - 0.2, 0.0))

    def wall(c):
        return f32(abs(float(c[1]) + 0.45))
    crit = [(ball, 0.02), (wall, 0.05)]
    for gr in (f32(2.0), 1.5):
        a = pm.refine_octree(crit, f32([-1, -0.5]), f32([2, 1]), gr)
        b = om.refine_octree(crit, f32([-1, -0.5]), f32([2, 1]), gr)
        assert len(a) == len(b) > 50
        for (oa, wa), (ob, wb) in zip(a, b):
            assert np.array_equal(oa, ob) and np.array_equal(wa, wb)
