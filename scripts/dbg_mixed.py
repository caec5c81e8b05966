import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ibamd
from conftest import rae_mesh, RAE_FAMILIES, seeded_field
f32 = np.float32
dom = ibamd.Domain(rae_mesh(), hypercube_families=RAE_FAMILIES, max_partition_size=16384, boundaries=False)
for k, part in dom.partitions.items():
    dpart = ibamd.to_backend(part, ibamd.hip)
    u = seeded_field(part.centers, kind="step")
    C = np.stack([np.ones_like(u), f32(0.5) + seeded_field(part.centers, seed=3) * f32(0.1)], axis=1)
    one = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), flags=ibamd.IBH_FORCE_MIXED))
    two = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), flags=ibamd.IBH_NO_FUSE))
    d = np.abs(one - two)
    bad = np.nonzero(d > 1e-4 * np.abs(two).max())[0]
    print("partition", k, dpart.info, "bad cells", bad.size, "max", d.max(), "scale", np.abs(two).max())
    if bad.size:
        img = np.zeros(u.shape[0], bool); img[part.image_in_domain] = True
        g = part.domain[bad]
        print("  bad are image:", img[bad].sum(), "global block ids:", np.unique(g // 64)[:20], "pos in block:", np.unique(g % 64)[:64])
