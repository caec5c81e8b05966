"""How many quads (2 x 2 leaf blocks = 16 x 16 cells, the work unit of k_sweep_quad) lie in complete SUPER-TILES -- the
four sibling quads of a quadtree node, 32 x 32 cells at one level, 16 consecutive blocks in the depth-first order -- on
the benchmark meshes.  The round-3 review asked for this count before building a 256-thread super-tile kernel for the
headline sweep (go on only at >= 60 %).  Geometry only (an upper bound: the side classes are not checked).
    python scripts/count_supertiles.py [workload ...]      (CPU; prints one JSON line per mesh)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def count(name):
    msh = bench.build_mesh(name)
    O = msh.block_origins.T.astype(np.float64)
    W = msh.block_widths.T.astype(np.float64)
    nb = O.shape[0]
    org = msh.origin.astype(np.float64)

    def group(start, k):   # k * k consecutive equal blocks tiling an aligned k x k square in z-order
        if start + k * k > nb:
            return False
        w = W[start, 0]
        if not np.all(W[start:start + k * k] == w):
            return False
        rel = (O[start] - org) / (k * w)
        if np.abs(rel - np.round(rel)).max() > 1e-6:
            return False
        i = np.arange(k * k)
        ex = np.stack([i & 1, i >> 1], 1) if k == 2 else \
            np.stack([(i & 1) | ((i >> 2) & 1) << 1, ((i >> 1) & 1) | ((i >> 3) & 1) << 1], 1)
        return np.abs((O[start:start + k * k] - O[start]) / w - ex).max() < 1e-6

    def scan(k):
        i = n = 0
        while i < nb:
            if group(i, k):
                n += 1
                i += k * k
            else:
                i += 1
        return n
    quads, supers = scan(2), scan(4)
    return {"workload": name, "blocks": nb, "quads": quads, "blocks_in_quads": round(4 * quads / nb, 3),
            "super_tiles": supers, "blocks_in_super_tiles": round(16 * supers / nb, 3),
            "quads_in_super_tiles": round(4 * supers / max(quads, 1), 3)}


for w in (sys.argv[1:] or ["rae2822_0.87M", "rae2822_3.47M"]):
    print(json.dumps(count(w)))
