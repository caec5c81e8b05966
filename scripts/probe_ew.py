"""ibh_ew_eval: the four-elements-per-thread interpreter against the one-element one (ibh_set_tuning "ew_scalar"), on the two
broadcast lines of the config-4 / config-5 closures.  Run on the GPU box: python scripts/probe_ew.py [cells]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ibamd  # noqa: E402
from ibamd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 7917568
H = ibamd.HipArray
rng = np.random.default_rng(0)
X, P0, R = (ibamd.hip(rng.uniform(0.5, 2.0, (n, 5)).astype(np.float32)) for _ in range(3))
p, T, nut = (ibamd.hip(rng.uniform(0.5, 2.0, n).astype(np.float32)) for _ in range(3))
lines = {"pseudo_time_residual_5_columns": lambda: ((H(X) - H(P0)) / 1e-5 - H(R)).t,
         "mu_t_from_p_T_nu_t": lambda: (H(p) / (H(T) * 287.0) * H(nut)).t}
out = {"cells": n}
for name, f in lines.items():
    for key, var in (("four_per_thread_us", 0), ("one_per_thread_us", 1)):
        _lib.call("ibh_set_tuning", b"ew_scalar", var)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        out[name + "_" + key] = round(e0.elapsed_time(e1) * 1e3 / 20, 1)
_lib.call("ibh_set_tuning", b"ew_scalar", 0)
print(json.dumps(out))
