"""The one JSON line of bench.py against the contract the driver reads (keys, types, units), checked on the committed lines of
the last profile set (profiles/r3_final/: produced by `python bench.py ...` on an MI355X, see its README) -- a regression
guard for the fields, not a measurement."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles", "r3_final")


def _line(name):
    with open(os.path.join(PROF, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_default_line_has_the_contract_fields():
    d = _line("bench.json")
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md publishes no number for this metric
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["unit"] == "Mcells*iters/s" and "workload" in d["config"] and "model" not in d["config"]
    # value = cells x steps / time: consistent with ms_per_step
    cells = d["config"]["cells_total"]
    assert abs(d["value"] - cells / (d["ms_per_step"] * 1e-3) / 1e6) <= 2e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3
    assert r["traffic"] is None or r["traffic"] > 16 * cells        # HBM-side bytes per launch >= the algorithmic 16 B/cell
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == d["unit"] and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert c["cores"] in c["threads_scanned"]


@pytest.mark.parametrize("name,kernel,bytes_per_cell", [
    ("bench_euler2d.json", "k_sweep_quad_euler", 32.0), ("bench_3d_4.6M.json", "k_sweep3_cols", 20.0),
    ("bench_3d_euler_33M.json", "k_sweep3_euler_cols", 40.0), ("bench_28M.json", "k_sweep_quad", 16.0)])
def test_secondary_lines_price_the_right_bytes(name, kernel, bytes_per_cell):
    d = _line(name)
    r = d["roofline"]
    assert r["kernel"] == kernel and r["alg_bytes_per_cell"] == bytes_per_cell
    assert abs(r["achieved"] - bytes_per_cell * r["cells_per_launch"] / (r["kernel_us"] * 1e-6) / 1e9) <= 2e-3 * r["achieved"]
    assert 0.0 < r["frac"] < 1.0
