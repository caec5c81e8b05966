"""profiles/current_pmc.json from a profile set: python scripts/refresh_current_pmc.py profiles/r3_final
One entry per (workload, kernel): HBM-side traffic (FETCH_SIZE, WRITE_SIZE in KB per launch; FETCH_SIZE counts 32-byte units
on gfx950: x2), kernel-trace average, VALU / LDS instructions per wave, VALU busy and memory-wait fractions -- what bench.py
attaches to a line whose workload and kernel match."""
import csv
import json
import os
import sys

P = sys.argv[1].rstrip("/") + "/"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def entry(tag, workload, kernel, match):
    s = json.load(open(P + "pmc_summary_%s.json" % tag))
    k = [x for x in s if x.startswith(match)][0]
    d = s[k]
    waves = d["SQ_WAVES"]
    avg = [float(r["AverageNs"]) for r in csv.DictReader(open(P + "kernel_stats_%s.csv" % tag)) if match in r["Name"]][0]
    return {"workload": workload, "kernel": kernel, "fetch_kb": d["FETCH_SIZE"], "write_kb": d["WRITE_SIZE"],
            "source": "%spmc_summary_%s.json + kernel_stats_%s.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ groups in "
                      "separate passes, per-launch averages, KB; FETCH_SIZE x2 on gfx950)" % (P, tag, tag),
            "kernel_trace_avg_us": round(avg / 1e3, 2),
            "valu_insts_per_wave": round(d["SQ_INSTS_VALU"] / waves, 1), "lds_insts_per_wave": round(d["SQ_INSTS_LDS"] / waves, 1),
            "valu_busy_frac": round(d["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * avg * 2.4), 3),
            "wave_wait_frac": round(d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 3)}


E = [entry("headline", "rae2822_0.87M", "k_sweep_quad", "k_sweep_quad<"),
     entry("euler2d", "rae2822_0.87M", "k_sweep_quad_euler", "k_sweep_quad_euler"),
     entry("3.47M", "rae2822_3.47M", "k_sweep_quad", "k_sweep_quad<"),
     entry("3d_4.6M", "sphere3d_4.6M", "k_sweep3_cols", "k_sweep3_cols"),
     entry("3d_euler_4.6M", "sphere3d_4.6M", "k_sweep3_euler_cols", "k_sweep3_euler_cols")]
# large lines: traffic passes only (no SQ groups, no kernel trace)
for tag, workload, kernel, match in (("28M", "rae2822_28M", "k_sweep_quad", "k_sweep_quad<"),
                                     ("3d_33M", "sphere3d_33M", "k_sweep3_cols", "k_sweep3_cols"),
                                     ("3d_euler_33M", "sphere3d_33M", "k_sweep3_euler_cols", "k_sweep3_euler_cols")):
    f = P + "pmc_summary_%s.json" % tag
    if os.path.exists(f):
        d = json.load(open(f))
        k = [x for x in d if x.startswith(match)][0]
        E.append({"workload": workload, "kernel": kernel, "fetch_kb": d[k]["FETCH_SIZE"], "write_kb": d[k]["WRITE_SIZE"],
                  "source": "%spmc_summary_%s.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, per-launch "
                            "averages, KB; FETCH_SIZE x2 on gfx950)" % (P, tag)})
for e in E:
    if "kernel_trace_avg_us" not in e:
        print(e["workload"], e["kernel"], round((2 * e["fetch_kb"] + e["write_kb"]) * 1024 / 1e6, 1), "MB (1e6 bytes)")
        continue
    print(e["workload"], e["kernel"], e["kernel_trace_avg_us"], "us", round((2 * e["fetch_kb"] + e["write_kb"]) * 1024 / 1e6, 1), "MB (1e6 bytes)",
          e["valu_insts_per_wave"], e["valu_busy_frac"], e["wave_wait_frac"])
json.dump({"entries": E}, open(os.path.join(ROOT, "profiles", "current_pmc.json"), "w"), indent=1)
