#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_case.sh into profiles/<tag>/summary_<case>.json and merge the counter
figures into profiles/pmc_latest.json (keyed by bench.py case name).     python tools/summarize_profile.py <tag> <case>

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by exactly 2x
(MI355X_MICROARCH.md, HBM section), so read bytes are reported both raw and x2-corrected.
"""
import csv, glob, json, os, statistics, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
case = sys.argv[2] if len(sys.argv) > 2 else "reach-joint"   # a bench.py case name
task = case.split("-")[0]                                     # reach | pnp
ctrl_id = {"joint": 0, "IK": 1, "mocap": 2}[case.split("-")[1]]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "gpurun_out", tag, case)
KERNEL = ("step_reach_kernel<%d" if task == "reach" else "step_pnp_kernel<%d") % ctrl_id      # any variant (<0, true> = multi-wave)
LANES = 64 if task == "reach" else 32           # active lanes per wave (PNP_LANES)
ALGO_BYTES = 939 if task == "reach" else 1363


def rows(pattern):
    out = []
    for f in glob.glob(os.path.join(src, pattern)):
        out += list(csv.DictReader(open(f)))
    return out


summary = {"tag": tag, "kernel": KERNEL}
import re
names = sorted({m.group(0) for r in rows("stats/*/*_kernel_trace.csv") for m in [re.search(r"step_\w+<[^>]*>", r["Kernel_Name"])] if m and KERNEL in r["Kernel_Name"]})
summary["kernel_variants_seen"] = names
kt = [r for r in rows("stats/*/*_kernel_trace.csv") if KERNEL in r["Kernel_Name"]]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in kt]
summary["kernel_trace"] = {"launches": len(dur), "avg_us": statistics.mean(dur), "median_us": statistics.median(dur),
                           "min_us": min(dur), "max_us": max(dur), "vgpr": int(kt[0]["VGPR_Count"]),
                           "agpr": int(kt[0]["Accum_VGPR_Count"]), "sgpr": int(kt[0]["SGPR_Count"]),
                           "scratch_bytes_per_lane": int(kt[0]["Scratch_Size"]), "lds_bytes_per_wg": int(kt[0]["LDS_Block_Size"]),
                           "grid": int(kt[0]["Grid_Size_X"]), "workgroup": int(kt[0]["Workgroup_Size_X"])}
stats = [r for r in rows("stats/*/*_kernel_stats.csv")]
summary["kernel_stats_top"] = [{"name": r["Name"][:90], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                "pct": float(r["Percentage"])} for r in stats[:4]]
counters = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_flops"):
    for r in rows(f"{d}/*/*_counter_collection.csv"):
        if KERNEL in r["Kernel_Name"]:
            counters.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
summary["counters_avg_per_launch"] = {k: statistics.mean(v) for k, v in counters.items()}
c = summary["counters_avg_per_launch"]
n_envs = summary["kernel_trace"]["grid"] * (64 if task == "reach" else 32) // summary["kernel_trace"]["workgroup"]   # two-wave: 128 threads per 64 envs
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    rd_raw, wr = c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
    summary["hbm"] = {"read_bytes_raw": rd_raw, "read_bytes_x2_corrected": 2 * rd_raw, "write_bytes": wr,
                      "bytes_per_launch_corrected": 2 * rd_raw + wr, "bytes_per_env_step_corrected": (2 * rd_raw + wr) / n_envs,
                      "algorithmic_bytes_per_env_step": ALGO_BYTES}
if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
    waves = c["SQ_WAVES"]
    summary["per_wave"] = {k: c[k] / waves for k in c if k.startswith("SQ_") and k != "SQ_WAVES"}
    summary["per_wave"]["waves"] = waves
flops = None
if "SQ_INSTS_VALU_FMA_F64" in c:
    # wave-level instruction counts -> FLOPs over the active lanes (FMA = 2); transcendental F64 (rcp, sqrt) counted as 1
    flops = LANES * (2 * c["SQ_INSTS_VALU_FMA_F64"] + c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0)
                     + c.get("SQ_INSTS_VALU_TRANS_F64", 0))
    summary["f64"] = {"flops_per_launch": flops, "flops_per_env_step": flops / n_envs,
                      "fma": c["SQ_INSTS_VALU_FMA_F64"], "add": c.get("SQ_INSTS_VALU_ADD_F64", 0),
                      "mul": c.get("SQ_INSTS_VALU_MUL_F64", 0), "trans": c.get("SQ_INSTS_VALU_TRANS_F64", 0)}
try:
    summary["bench_line"] = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
except Exception:
    pass
os.makedirs(os.path.join(root, "profiles", tag), exist_ok=True)
json.dump(summary, open(os.path.join(root, "profiles", tag, f"summary_{case}.json"), "w"), indent=1)
for f in glob.glob(os.path.join(src, "stats/*/*_kernel_stats.csv")):
    import shutil; shutil.copy(f, os.path.join(root, "profiles", tag, f"kernel_stats_{case}.csv"))
if "hbm" in summary:
    path = os.path.join(root, "profiles", "pmc_latest.json")
    allc = {"cases": {}}
    if os.path.exists(path):
        old = json.load(open(path))
        if "cases" in old: allc = old
    try:
        sha = open(os.path.join(src, "src_sha256.txt")).read().strip()          # written on the GPU box by tools/profile_case.sh
    except OSError:
        sha = None
    allc["cases"][case] = {"n_envs": n_envs, "src_sha256": sha, "hbm_bytes_per_launch": summary["hbm"]["bytes_per_launch_corrected"],
                           "f64_flops_per_launch": flops, "kernel_avg_us": summary["kernel_trace"]["avg_us"],
                           "sq_wait_any_frac": (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]) if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c else None,
                           "scratch_bytes_per_lane": summary["kernel_trace"]["scratch_bytes_per_lane"],
                           "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2 (gfx950), tag {tag}"}
    json.dump(allc, open(path, "w"), indent=1)
print(json.dumps(summary, indent=1))
