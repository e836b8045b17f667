#!/usr/bin/env python3
"""Summarises gpurun_out/prof_<tag>/ (made by scripts/profile_gpu.sh) into profiles/<tag>_summary.json/.md
and copies the rocprofv3 kernel_stats.csv next to it."""
import os
import collections, csv, glob, json, os, shutil, statistics as st, sys

tag = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "step_kernel"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
out = {"tag": tag, "kernel_filter": kernel}
ks = sorted(glob.glob(f"{src}/trace/*/*_kernel_stats.csv"), key=os.path.getmtime, reverse=True)  # newest run of the tag
if ks:
    shutil.copy(ks[0], f"profiles/{tag}_kernel_stats.csv")
    for r in csv.DictReader(open(ks[0])):
        if kernel in r["Name"]:
            out["kernel"] = r["Name"]
            out["calls"] = int(r["Calls"])
            out["avg_ns"] = float(r["AverageNs"])
            out["min_ns"] = float(r["MinNs"])
            out["max_ns"] = float(r["MaxNs"])
            out["percent_of_gpu_time"] = float(r["Percentage"])
counters = {}
for p in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
    # a tag that was profiled twice keeps both runs' files after gpurun merges them back: use the newest
    fs = sorted(glob.glob(f"{src}/{p}/*/*_counter_collection.csv"), key=os.path.getmtime, reverse=True)
    if not fs:
        continue
    d = collections.defaultdict(list)
    meta = None
    for r in csv.DictReader(open(fs[0])):
        if kernel in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = r
    for k, v in d.items():
        counters[k] = st.median(v)
    if meta:
        out["grid_size"] = int(meta["Grid_Size"])
        out["workgroup_size"] = int(meta["Workgroup_Size"])
        out["lds_block_size"] = int(meta["LDS_Block_Size"])
        out["vgpr_count_field"] = int(meta["VGPR_Count"])
        out["sgpr_count_field"] = int(meta["SGPR_Count"])
out["counters_per_launch_median"] = counters
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the
    # bytes of coalesced streaming reads -> doubled; WRITE_SIZE is exact for streaming stores.
    rd = counters["FETCH_SIZE"] * 1024 * 2
    wr = counters["WRITE_SIZE"] * 1024
    out["hbm_read_bytes_per_launch_corrected"] = rd
    out["hbm_write_bytes_per_launch"] = wr
    out["hbm_bytes_per_launch"] = rd + wr
waves = counters.get("SQ_WAVES")
if waves:
    out["per_wave"] = {k: v / waves for k, v in counters.items() if k.startswith("SQ_")}
    if "hbm_bytes_per_launch" in out:
        out["hbm_bytes_per_env_step"] = out["hbm_bytes_per_launch"] / waves
bl = f"{src}/bench_line.json"
if os.path.exists(bl) and os.path.getsize(bl):
    try:
        out["bench_line_under_profiler"] = json.loads(open(bl).read())
    except Exception:
        pass
json.dump(out, open(f"profiles/{tag}_summary.json", "w"), indent=1)
if "hbm_bytes_per_launch" in out and waves:
    b = out.get("bench_line_under_profiler", {}).get("config", {})
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash, step_kernel_isa_hash
    json.dump({"envs": int(waves), "board": b.get("board", [20, 20, 4]), "hbm_bytes_per_launch": out["hbm_bytes_per_launch"],
               "kernel_source_hash": kernel_source_hash(), "step_kernel_isa_hash": step_kernel_isa_hash(*b.get("board", [20, 20, 4])),
               "source": f"profiles/{tag}_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 per MI355X_MICROARCH.md)"},
              open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
