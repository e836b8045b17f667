#!/usr/bin/env python3
"""Summarises scripts/rccl_tax.sh: per variant the step kernel's duration by the profiler's own timestamps, the gap
between the end of one step launch and the start of the next, every OTHER kernel that ran, and the bench line."""
import csv, glob, json, statistics as st, sys

out = {}
for var in ("plain", "rccl", "gloo"):
    d = f"{sys.argv[1]}/{var}"
    tr = sorted(glob.glob(f"{d}/*/*_kernel_trace.csv"))
    rec = {}
    if tr:
        rows = list(csv.DictReader(open(tr[-1])))
        steps = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "step_kernel" in r["Kernel_Name"]))
        steps = steps[len(steps) // 2:]            # the second half: clocks settled, the timed region
        dur = [e - s for s, e in steps]
        gap = [steps[i + 1][0] - steps[i][1] for i in range(len(steps) - 1)]
        rec["step_launches_used"] = len(steps)
        rec["step_dur_ns"] = {"mean": st.mean(dur), "median": st.median(dur), "p10": sorted(dur)[len(dur) // 10], "p90": sorted(dur)[len(dur) * 9 // 10]}
        rec["gap_ns"] = {"mean": st.mean(gap), "median": st.median(gap), "p90": sorted(gap)[len(gap) * 9 // 10], "max": max(gap)}
        rec["period_ns_mean"] = (steps[-1][0] - steps[0][0]) / (len(steps) - 1)
        # the timed region = the last `--steps` launches: its duration profile in blocks of 50 launches, and the longest idle
        # gap of the GPU in the 2,000 trace entries before it (a cold GPU ramps its clocks over the next ~150 launches)
        allr = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "step_kernel" in r["Kernel_Name"]) for r in rows))
        si = [i for i, r in enumerate(allr) if r[2]]
        nt = int(sys.argv[2]) if len(sys.argv) > 2 else 400
        win = si[-nt:]
        wd = [allr[i][1] - allr[i][0] for i in win]
        lead = allr[max(0, win[0] - 2000): win[0] + 1]
        rec["timed_window"] = {"launches": len(win), "dur_ns_mean": st.mean(wd), "dur_ns_median": st.median(wd),
                               "dur_us_mean_by_block_of_50": [round(st.mean(wd[b:b + 50]) / 1e3, 1) for b in range(0, len(wd), 50)],
                               "longest_idle_gap_before_us": round(max(lead[i + 1][0] - lead[i][1] for i in range(len(lead) - 1)) / 1e3, 1)}
        others = {}
        for r in rows:
            if "step_kernel" not in r["Kernel_Name"]:
                k = r["Kernel_Name"][:90]
                others.setdefault(k, [0, 0])
                others[k][0] += 1
                others[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        rec["other_kernels"] = {k: {"calls": v[0], "total_ns": v[1]} for k, v in sorted(others.items(), key=lambda kv: -kv[1][1])[:12]}
        rec["grid_wg_lds_vgpr"] = next(([r["Grid_Size_X"], r["Workgroup_Size_X"], r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Queue_Id"), r.get("Stream_Id")] for r in rows if "step_kernel" in r["Kernel_Name"]), None)
    for line in open(f"{d}.log"):
        if line.startswith("{") and '"metric"' in line:
            b = json.loads(line)
            rec["bench"] = {"ms_per_step": b["ms_per_step"], "kernel_ms": b["roofline"]["kernel_ms"], "value": b["value"]}
    out[var] = rec
print(json.dumps(out, indent=1))
