#!/bin/bash
# Exact per-wave instruction counts of the step / rollout kernels for one or more builds (run on the GPU box):
#   scripts/pmc_insts.sh ab/libA.so ab/libB.so ...   ->  gpurun_out/pmc_insts.txt
# Counter pass only (--pmc with --kernel-trace; no other tracing domain).
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/pmc_insts.txt
for L in "$@"; do
  D=gpurun_out/pmc_$(basename $L .so)
  rm -rf $D
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $D -- python3 scripts/ab_bench.py $L 1 10 > $D.log 2>&1 || { echo "rocprofv3 failed for $L" >> gpurun_out/pmc_insts.txt; continue; }
  python3 - "$L" $D >> gpurun_out/pmc_insts.txt <<'PY'
import sys, glob, csv, collections, statistics
lib, d = sys.argv[1:3]
for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "step" if "step_kernel" in k else "rollout" if "rollout_kernel" in k else None
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        w = statistics.median(c["SQ_WAVES"])
        print(lib, k, " ".join(f"{n[3:]}={statistics.median(v)/w:.1f}" for n, v in sorted(c.items()) if n != "SQ_WAVES"))
PY
done
cat gpurun_out/pmc_insts.txt
