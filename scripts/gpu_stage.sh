#!/bin/bash
# One GPU-box call: the parity suite, then the one-process A/B of the builds under ab/, then per-wave instruction
# counters of each build.   usage: scripts/gpu_stage.sh <tag> [pytest -k expr]
TAG=${1:-stage}; K=${2:-}
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ -n "$K" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$K" > gpurun_out/pytest_$TAG.log 2>&1
else
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1
fi
RC=$?
tail -5 gpurun_out/pytest_$TAG.log
if [ $RC -ne 0 ]; then echo "pytest failed rc=$RC"; exit $RC; fi
if ls ab/*.so >/dev/null 2>&1; then
  timeout -k 10 300 python scripts/ab_bench.py ab/*.so 5 100 > gpurun_out/ab_$TAG.log 2>&1 && cat gpurun_out/ab_$TAG.log || { tail -5 gpurun_out/ab_$TAG.log; exit 1; }
  timeout -k 10 400 bash scripts/pmc_insts.sh ab/*.so > /dev/null 2>&1 && cp gpurun_out/pmc_insts.txt gpurun_out/pmc_insts_$TAG.txt && cat gpurun_out/pmc_insts_$TAG.txt
fi
