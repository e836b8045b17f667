#!/bin/bash
# Why does an RCCL communicator cost the step kernel 5-7 %?    Kernel traces of the same bench run
# without torch.distributed, with an RCCL process group (world size 1, no gather), and with a gloo one; the summariser
# compares the step kernel's own duration (profiler timestamps) and the gaps between consecutive launches.
# usage (GPU box): scripts/rccl_tax.sh <tag>
TAG=${1:-r03}
export TMPDIR=/tmp
OUT=gpurun_out/rccl_tax_$TAG
mkdir -p $OUT
COMMON="--no-cpu-baseline --no-fused --steps 400 --warmup 20"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/plain -- python3 bench.py $COMMON > $OUT/plain.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rccl -- python3 bench.py $COMMON --force-dist --gather-envs 0 > $OUT/rccl.log 2>&1 || exit 1
GVEC_BENCH_DIST_INIT=gloo rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gloo -- python3 bench.py $COMMON --force-dist --gather-envs 0 > $OUT/gloo.log 2>&1 || exit 1
python3 scripts/rccl_tax_summary.py $OUT > $OUT/summary.json && cat $OUT/summary.json
