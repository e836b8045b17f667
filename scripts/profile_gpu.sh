#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of the bench command.
# usage: scripts/profile_gpu.sh <tag> [extra bench args]
# Outputs under gpurun_out/prof_<tag>/ ; summarise afterwards with scripts/summarize_profile.py <tag>.
set -e
TAG=${1:-run}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
# kernel trace: bench.py's default step counts, so the kernel's average here is the one bench.py reports
T="python3 bench.py --no-cpu-baseline --no-fused $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $T > $OUT/trace.log 2>&1
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fused $@"
# PMC passes: counters only (never combined with sys/hip/hsa tracing)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- $B > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1
grep -h '"metric"' $OUT/trace.log | tail -1 > $OUT/bench_line.json || true
echo "profile $TAG done"
