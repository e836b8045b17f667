#!/bin/bash
# The other BASELINE.json configurations through bench.py (parity-test cases, not bench lines) -> gpurun_out/bench_configs.txt
mkdir -p gpurun_out; : > gpurun_out/bench_configs.txt
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', '|', round(d['value']/1e6,1), 'M/s kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3), 'fused', round(d['fused_rollout']['env_steps_per_s_per_gpu']/1e6), 'M/s')
" >> gpurun_out/bench_configs.txt; }
run --envs-per-gpu 65536 --width 15 --height 15 --players 2
run --envs-per-gpu 4096 --width 10 --height 10 --players 2 --fog 0
run --envs-per-gpu 32768
run --envs-per-gpu 65536 --width 25 --height 25 --players 4
run --envs-per-gpu 32768 --width 32 --height 32 --players 8
run --envs-per-gpu 98304 --mixed
cat gpurun_out/bench_configs.txt
