#!/usr/bin/env python3
"""Per-turn and fused step times of two or more builds at several board sizes (one process, one device).
usage: scripts/ab_bench_sizes.py libA.so libB.so ..."""
import sys, os, statistics as st
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generalsreinforcementlearning_amd import _lib
from generalsreinforcementlearning_amd.vec_engine import VecEngine
cfgs = [(32768, 32, 32, 8), (65536, 25, 25, 4), (65536, 15, 15, 2)]
for (B, W, H, P) in cfgs:
    for path in sys.argv[1:]:
        L = _lib.load_from(os.path.abspath(path))
        e = VecEngine(B, W, H, P, auto_reset=True, lib=L, stream=torch.cuda.current_stream().cuda_stream)
        e.reset_generated(5); e.build_board_pool(1024, 7); e.rollout(10, 1, 0, fused=False, want_stats=False)
        ts = []
        for r in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); e.rollout(50, 1, 0, fused=False, want_stats=False); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 50)
        m = st.median(ts)
        fs = []
        for r in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); e.rollout(32, 2, 0, fused=True, want_stats=False); b.record(); torch.cuda.synchronize()
            fs.append(a.elapsed_time(b) / 32)
        f = st.median(fs)
        print(f"{B}x{W}x{H} P{P} {os.path.basename(path)}: {m*1e3:.1f} us/step -> {B/m/1e3:.1f} M steps/s | fused {f*1e3:.1f} us/turn -> {B/f/1e3:.1f} M steps/s", flush=True)
        del e
