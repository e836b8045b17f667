#!/usr/bin/env python3
"""Interleaved A/B timing of two builds of libgvec_hip.so in ONE process on ONE device
(cdna_hip_programming.md 5.4 rule 24).   usage: scripts/ab_bench.py [--dup] libA.so libB.so [libC.so ...] [rounds] [steps]"""
import statistics as st
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generalsreinforcementlearning_amd import _lib
from generalsreinforcementlearning_amd.vec_engine import VecEngine

paths = [a for a in sys.argv[1:] if a.endswith(".so")]
if "--dup" in sys.argv:  # every build twice, interleaved: shows how much of a difference is buffer placement / order
    sys.argv.remove("--dup")
    paths = paths + paths
nums = [int(a) for a in sys.argv[1:] if not a.endswith(".so")]
rounds = nums[0] if len(nums) > 0 else 7
steps = nums[1] if len(nums) > 1 else 100
B, W, H, P = 262144, 20, 20, 4
engs = []
for p in paths:
    L = _lib.load_from(os.path.abspath(p))
    e = VecEngine(B, W, H, P, auto_reset=True, lib=L, stream=torch.cuda.current_stream().cuda_stream)
    e.reset_generated(1000003)
    e.build_board_pool(4096, 7919)
    e.rollout(20, 1, 0, fused=False, want_stats=False)
    engs.append(e)
torch.cuda.synchronize()
res = [[] for _ in engs]
fres = [[] for _ in engs]
for r in range(rounds):
    for i, e in enumerate(engs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        e.rollout(steps, 1, 0, fused=False, want_stats=False)
        b.record()
        torch.cuda.synchronize()
        res[i].append(a.elapsed_time(b) / steps)
        a.record()
        e.rollout(32, 2, 0, fused=True, want_stats=False)
        b.record()
        torch.cuda.synchronize()
        fres[i].append(a.elapsed_time(b) / 32)
for i, p in enumerate(paths):
    m, f = st.median(res[i]), st.median(fres[i])
    print(f"{p}: per-turn median {m*1e3:.1f} us/step (min {min(res[i])*1e3:.1f}) -> {B/m/1e3:.1f} M steps/s | fused {f*1e3:.1f} us/turn -> {B/f/1e3:.1f} M steps/s")
