#!/usr/bin/env python3
"""How long the step kernel takes when every board is frozen (finished game, no auto-reset): it then loads and stores
header, planes and armies and does no turn work - a quick in-situ check of the memory side of the kernel
(the standalone version with every array is scripts/microbench/copy_pattern.hip).  Run on a GPU box."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import generalsreinforcementlearning_amd as g
B=262144
e = g.VecEngine(B, 20, 20, 4, auto_reset=False, stream=torch.cuda.current_stream().cuda_stream)
e.reset_generated(123)
e.rollout(5, 1, 0, fused=False, want_stats=False)
def timeit(n=100):
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record(); e.rollout(n, 1, 0, fused=False, want_stats=False); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
print("live boards: %.1f us/step" % timeit())
e.write_state({"done": np.ones(B, np.uint8)})
for _ in range(3): t=timeit()
print("all boards frozen (load hdr+planes+army, store hdr+planes+army, no compute): %.1f us/step -> %.0f GB/s of 5388 B/env" % (t, B*5388/t/1e3))
