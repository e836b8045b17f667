#!/usr/bin/env python3
"""gvec_observe (Serializer.StateToTensor on the device, one player / all players) over board sizes: ms and write rate.
usage: scripts/bench_observe_sizes.py [B]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import generalsreinforcementlearning_amd as g
from generalsreinforcementlearning_amd._lib import check

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rows = []
for (w, h, p) in ((15, 15, 2), (16, 16, 2), (10, 10, 2), (20, 20, 4), (25, 25, 4), (32, 32, 4)):
    eng = g.VecEngine(B, w, h, p, stream=torch.cuda.current_stream().cuda_stream)
    eng.reset_generated(1)
    eng.rollout(30, 1, 0, fused=True, want_stats=False)
    obs = torch.empty(B * p * 9 * w * h, dtype=torch.float32, device="cuda")
    for player, label in ((0, "one_player"), (-1, "all_players")):
        n = 1 if player == 0 else p
        check(eng.L.gvec_observe(eng.h, player, obs.data_ptr(), 1)); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            check(eng.L.gvec_observe(eng.h, player, obs.data_ptr(), 1))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        rows.append({"board": f"{w}x{h}", "players": p, "what": label, "ms": ms, "write_GBps": B * n * 9 * w * h * 4 / ms / 1e6})
    eng.close()
print(json.dumps({"envs": B, "rows": rows}))
