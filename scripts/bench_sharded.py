#!/usr/bin/env python3
"""What the fan-out of a sharded handle costs per call (worker-thread hand-off), on one GPU: a plain handle against sharded
handles with 1 and 2 shards on device 0, per-turn rollouts issued one call per turn and 50 turns per call.
usage: scripts/bench_sharded.py [B]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import generalsreinforcementlearning_amd as g

B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
out = {"boards": B}
for name, devs in (("plain", None), ("sharded_x1", [0]), ("sharded_x2_same_device", [0, 0])):
    e = g.VecEngine(B, 20, 20, 4, auto_reset=True, devices=devs)
    e.reset_generated(1)
    e.build_board_pool(1024, 2)
    e.rollout(100, 3, 0, fused=False, want_stats=False)
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        e.rollout(1, 3, 0, fused=False, want_stats=False)
    e.synchronize()
    per_call = (time.perf_counter() - t0) / 200 * 1e3
    t0 = time.perf_counter()
    for _ in range(4):
        e.rollout(50, 3, 0, fused=False, want_stats=False)
    e.synchronize()
    per_turn_50 = (time.perf_counter() - t0) / 200 * 1e3
    out[name] = {"ms_per_turn_one_call_per_turn": per_call, "ms_per_turn_50_turns_per_call": per_turn_50}
    e.close()
print(json.dumps(out))
