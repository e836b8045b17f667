#!/usr/bin/env python3
"""Reset cost (SURVEY 8d "reported separately"): B boards 20x20 4P through the three ways a batch can start -
gvec_reset_generated (parallel counter RNG), gvec_reset_go_seeded (Go's math/rand per board: 607-word state each) and
gvec_reset (boards uploaded from the host).   usage: scripts/bench_reset.py [B]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import generalsreinforcementlearning_amd as g

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
eng = g.VecEngine(B, 20, 20, 4)
def t(f, n=3):
    f(); eng.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    eng.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
out = {"boards": B}
out["reset_generated_ms"] = t(lambda: eng.reset_generated(3))
seeds = np.arange(1, B + 1, dtype=np.int64)
out["reset_go_seeded_ms"] = t(lambda: eng.reset_go_seeded(seeds))
st = eng.game_state(fields=("army", "owner", "type", "width", "height", "players"))
out["reset_uploaded_ms"] = t(lambda: eng.reset(st["army"], st["owner"], st["type"], st["width"], st["height"], st["players"]))
for k in ("reset_generated_ms", "reset_go_seeded_ms", "reset_uploaded_ms"):
    out[k.replace("_ms", "_boards_per_s")] = B / out[k] * 1e3
print(json.dumps(out))
