#!/usr/bin/env python3
"""Host-buffer API latencies (read_state / step / vector env step) at a modest batch: what a Python consumer sees."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import generalsreinforcementlearning_amd as g
from generalsreinforcementlearning_amd.vector_env import GeneralsVecEnv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
e = g.VecEngine(B, 20, 20, 4)
e.reset_generated(1)
acts = e.agent_actions(1)
def t(f, n=20):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print(f"B={B}: game_state() {t(lambda: e.game_state()):.2f} ms | step(host acts) {t(lambda: e.step(acts)):.2f} ms | "
      f"step+mask {t(lambda: e.step(acts, want_mask=True)):.2f} ms | legal_action_mask_bits {t(lambda: e.legal_action_mask_bits()):.2f} ms")
env = GeneralsVecEnv(num_envs=B, board_width=20, board_height=20, max_players=4)
obs, info = env.reset(seed=3)
def vstep():
    m = info_holder[0]["valid_actions_mask"]
    a = np.array([np.flatnonzero(r)[0] if r.any() else 0 for r in m[: 64]] + [0] * (B - 64))
    o, r, term, trunc, inf = env.step(a)
    info_holder[0] = inf
info_holder = [info]
print(f"GeneralsVecEnv.step {t(vstep, 10):.2f} ms  ({B / t(vstep, 10) * 1e3 / 1e6:.2f} M env-steps/s through the gym-style host API)")
