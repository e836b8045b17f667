#!/usr/bin/env python3
"""What a Python consumer sees: host-buffer API latencies (read_state / step), and the gym-style vector env in its
two modes - numpy in / out (the gym kernel's outputs copied to pinned host buffers) and device_outputs (CUDA tensors,
actions decoded on the device, one launch per step).   usage: scripts/bench_host_api.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
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
      f"step+mask {t(lambda: e.step(acts, want_mask=True)):.2f} ms | legal_action_mask_bits {t(lambda: e.legal_action_mask_bits()):.2f} ms", flush=True)
del e

for label, kw in (("default: numpy in / out over the gym kernels", dict()),):
    env = GeneralsVecEnv(num_envs=B, board_width=20, board_height=20, max_players=4, **kw)
    obs, info = env.reset(seed=3)
    holder = [info]
    def vstep():
        m = holder[0]["valid_actions_mask"]
        a = np.argmax(m, axis=1)                      # the first valid action of every env (0 where there is none)
        o, r, term, trunc, inf = env.step(a)
        holder[0] = inf
    ms = t(vstep, 10)
    print(f"GeneralsVecEnv.step ({label})   {ms:8.2f} ms  ({B / ms * 1e3 / 1e6:.3f} M env-steps/s)", flush=True)
    env.close()

for BB in sorted({B, 65536}):
    env = GeneralsVecEnv(num_envs=BB, board_width=20, board_height=20, max_players=4, device_outputs=True)
    obs, info = env.reset(seed=3)
    holder = [info]
    def dstep():
        a = torch.argmax(holder[0]["valid_actions_mask"].to(torch.uint8), dim=1)
        o, r, term, trunc, inf = env.step(a)
        holder[0] = inf
    def timed(n):
        dstep(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): dstep()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    ms = timed(50)
    print(f"GeneralsVecEnv.step (device_outputs, B={BB}, argmax policy on the device) {ms:8.4f} ms  ({BB / ms * 1e3 / 1e6:.3f} M env-steps/s)", flush=True)
    fixed = torch.argmax(holder[0]["valid_actions_mask"].to(torch.uint8), dim=1)
    def fstep():
        env.step(fixed)
    dstep = fstep
    ms = timed(200)
    print(f"GeneralsVecEnv.step (device_outputs, B={BB}, the step alone: one gvec_gym_step launch) {ms:8.4f} ms  ({BB / ms * 1e3 / 1e6:.3f} M env-steps/s)", flush=True)
    env.close()
