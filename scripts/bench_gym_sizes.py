#!/usr/bin/env python3
"""gvec_gym_step alone (device_outputs, fixed actions) over board sizes: ms per step, bytes written per env (observation +
mask) and the rate they leave at.   usage: scripts/bench_gym_sizes.py [B]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generalsreinforcementlearning_amd.vector_env import GeneralsVecEnv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rows = []
for (w, h, p) in ((15, 15, 2), (16, 16, 2), (10, 10, 2), (20, 20, 4), (20, 20, 2), (25, 25, 4), (32, 32, 8)):
    env = GeneralsVecEnv(B, board_width=w, board_height=h, max_players=p, seed=1, device_outputs=True)
    obs, info = env.reset()
    a = torch.argmax(info["valid_actions_mask"].to(torch.uint8), dim=1)
    for _ in range(20):
        env.step(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        env.step(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    out_b = 9 * w * h * 4 + 5 * w * h
    state_b = env.engine.step_traffic_bytes() if hasattr(env.engine, "step_traffic_bytes") else None
    rows.append({"board": f"{w}x{h}", "players": p, "ms": ms, "M_env_steps_s": B / ms / 1e3, "obs_mask_bytes": out_b,
                 "obs_mask_GBps": B * out_b / ms / 1e6, "state_bytes": state_b})
    env.close()
print(json.dumps({"envs": B, "rows": rows}))
