#!/usr/bin/env python3
"""Does a HIP graph buy the small-batch gym step anything?  GeneralsVecEnv (device_outputs) at B envs: eager steps (one
gvec_gym_step launch each) against ONE captured graph of six consecutive steps (the buffer rotation's period) replayed.
Round 2 measured a captured FOUR-launch step at 0.757 ms per replay against 0.070 ms eager and could not say why; run this
under `rocprofv3 --kernel-trace --stats` to see what a replay executes.   usage: scripts/graph_replay.py [B] [iters]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generalsreinforcementlearning_amd.vector_env import GeneralsVecEnv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 600
env = GeneralsVecEnv(num_envs=B, board_width=20, board_height=20, max_players=4, device_outputs=True)
obs, info = env.reset(seed=3)
acts = torch.argmax(info["valid_actions_mask"].to(torch.uint8), dim=1)


def timed(f, n):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = timed(lambda: env.step(acts), N)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(N):
    env.step(acts)
ev1.record()
torch.cuda.synchronize()
eager_gpu = ev0.elapsed_time(ev1) / N
out = {"envs": B, "eager_ms_per_step_wall": eager, "eager_ms_per_step_gpu_events": eager_gpu}
try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        env.engine.set_stream(torch.cuda.current_stream().cuda_stream)   # the capture stream
        for _ in range(6):
            env.step(acts)
    env.engine.set_stream(torch.cuda.current_stream().cuda_stream)
    rep = timed(g.replay, N // 6)
    out.update({"graph_steps_per_replay": 6, "graph_ms_per_replay": rep, "graph_ms_per_step": rep / 6})
except Exception as e:  # noqa: BLE001
    out["graph_error"] = repr(e)[:300]
print(json.dumps(out))
