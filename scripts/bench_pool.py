#!/usr/bin/env python3
"""The collection loop `train_dqn_parallel.py` runs (ParallelEnvPool + ReplayBuffer), in its two forms here:
host (numpy observations, `ReplayBuffer.push_batch`) and resident (DeviceReplayBuffer + gvec_pool_collect: env, policy
and ring on the GPU).  Random valid-action policy.  Prints one JSON line.   usage: scripts/bench_pool.py [B] [W] [H] [P]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from generalsreinforcementlearning_amd.env_pool import ParallelVecEnvPool, ReplayBuffer, DeviceReplayBuffer
from generalsreinforcementlearning_amd.vector_env import GeneralsVecEnv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
W = int(sys.argv[2]) if len(sys.argv) > 2 else 15
H = int(sys.argv[3]) if len(sys.argv) > 3 else 15
P = int(sys.argv[4]) if len(sys.argv) > 4 else 2
out = {"envs": B, "board": f"{W}x{H}", "players": P, "max_steps_per_episode": 200}
mk = lambda dev: (lambda n: GeneralsVecEnv(n, board_width=W, board_height=H, max_players=P, seed=1, board_pool=1024, device_outputs=dev))

rng = np.random.default_rng(0)
def host_policy(states, masks, workers, rngs):
    return (masks * rng.random(masks.shape, dtype=np.float32)).argmax(1)
def device_policy(states, masks, workers, gen):
    return (masks * torch.rand(masks.shape, device=masks.device, generator=gen)).argmax(1)
def fixed_policy(states, masks, workers, gen):
    return fixed

# ---- host form
cap = max(4 * B, 50000)
pool = ParallelVecEnvPool(B, mk(False), host_policy, ReplayBuffer(cap), max_steps_per_episode=200, batched_actions=True)
pool.collect(3)
n = max(3, min(30, (1 << 17) // B))
t0 = time.perf_counter(); pool.collect(n); dt = time.perf_counter() - t0
out["host"] = {"ms_per_vector_step": dt / n * 1e3, "transitions_per_s": B * n / dt}
pool._env.close(); del pool

# ---- resident form
obs_bytes = 9 * W * H * 4
cap = min(max(4 * B, 200000), int(40e9 // (2 * obs_bytes)))
buf = DeviceReplayBuffer(cap)
pool = ParallelVecEnvPool(B, mk(True), device_policy, buf, max_steps_per_episode=200, batched_actions=True)
pool.collect(20); torch.cuda.synchronize()
n = 300
t0 = time.perf_counter(); pool.collect(n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["resident"] = {"ms_per_vector_step": dt / n * 1e3, "transitions_per_s": B * n / dt, "ring_capacity": cap,
                   "ring_bytes_per_transition": 2 * obs_bytes + 17}
# the same with a policy that costs nothing: the env step + the collection alone
fixed = device_policy(pool._state, pool._mask, None, pool._generator)
pool.action_fn = fixed_policy
pool.collect(20); torch.cuda.synchronize()
t0 = time.perf_counter(); pool.collect(n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["resident_no_policy"] = {"ms_per_vector_step": dt / n * 1e3, "transitions_per_s": B * n / dt}
# gvec_pool_collect alone, by events
env, dc = pool._env, pool._dc
s, m = pool._state, pool._mask
ns, r, te, tr, info = env.step(fixed)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    dc.collect(s, fixed, ns, r, te, tr, info["reset"], env.needs_reset_buffer())
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 100
out["gvec_pool_collect"] = {"ms": ms, "bytes_moved": 4 * B * obs_bytes, "GBps": 4 * B * obs_bytes / ms / 1e6}
x = buf.sample_arrays(256); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    x = buf.sample_arrays(256)
torch.cuda.synchronize()
out["sample_arrays_256_ms"] = (time.perf_counter() - t0) / 20 * 1e3
out["episodes"], out["transitions"] = pool.total_episodes, pool.total_env_steps
print(json.dumps(out))
