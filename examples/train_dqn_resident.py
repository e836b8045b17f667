#!/usr/bin/env python3
"""A DQN learner over the resident collection loop - what `python/train_dqn_parallel.py` of the reference does with
ParallelEnvPool + ReplayBuffer over gRPC, here with env, policy, replay ring and learner all on one MI355X:

    GeneralsVecEnv(device_outputs=True)  ->  epsilon-greedy over the masked Q values (one forward for all envs)
        ->  gvec_gym_step  ->  gvec_pool_collect (ring in HBM, episode results)  ->  DeviceReplayBuffer.sample_arrays

The network, loss and exploration scheme follow the reference's choices in outline (a small conv net over the (9, H, W)
observation, Huber loss, gradient clipping at 1.0, a hard target update, fixed per-worker exploration rates spread
geometrically over the workers); the code is this repo's own.  An example, not part of the measured hot path.

    python examples/train_dqn_resident.py --num-envs 4096 --updates 200
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import torch.nn.functional as F

from generalsreinforcementlearning_amd.env_pool import DeviceReplayBuffer, ParallelVecEnvPool
from generalsreinforcementlearning_amd.vector_env import GeneralsVecEnv


class QNet(nn.Module):
    def __init__(self, obs_shape, n_actions, width=64):
        super().__init__()
        c, h, w = obs_shape
        self.body = nn.Sequential(nn.Conv2d(c, width, 3, padding=1), nn.ReLU(), nn.Conv2d(width, width, 3, padding=1), nn.ReLU(),
                                  nn.Conv2d(width, 5, 1))
        self.n_actions = n_actions

    def forward(self, x):
        # five action planes (up, right, down, left, half) per tile -> the env's index tile * 5 + d
        return self.body(x).permute(0, 2, 3, 1).reshape(x.shape[0], self.n_actions)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--board", type=int, default=15)
    ap.add_argument("--buffer-size", type=int, default=1_000_000)
    ap.add_argument("--batch-size", type=int, default=1024)
    ap.add_argument("--updates", type=int, default=200)
    ap.add_argument("--collect-per-update", type=int, default=4, help="vector steps between two learner steps")
    ap.add_argument("--warmup-steps", type=int, default=8)
    ap.add_argument("--gamma", type=float, default=0.99)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--target-every", type=int, default=50)
    ap.add_argument("--max-steps-per-episode", type=int, default=200)
    ap.add_argument("--eps-base", type=float, default=0.4)
    ap.add_argument("--seed", type=int, default=42)
    a = ap.parse_args(argv)

    torch.manual_seed(a.seed)
    dev = torch.device("cuda", 0)
    B, n_actions, obs_shape = a.num_envs, a.board * a.board * 5, (9, a.board, a.board)
    q, target = QNet(obs_shape, n_actions).to(dev), QNet(obs_shape, n_actions).to(dev)
    target.load_state_dict(q.state_dict())
    opt = torch.optim.Adam(q.parameters(), lr=a.lr)
    # one fixed exploration rate per worker, eps_base ** (1 .. 8) across the pool
    eps = a.eps_base ** (1 + 7 * torch.arange(B, device=dev) / max(B - 1, 1))

    def policy(states, masks, _workers, gen):
        with torch.no_grad():
            qv = q(states).masked_fill(~masks, float("-inf"))
            greedy = qv.argmax(1)
            explore = (masks * torch.rand(masks.shape, device=dev, generator=gen)).argmax(1)      # a uniform valid action
            return torch.where(torch.rand(B, device=dev, generator=gen) < eps, explore, greedy)

    buf = DeviceReplayBuffer(a.buffer_size)
    pool = ParallelVecEnvPool(B, lambda n: GeneralsVecEnv(n, board_width=a.board, board_height=a.board, max_players=2, max_turns=a.max_steps_per_episode,
                                                         seed=a.seed, device_outputs=True),
                              policy, buf, max_steps_per_episode=a.max_steps_per_episode, seed=a.seed, batched_actions=True)
    pool.collect(a.warmup_steps)
    losses, t0 = [], time.perf_counter()
    for u in range(a.updates):
        pool.collect(a.collect_per_update)
        s, act, r, ns, d = buf.sample_arrays(a.batch_size)
        with torch.no_grad():
            tq = r.float() + a.gamma * target(ns).max(1).values * (~d).float()
        loss = F.smooth_l1_loss(q(s).gather(1, act[:, None]).squeeze(1), tq)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        nn.utils.clip_grad_norm_(q.parameters(), 1.0)
        opt.step()
        if (u + 1) % a.target_every == 0:
            target.load_state_dict(q.state_dict())
        if (u + 1) % 50 == 0 or u + 1 == a.updates:
            losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    results = pool.pop_episode_results()
    out = {"updates": a.updates, "env_steps": pool.total_env_steps, "episodes": pool.total_episodes, "seconds": dt,
           "env_steps_per_s": a.updates * a.collect_per_update * B / dt, "updates_per_s": a.updates / dt, "loss": losses,
           "mean_episode_reward": sum(x[0] for x in results) / max(len(results), 1), "ring_fill": len(buf)}
    pool._env.close()
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
