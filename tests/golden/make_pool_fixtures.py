#!/usr/bin/env python3
"""Build-container only: records what the REFERENCE's own ParallelEnvPool and ReplayBuffer
(python/generals_gym/vector_env.py, replay_buffer.py - loaded from /root/reference as they lie; neither needs gymnasium,
only the package's __init__ does, so the two files are loaded as a two-module package of their own) do with the scripted
environment of tests/_scripted_env.py, and commits the result as tests/golden/pool_fixtures.json:

  pool    3 workers x 4 whole episodes, max_steps_per_episode 8, seed 42, the reference test's random policy:
          per worker, the transitions pushed in order - (state ids, action, reward, next-state ids, done) - and the
          (episode_reward, episode_length, worker_id) results
  buffer  capacity 5, 12 pushes, random.seed(7): total_pushed, len, and two consecutive sample(3) draws

tests/test_env_pool.py replays both against ParallelVecEnvPool / ReplayBuffer on every box (no reference needed there)."""
import importlib.util
import json
import os
import random
import sys
import threading
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _scripted_env as S  # noqa: E402

REF = "/root/reference/python/generals_gym"


def load_reference():
    pkg = types.ModuleType("refgym")
    pkg.__path__ = [REF]
    sys.modules["refgym"] = pkg
    mods = {}
    for name in ("replay_buffer", "vector_env"):
        spec = importlib.util.spec_from_file_location(f"refgym.{name}", os.path.join(REF, f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"refgym.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods["replay_buffer"].ReplayBuffer, mods["vector_env"].ParallelEnvPool


def ids(o):
    return [int(o[0, 0, 0]), int(o[0, 0, 1]), int(o[0, 1, 0]), int(o[0, 1, 1])]


def main():
    ReplayBuffer, ParallelEnvPool = load_reference()
    W, E, MAXS, SEED = 3, 4, 8, 42
    release = threading.Event()

    def gate(worker, episode):
        if episode >= E:
            release.wait(30)            # the worker has played its quota: parked until the pool is told to stop

    buf = ReplayBuffer(100000)
    pool = ParallelEnvPool(num_envs=W, env_factory=lambda w: S.ScriptedEnv(w, gate), action_fn=S.random_action_fn, replay_buffer=buf,
                           max_steps_per_episode=MAXS, seed=SEED)
    pool.start()
    t0 = time.time()
    while pool.total_episodes < W * E:
        assert time.time() - t0 < 30 and pool.alive_workers == W
        time.sleep(0.01)
    pool._stop_event.set()
    release.set()
    pool.stop(join_timeout=10.0)
    assert pool.alive_workers == 0 and pool.total_episodes == W * E
    per = {w: [] for w in range(W)}
    for s, a, r, ns, d in buf._buffer:
        assert isinstance(a, int)
        per[int(s[0, 0, 0])].append({"state": ids(s), "action": a, "reward": r, "next_state": ids(ns), "done": bool(d)})
    results = {w: [] for w in range(W)}
    for rew, length, w in pool.pop_episode_results():
        results[w].append([rew, length, w])
    assert pool.pop_episode_results() == []
    out = {"source": "python/generals_gym/vector_env.py:28-192 + replay_buffer.py:13-55, run by tests/golden/make_pool_fixtures.py",
           "pool": {"num_envs": W, "episodes_per_worker": E, "max_steps_per_episode": MAXS, "seed": SEED,
                    "total_env_steps": pool.total_env_steps, "transitions": {str(w): per[w] for w in per},
                    "episode_results": {str(w): results[w] for w in results}}}
    # ---- ReplayBuffer alone
    rb = ReplayBuffer(5)
    for i in range(12):
        rb.push(S.obs_of(0, 0, i, -1), i, i * 0.5, S.obs_of(0, 0, i + 1, i), i % 4 == 3)
    random.seed(7)
    draws = [[[ids(s), a, r, ids(ns), bool(d)] for s, a, r, ns, d in rb.sample(3)] for _ in range(2)]
    try:
        rb.sample(6)
        too_many = None
    except ValueError as e:
        too_many = "ValueError"
    try:
        ReplayBuffer(0)
        zero = None
    except ValueError:
        zero = "ValueError"
    out["buffer"] = {"capacity": 5, "pushes": 12, "seed": 7, "total_pushed": rb.total_pushed, "len": len(rb), "draws": draws,
                     "sample_more_than_len": too_many, "capacity_zero": zero}
    json.dump(out, open(os.path.join(HERE, "pool_fixtures.json"), "w"), indent=1)
    print("wrote pool_fixtures.json:", {w: len(per[w]) for w in per}, "transitions;", pool.total_env_steps, "steps")


if __name__ == "__main__":
    main()
