"""ParallelVecEnvPool / ReplayBuffer against what the REFERENCE's own ParallelEnvPool / ReplayBuffer did
with the same scripted environment (tests/golden/pool_fixtures.json, recorded by tests/golden/make_pool_fixtures.py from
python/generals_gym/vector_env.py + replay_buffer.py), then - under -m gpu - over the real GeneralsVecEnv."""
import json
import os
import random
import threading
import time

import numpy as np
import pytest

import _scripted_env as S
from generalsreinforcementlearning_amd.env_pool import ParallelVecEnvPool, ReplayBuffer

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pool_fixtures.json")))


def _ids(o):
    return [int(o[0, 0, 0]), int(o[0, 0, 1]), int(o[0, 1, 0]), int(o[0, 1, 1])]


@pytest.mark.parametrize("batched", [False, True], ids=["per_env_action_fn", "batched_action_fn"])
def test_pool_pushes_what_the_reference_pool_pushes(batched):
    """Per worker: the same transitions in the same order, the same (episode_reward, episode_length, worker_id) results -
    max_steps_per_episode cuts, the env's own terminations and truncations, the private per-worker RNG streams."""
    f = FIX["pool"]
    W, E = f["num_envs"], f["episodes_per_worker"]

    class Buf:                                   # a reference-shaped buffer: push() only
        def __init__(self):
            self.items, self.total_pushed = [], 0

        def push(self, s, a, r, ns, d):
            self.items.append((s, a, r, ns, d))
            self.total_pushed += 1

    def batched_fn(states, masks, workers, rngs):
        return np.array([S.random_action_fn(states[i], masks[i], w, rngs[w]) for i, w in enumerate(workers)])

    buf = Buf()
    pool = ParallelVecEnvPool(num_envs=W, env_factory=S.ScriptedVecEnv, action_fn=batched_fn if batched else S.random_action_fn,
                              replay_buffer=buf, max_steps_per_episode=f["max_steps_per_episode"], seed=f["seed"], batched_actions=batched)
    done_eps = {w: 0 for w in range(W)}
    results = {w: [] for w in range(W)}
    while min(done_eps.values()) < E:
        pool.collect(1)
        for rew, length, w in pool.pop_episode_results():
            results[w].append([rew, length, w])
            done_eps[w] += 1
    assert pool.total_env_steps == buf.total_pushed
    per = {w: [] for w in range(W)}
    for s, a, r, ns, d in buf.items:
        assert isinstance(a, int) and isinstance(d, bool) and isinstance(r, float)
        per[int(s[0, 0, 0])].append({"state": _ids(s), "action": a, "reward": r, "next_state": _ids(ns), "done": d})
    for w in range(W):
        want = f["transitions"][str(w)]
        got = [t for t in per[w] if t["state"][1] < E]            # the reference stopped every worker after E episodes
        assert got == want, (w, next((i, g, x) for i, (g, x) in enumerate(zip(got, want)) if g != x) if len(got) == len(want) else (len(got), len(want)))
        assert results[w][:E] == f["episode_results"][str(w)]


def test_replay_buffer_matches_the_reference_buffer():
    f = FIX["buffer"]
    rb = ReplayBuffer(f["capacity"])
    for i in range(f["pushes"]):
        rb.push(S.obs_of(0, 0, i, -1), i, i * 0.5, S.obs_of(0, 0, i + 1, i), i % 4 == 3)
    assert rb.total_pushed == f["total_pushed"] and len(rb) == f["len"]
    random.seed(f["seed"])
    draws = [[[_ids(s), a, r, _ids(ns), d] for s, a, r, ns, d in rb.sample(3)] for _ in range(2)]
    assert draws == f["draws"]
    random.seed(f["seed"])
    st, ac, rw, nx, dn = rb.sample_arrays(3)                     # the same draw as arrays
    assert [[_ids(st[i]), int(ac[i]), float(rw[i]), _ids(nx[i]), bool(dn[i])] for i in range(3)] == f["draws"][0]
    with pytest.raises(ValueError):
        rb.sample(6)
    with pytest.raises(ValueError):
        ReplayBuffer(0)
    assert f["sample_more_than_len"] == "ValueError" and f["capacity_zero"] == "ValueError"


def test_push_batch_equals_pushes():
    a, b = ReplayBuffer(7), ReplayBuffer(7)
    rng = np.random.default_rng(0)
    n = 0
    for k in (3, 5, 1, 9, 2):
        s, ns = rng.random((k, 9, 2, 2), np.float32), rng.random((k, 9, 2, 2), np.float32)
        ac, rw, dn = rng.integers(0, 50, k), rng.random(k), rng.random(k) < 0.3
        a.push_batch(s, ac, rw, ns, dn)
        for i in range(k):
            b.push(s[i], int(ac[i]), float(rw[i]), ns[i], bool(dn[i]))
        n += k
        assert a.total_pushed == b.total_pushed == n and len(a) == len(b)
        random.seed(1)
        x = a.sample(len(a))
        random.seed(1)
        y = b.sample(len(b))
        for (s1, a1, r1, n1, d1), (s2, a2, r2, n2, d2) in zip(x, y):
            assert np.array_equal(s1, s2) and a1 == a2 and r1 == r2 and np.array_equal(n1, n2) and d1 == d2


def test_replay_buffer_thread_safety():
    """python/test_parallel_env.py:19-51, against this buffer."""
    capacity, per_thread, n_threads = 500, 1000, 4
    buf = ReplayBuffer(capacity)
    errors = []

    def pusher(tid):
        try:
            for i in range(per_thread):
                buf.push(np.zeros((9, 5, 5), np.float32), i, 0.5, np.zeros((9, 5, 5), np.float32), False)
                if len(buf) >= 32:
                    assert len(buf.sample(32)) == 32
        except Exception as e:
            errors.append((tid, e))

    ts = [threading.Thread(target=pusher, args=(t,)) for t in range(n_threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors and buf.total_pushed == n_threads * per_thread and len(buf) == capacity


def test_pool_thread_start_stop_and_env_recreation():
    """start / stop / alive_workers (vector_env.py:62-112) and the retry pattern (:114-134, :141-150): a step that raises
    makes the pool recreate its env; a factory that keeps failing kills the collector after max_env_retries."""
    made = {"n": 0}

    class Flaky(S.ScriptedVecEnv):
        def step(self, actions):
            if made["n"] == 1 and self.t.max() >= 2:
                raise RuntimeError("lost connection")
            return super().step(actions)

    def factory(n):
        made["n"] += 1
        if made["n"] >= 3:
            raise RuntimeError("server gone")
        return Flaky(n)

    buf = ReplayBuffer(1000)
    pool = ParallelVecEnvPool(4, factory, S.random_action_fn, buf, max_steps_per_episode=5, max_env_retries=2, retry_sleep_s=0.01)
    pool.start()
    with pytest.raises(RuntimeError):
        pool.start()
    assert pool.alive_workers == 4
    t0 = time.time()
    while pool.total_episodes < 8 and time.time() - t0 < 20:
        time.sleep(0.01)
    assert made["n"] == 2 and pool.total_episodes >= 8          # env 1 failed once, env 2 took over
    made["n"] = 2

    class Dead(S.ScriptedVecEnv):
        def step(self, actions):
            raise RuntimeError("down")
    pool._env.__class__ = Dead                                   # the next step fails and the factory refuses: the collector dies
    t0 = time.time()
    while pool.alive_workers and time.time() - t0 < 20:
        time.sleep(0.01)
    assert pool.alive_workers == 0
    pool.stop(join_timeout=5.0)
    assert pool.total_env_steps == buf.total_pushed > 0


# ---- over the real vector env ------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_pool_over_generals_vec_env_pushes_the_env_s_own_transitions():
    """A random `action_fn` drives ParallelVecEnvPool over GeneralsVecEnv; a twin env stepped by hand with the same actions
    says what every transition must be: state / next_state are the observations GeneralsVecEnv.step returned, reward and
    done its outputs, the step after an episode's end (re-deal) is not a transition, and an episode cut at
    max_steps_per_episode restarts on a fresh board."""
    from generalsreinforcementlearning_amd.vector_env import GeneralsVecEnv
    B, MAXS = 24, 12
    kw = dict(board_width=8, board_height=8, max_players=2, max_turns=30, seed=6, board_pool=32)
    log = []

    def action_fn(state, valid_mask, worker_id, rng):
        v = np.flatnonzero(valid_mask)
        a = int(rng.choice(list(v))) if len(v) and rng.random() > 0.1 else int(rng.randrange(len(valid_mask)))   # 10 %: maybe invalid
        log.append((worker_id, a))
        return a

    class Buf:
        def __init__(self):
            self.items, self.total_pushed = [], 0

        def push(self, s, a, r, ns, d):
            self.items.append((s, a, r, ns, d))
            self.total_pushed += 1

    buf = Buf()
    pool = ParallelVecEnvPool(B, lambda n: GeneralsVecEnv(n, **kw), action_fn, buf, max_steps_per_episode=MAXS, seed=3)
    STEPS = 90
    pool.collect(STEPS)
    results = pool.pop_episode_results()
    # the twin
    twin = GeneralsVecEnv(B, **kw)
    obs, info = twin.reset()
    obs = obs.copy()
    ep_len, ep_rew = np.zeros(B, int), np.zeros(B)
    want, want_results, it = [], [], iter(log)
    starting = np.zeros(B, bool)
    for k in range(STEPS):
        acts = np.array([0 if starting[w] else next(it)[1] for w in range(B)])
        nobs, rew, term, trunc, info = twin.step(acts)
        nobs = nobs.copy()
        cut = np.zeros(B, bool)
        for w in range(B):
            if info["reset"][w]:
                continue
            d = bool(term[w] or trunc[w])
            want.append((obs[w], int(acts[w]), float(rew[w]), nobs[w], d))
            ep_len[w] += 1
            ep_rew[w] += rew[w]
            if d or ep_len[w] >= MAXS:
                want_results.append((float(ep_rew[w]), int(ep_len[w]), w))
                cut[w] = not d
                ep_len[w], ep_rew[w] = 0, 0.0
        if cut.any():
            twin.force_reset(cut)
        starting = np.array([bool(term[w] or trunc[w]) and not info["reset"][w] for w in range(B)]) | cut
        obs = nobs
    assert len(buf.items) == len(want) == pool.total_env_steps
    for i, ((s, a, r, ns, d), (ws, wa, wr, wns, wd)) in enumerate(zip(buf.items, want)):
        assert s.shape == (9, 8, 8) and s.dtype == np.float32 and isinstance(a, int) and isinstance(d, bool), i
        assert np.array_equal(s, ws) and a == wa and r == wr and np.array_equal(ns, wns) and d == wd, i
    assert results == want_results and pool.total_episodes == len(results)
    kinds = {"cut": sum(1 for r in results if r[1] == MAXS), "ended": sum(1 for r in results if r[1] < MAXS)}
    assert kinds["cut"] > 0 and len(results) >= B
    twin.close()


@pytest.mark.gpu
def test_pool_thread_over_generals_vec_env_with_batched_policy():
    """The threaded form with one batched policy call per vector step and this package's ReplayBuffer (push_batch)."""
    from generalsreinforcementlearning_amd.vector_env import GeneralsVecEnv
    B = 256
    rng = np.random.default_rng(0)

    def policy(states, masks, workers, rngs):
        assert states.shape == (len(workers), 9, 10, 10) and masks.shape == (len(workers), 500)
        pick = (masks * rng.random(masks.shape)).argmax(1)       # a random valid action per env (0 if none)
        return pick

    buf = ReplayBuffer(20000)
    pool = ParallelVecEnvPool(B, lambda n: GeneralsVecEnv(n, board_width=10, board_height=10, max_players=2, max_turns=40, seed=1, board_pool=64),
                              policy, buf, max_steps_per_episode=25, batched_actions=True)
    pool.start()
    t0 = time.time()
    while pool.total_episodes < 2 * B and time.time() - t0 < 60:
        assert pool.alive_workers == B
        time.sleep(0.05)
    pool.stop(join_timeout=10.0)
    assert pool.alive_workers == 0 and pool.total_episodes >= 2 * B
    assert pool.total_env_steps == buf.total_pushed and len(buf) == min(20000, buf.total_pushed)
    s, a, r, ns, d = buf.sample_arrays(64)
    assert s.shape == (64, 9, 10, 10) and s.dtype == np.float32 and ns.shape == s.shape and d.dtype == bool
    res = pool.pop_episode_results()
    assert len(res) >= 2 * B and {w for _, _, w in res} == set(range(B)) and max(l for _, l, _ in res) <= 25
    assert pool.pop_episode_results() == []


def _first_valid_host(states, masks, workers, rngs):
    return np.argmax(masks, axis=1)


def _first_valid_device(states, masks, workers, generator):
    import torch
    return torch.argmax(masks.to(torch.uint8), dim=1)


@pytest.mark.gpu
@pytest.mark.parametrize("max_turns,max_steps,B,board,STEPS", [(500, 7, 96, (7, 6), 40), (5, 9, 96, (7, 6), 40), (11, 4, 96, (7, 6), 40),
                                                            (500, 3, 16448, (5, 5), 8)],
                         ids=["cut_by_pool", "truncated_by_env", "mixed", "large_pool"])
def test_device_pool_equals_host_pool(max_turns, max_steps, B, board, STEPS):
    """The resident form (DeviceReplayBuffer + gvec_pool_collect) against the host form of the same pool over the same
    games: ring contents slot for slot (the ring wraps), counters, and the episode results in order."""
    import torch
    from generalsreinforcementlearning_amd.env_pool import DeviceReplayBuffer
    from generalsreinforcementlearning_amd.vector_env import GeneralsVecEnv
    CAP = B * 5 + 17
    mk = lambda dev_out: (lambda n: GeneralsVecEnv(n, board_width=board[0], board_height=board[1], max_players=2, max_turns=max_turns, seed=5,
                                                  board_pool=16, device_outputs=dev_out))
    hbuf = ReplayBuffer(CAP)
    host = ParallelVecEnvPool(B, mk(False), _first_valid_host, hbuf, max_steps_per_episode=max_steps, batched_actions=True)
    host.collect(STEPS)
    dbuf = DeviceReplayBuffer(CAP)
    dev = ParallelVecEnvPool(B, mk(True), _first_valid_device, dbuf, max_steps_per_episode=max_steps, batched_actions=True)
    dev.collect(STEPS)
    assert dev.total_env_steps == host.total_env_steps > CAP and len(dbuf) == len(hbuf) == CAP
    cursor, size, pushed, _ = dbuf.counters.tolist()
    assert (cursor, size, pushed) == (hbuf._cursor, hbuf._size, hbuf._pushed)
    assert np.array_equal(dbuf.state.cpu().numpy(), hbuf._state) and np.array_equal(dbuf.next_state.cpu().numpy(), hbuf._next)
    assert np.array_equal(dbuf.action.cpu().numpy(), hbuf._action) and np.array_equal(dbuf.reward.cpu().numpy(), hbuf._reward)
    assert np.array_equal(dbuf.done.cpu().numpy(), hbuf._done)
    assert dev.total_episodes == host.total_episodes > B
    want = host.pop_episode_results()
    got = dev.pop_episode_results()
    assert got == want and all(isinstance(l, int) and isinstance(w, int) and isinstance(r, float) for r, l, w in got)
    assert dev.pop_episode_results() == []
    dev.collect(3)
    host.collect(3)
    assert dev.pop_episode_results() == host.pop_episode_results() and dev.total_episodes == host.total_episodes
    for p in (host, dev):
        p._env.close()


@pytest.mark.gpu
def test_device_replay_buffer_sampling_and_result_log_overflow():
    import torch
    from generalsreinforcementlearning_amd.env_pool import DeviceReplayBuffer
    from generalsreinforcementlearning_amd.vector_env import GeneralsVecEnv
    B = 64
    buf = DeviceReplayBuffer(1000)
    with pytest.raises(ValueError):
        DeviceReplayBuffer(0)
    pool = ParallelVecEnvPool(B, lambda n: GeneralsVecEnv(n, board_width=6, board_height=6, max_players=2, seed=2, board_pool=8, device_outputs=True),
                              _first_valid_device, buf, max_steps_per_episode=3, batched_actions=True, result_capacity=100)
    pool.collect(2)
    with pytest.raises(ValueError):
        buf.sample_arrays(2 * B + 1)                        # fewer held than asked for
    pool.collect(10)                                        # 3 rounds of 64 episodes: 192 results, 100 kept
    episodes, held, dropped, _ = pool._dc.counters.tolist()
    assert episodes == pool.total_episodes == held + dropped and held == 100 and dropped > 0
    res = pool.pop_episode_results()
    assert len(res) == 100 and [w for _, _, w in res[:B]] == list(range(B)) and all(l == 3 for _, l, _ in res)
    buf.manual_seed(11)
    idx = buf.sample_indices(200)
    assert idx.unique().numel() == 200 and int(idx.max()) < len(buf)
    buf.manual_seed(11)
    s, a, r, ns, d = buf.sample_arrays(200)
    assert s.is_cuda and s.shape == (200, 9, 6, 6) and torch.equal(s, buf.state[idx]) and torch.equal(a, buf.action[idx])
    assert torch.equal(ns, buf.next_state[idx]) and torch.equal(r, buf.reward[idx]) and torch.equal(d, buf.done[idx])
    tuples = buf.sample(5)
    assert len(tuples) == 5 and tuples[0][0].shape == (9, 6, 6) and isinstance(tuples[0][1], int) and isinstance(tuples[0][4], bool)
    # a learner pushing by itself lands in the same ring
    before = buf.total_pushed
    buf.push_batch(s[:7], a[:7], r[:7], ns[:7], d[:7])
    buf.push(s[0].cpu().numpy(), 3, 0.5, ns[0].cpu().numpy(), True)
    assert buf.total_pushed == before + 8
    # argument checks of the entry point
    from generalsreinforcementlearning_amd._lib import CollectArgs, lib
    import ctypes
    bad = CollectArgs()
    assert lib().gvec_pool_collect(0, None, ctypes.byref(bad)) == -1
    with pytest.raises(ValueError):
        ParallelVecEnvPool(B, lambda n: GeneralsVecEnv(n, board_width=6, board_height=6, seed=2, board_pool=8, device_outputs=True),
                           _first_valid_device, DeviceReplayBuffer(B - 1), batched_actions=True).collect(1)
    pool._env.close()


@pytest.mark.gpu
def test_device_pool_thread_runs_and_stops():
    from generalsreinforcementlearning_amd.env_pool import DeviceReplayBuffer
    from generalsreinforcementlearning_amd.vector_env import GeneralsVecEnv
    B = 512
    buf = DeviceReplayBuffer(50000)
    pool = ParallelVecEnvPool(B, lambda n: GeneralsVecEnv(n, board_width=10, board_height=10, max_players=2, max_turns=40, seed=1, board_pool=64,
                                                         device_outputs=True),
                              _first_valid_device, buf, max_steps_per_episode=25, batched_actions=True)
    pool.start()
    t0 = time.time()
    seen = []
    while pool.total_episodes < 4 * B and time.time() - t0 < 60:
        seen += pool.pop_episode_results()                  # read while the collector runs
        time.sleep(0.02)
    pool.stop(join_timeout=10.0)
    seen += pool.pop_episode_results()
    assert pool.alive_workers == 0 and pool.total_episodes == len(seen) >= 4 * B
    assert pool.total_env_steps == buf.total_pushed == sum(l for _, l, _ in seen) + int(pool._dc.episode_length.sum())
    assert {w for _, _, w in seen} == set(range(B)) and max(l for _, l, _ in seen) <= 25
    # a restart opens a new env; what was counted stays counted (like the host form)
    before, pushed = pool.total_episodes, buf.total_pushed
    pool.start()
    t0 = time.time()
    while pool.total_episodes < before + B and time.time() - t0 < 60:
        time.sleep(0.02)
    pool.stop(join_timeout=10.0)
    assert pool.total_episodes >= before + B and buf.total_pushed > pushed


def test_push_batch_is_k_pushes():
    """Every (capacity, batch sizes) case: the ring after push_batch calls equals the ring after the same transitions pushed
    one by one - runs that fit, runs that wrap, and batches larger than the ring."""
    rng = np.random.default_rng(4)
    for cap, sizes in ((7, [3, 3, 3, 7, 1, 20, 2]), (5, [5, 5]), (16, [1, 30, 16, 15, 2]), (3, [10])):
        one, many, t = ReplayBuffer(cap), ReplayBuffer(cap), 0
        for k in sizes:
            s = rng.random((k, 2, 3), dtype=np.float32)
            ns = rng.random((k, 2, 3), dtype=np.float32)
            a, r, d = np.arange(t, t + k), rng.random(k), rng.random(k) < 0.3
            t += k
            many.push_batch(s, a, r, ns, d)
            for i in range(k):
                one.push(s[i], int(a[i]), float(r[i]), ns[i], bool(d[i]))
            assert (many._cursor, many._size, many._pushed) == (one._cursor, one._size, one._pushed)
            n = one._size
            order = (one._cursor - n + np.arange(n)) % cap if n == cap else np.arange(n)
            for f in ("_state", "_next", "_action", "_reward", "_done"):
                assert np.array_equal(getattr(many, f)[order], getattr(one, f)[order]), (cap, k, f)


@pytest.mark.gpu
def test_reference_shaped_training_code_runs_on_the_drop_in_names():
    """`from ...generals_gym import GeneralsEnv, ParallelEnvPool, ReplayBuffer` with the reference's own call shapes
    (python/train_dqn_parallel.py:72-121, python/test_parallel_env.py): env_factory(worker_id) -> one GeneralsEnv,
    action_fn(state, valid_mask, worker_id, rng) -> int, buffer.sample -> list of tuples."""
    from generalsreinforcementlearning_amd.generals_gym import GeneralsEnv, ParallelEnvPool, ReplayBuffer as RB
    made, asked = [], []

    def make_env(worker_id):
        made.append(worker_id)
        return GeneralsEnv(server_address="localhost:50051", board_width=8, board_height=8, max_players=2, fog_of_war=True, max_turns=30,
                           collect_experiences=False)

    def action_fn(state, valid_mask, worker_id, rng):
        assert state.shape == (9, 8, 8) and valid_mask.shape == (320,) and 0 <= worker_id < 16
        asked.append(worker_id)
        valid = np.where(valid_mask)[0]
        return int(rng.choice(list(valid))) if len(valid) else 0

    buffer = RB(5000)
    pool = ParallelEnvPool(num_envs=16, env_factory=make_env, action_fn=action_fn, replay_buffer=buffer, max_steps_per_episode=12, seed=7)
    assert pool.env_factory is make_env
    pool.start()
    t0 = time.time()
    while buffer.total_pushed < 1500 and time.time() - t0 < 60:
        time.sleep(0.05)
    pool.stop()
    assert made == [0] and pool.total_env_steps == buffer.total_pushed >= 1500 and pool.total_episodes >= 16 * 5
    assert set(asked) == set(range(16))
    batch = buffer.sample(32)
    states, actions, rewards, next_states, dones = zip(*batch)
    assert np.array(states).shape == (32, 9, 8, 8) and all(isinstance(a, int) for a in actions) and all(isinstance(d, bool) for d in dones)
    res = pool.pop_episode_results()
    assert len(res) == pool.total_episodes and max(l for _, l, _ in res) <= 12
    wrong = ParallelEnvPool(2, lambda w: object(), action_fn, RB(10), max_env_retries=1)
    with pytest.raises(RuntimeError) as err:                 # like a factory that fails: retried, then the pool gives up
        wrong.collect(1)
    assert isinstance(err.value.__cause__, TypeError)


@pytest.mark.gpu
def test_device_replay_buffer_draws_distinct_slots_from_a_big_ring():
    """Past 65,536 transitions a small draw no longer permutes the whole ring: it draws with replacement, drops repeats and
    tops up - still batch_size DISTINCT slots, all inside the filled part, different from draw to draw."""
    import torch
    from generalsreinforcementlearning_amd.env_pool import DeviceReplayBuffer
    buf = DeviceReplayBuffer(200_000)
    n = 150_000
    s = torch.arange(n, dtype=torch.float32, device="cuda").reshape(n, 1, 1, 1).expand(n, 2, 2, 2).contiguous()
    buf.push_batch(s, torch.arange(n), torch.zeros(n), s + 0.5, torch.zeros(n, dtype=torch.bool))
    assert len(buf) == n and buf.total_pushed == n
    buf.manual_seed(5)
    a = buf.sample_indices(4096)
    b = buf.sample_indices(4096)
    for idx in (a, b):
        assert idx.numel() == 4096 and idx.unique().numel() == 4096 and int(idx.min()) >= 0 and int(idx.max()) < n
    assert not torch.equal(a, b) and int(a.max()) > n // 2              # spread over the ring, not a prefix
    st, ac, rw, ns, dn = buf.sample_arrays(512)
    assert torch.equal(st[:, 0, 0, 0].long(), ac) and torch.equal(ns, st + 0.5)      # a row's fields belong together
    with pytest.raises(ValueError):
        buf.sample_indices(n + 1)
