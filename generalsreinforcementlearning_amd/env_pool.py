"""ParallelVecEnvPool + ReplayBuffer — the collection surface `train_dqn_parallel.py` consumes, over ONE vector env.

Reference: python/generals_gym/vector_env.py:28-192 (ParallelEnvPool: N GeneralsEnv instances, one worker thread each,
every worker running whole episodes and pushing (state, action, reward, next_state, done) into a shared buffer) and
python/generals_gym/replay_buffer.py:13-55 (thread-safe ring).  Same constructor keywords, same properties
(`total_env_steps`, `total_episodes`, `alive_workers`), same `start / stop / pop_episode_results`, same per-worker
behaviour - the sequence of transitions worker w pushes and the (episode_reward, episode_length, worker_id) results it
reports are those of the reference's worker w given the same env behaviour and the same `action_fn`
(tests/test_env_pool.py replays fixtures recorded from the reference's own classes).

What differs, by construction of a vector env:
  * `env_factory(num_envs)` is called ONCE and returns the vector env (GeneralsVecEnv, numpy mode) - the reference calls
    `env_factory(worker_id)` once per worker;
  * one collector thread steps all "workers" (= env indices) in lock-step instead of N threads doing gRPC round trips;
  * an episode that ends (terminated / truncated, or cut at `max_steps_per_episode` - the pool then asks the env to
    re-deal that board with `force_reset`) starts its successor on the env's next step, which returns the new episode's
    first observation with info["reset"] set; that step is the worker's `env.reset()` and is not a transition;
  * with a `DeviceReplayBuffer` (and `GeneralsVecEnv(device_outputs=True)`, `batched_actions=True`) the whole loop is
    resident on the GPU: `action_fn(states, valid_masks, None, torch_generator) -> CUDA int64 actions[num_envs]`, transitions
    appended to the ring in HBM and episode results logged by `gvec_pool_collect` - the host only enqueues launches;
  * `batched_actions=True`: `action_fn(states[k], valid_masks[k], worker_ids[k], rngs) -> actions[k]` for the k workers
    that play this step (rngs: the list of all workers' RNGs, indexed by worker id) - one policy forward for all envs;
    the default keeps the reference's per-env signature `(state, valid_mask, worker_id, rng) -> int`
    with one private `random.Random(seed * 1000 + worker_id)` per worker (vector_env.py:138).
"""
import logging
import random
import threading
import time

import numpy as np

logger = logging.getLogger(__name__)


class ReplayBuffer:
    """Thread-safe ring-buffer replay memory (replay_buffer.py:13-55): `push` evicts the oldest transition when full,
    `sample` draws uniformly without replacement with the module-level `random` (same indices as the reference's
    `random.sample(list, k)` for the same seed: the draw depends on the length only), `total_pushed` is the monotonic
    env-step counter, `len()` the fill.  Stored as arrays (one slab per field, allocated at the first push) rather than a
    list of tuples, so `push_batch` writes a whole vector step under one lock and `sample_arrays` hands a learner stacked
    batches without a Python loop."""

    def __init__(self, capacity):
        if capacity <= 0:
            raise ValueError(f"capacity must be positive, got {capacity}")   # replay_buffer.py:22-23
        self.capacity = int(capacity)
        self._state = self._next = self._action = self._reward = self._done = None
        self._size = 0          # transitions held
        self._cursor = 0        # slot the next transition goes to
        self._pushed = 0        # every push since construction
        self._guard = threading.Lock()

    def _alloc(self, state):
        s = np.asarray(state)
        self._state = np.empty((self.capacity,) + s.shape, s.dtype)
        self._next = np.empty((self.capacity,) + s.shape, s.dtype)
        self._action = np.empty(self.capacity, np.int64)
        self._reward = np.empty(self.capacity, np.float64)
        self._done = np.empty(self.capacity, bool)

    def push(self, state, action, reward, next_state, done):
        with self._guard:
            if self._state is None:
                self._alloc(state)
            slot = self._cursor
            self._state[slot], self._next[slot] = state, next_state
            self._action[slot], self._reward[slot], self._done[slot] = action, reward, done
            self._cursor = (slot + 1) % self.capacity                       # the oldest slot is the next to go (:31-36)
            self._size = min(self._size + 1, self.capacity)
            self._pushed += 1

    def push_batch(self, states, actions, rewards, next_states, dones):
        """k transitions in order (equivalent to k `push` calls) under one lock."""
        k = len(actions)
        if k == 0:
            return
        with self._guard:
            if self._state is None:
                self._alloc(states[0])
            first = self._cursor
            if k > self.capacity:                                            # only the last `capacity` survive, as with k pushes
                keep = slice(k - self.capacity, k)
                first = (first + k - self.capacity) % self.capacity
                states, actions, rewards, next_states, dones = states[keep], actions[keep], rewards[keep], next_states[keep], dones[keep]
            n = len(actions)
            head = min(n, self.capacity - first)                             # two contiguous runs (the ring wraps at most once): memcpy, not a gather
            for dst, src in ((self._state, states), (self._next, next_states), (self._action, actions), (self._reward, rewards), (self._done, dones)):
                dst[first:first + head] = src[:head]
                if head < n:
                    dst[:n - head] = src[head:]
            self._cursor = (self._cursor + k) % self.capacity
            self._size = min(self._size + k, self.capacity)
            self._pushed += k

    def _item(self, i):
        return (self._state[i], int(self._action[i]), float(self._reward[i]), self._next[i], bool(self._done[i]))

    def sample(self, batch_size):
        """List of (state, action, reward, next_state, done) tuples, like the reference (:40-43); ValueError when the
        buffer holds fewer than batch_size transitions (random.sample's own)."""
        with self._guard:
            return [self._item(i) for i in random.sample(range(self._size), batch_size)]

    def sample_arrays(self, batch_size):
        """The same draw as stacked arrays: (states, actions, rewards, next_states, dones)."""
        with self._guard:
            idx = np.asarray(random.sample(range(self._size), batch_size), np.int64)
            return self._state[idx], self._action[idx], self._reward[idx], self._next[idx], self._done[idx]

    @property
    def total_pushed(self):
        with self._guard:
            return self._pushed

    def __len__(self):
        with self._guard:
            return self._size


class DeviceReplayBuffer:
    """The replay ring resident in HBM: what `ReplayBuffer` is to a host collector, for a pool that never leaves the GPU
    (`ParallelVecEnvPool` over a `GeneralsVecEnv(device_outputs=True)`).  Transitions are appended by `gvec_pool_collect`
    (one wavefront moves one row) straight from the gym kernel's output buffers; `sample_arrays` gathers a batch into CUDA
    tensors a learner consumes in place.  288 GB of HBM hold 17 million 15x15 transitions (2 x 8,100 B of observation each).
    Same semantics as replay_buffer.py:13-55 - the oldest transition is overwritten once `capacity` is reached, `sample`
    draws uniformly without replacement and raises ValueError when fewer than batch_size are held, `total_pushed` counts
    every push - with one difference: the draw comes from a torch generator on the device, not from `random`."""

    def __init__(self, capacity, device=0):
        if capacity <= 0:
            raise ValueError(f"capacity must be positive, got {capacity}")
        import torch
        self._t = torch
        self.capacity = int(capacity)
        self.device = torch.device("cuda", device)
        self.counters = torch.zeros(4, dtype=torch.int64, device=self.device)     # cursor, size, total pushed, 0
        self.state = self.next_state = self.action = self.reward = self.done = None
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(0)
        self._guard = threading.Lock()

    def allocate(self, obs_shape):
        """The five slabs of the ring (at the first push; a collector calls it with the env's observation shape)."""
        if self.state is None:
            t, dev, cap = self._t, self.device, self.capacity
            self.obs_shape = tuple(obs_shape)
            self.state = t.empty((cap,) + self.obs_shape, dtype=t.float32, device=dev)
            self.next_state = t.empty((cap,) + self.obs_shape, dtype=t.float32, device=dev)
            self.action = t.empty(cap, dtype=t.int64, device=dev)
            self.reward = t.empty(cap, dtype=t.float64, device=dev)
            self.done = t.empty(cap, dtype=t.bool, device=dev)
        return self

    def manual_seed(self, seed):
        self._gen.manual_seed(int(seed))

    def push_batch(self, states, actions, rewards, next_states, dones):
        """k transitions in order, for a learner that pushes by itself (the pool appends through gvec_pool_collect)."""
        t = self._t
        dev = self.device
        actions = t.as_tensor(actions, dtype=t.int64, device=dev).reshape(-1)
        k = int(actions.numel())
        if k == 0:
            return
        states = t.as_tensor(states, dtype=t.float32, device=dev)
        next_states = t.as_tensor(next_states, dtype=t.float32, device=dev)
        with self._guard:
            self.allocate(states.shape[1:])
            cursor, size, pushed = (int(v) for v in self.counters[:3].tolist())
            lo = max(0, k - self.capacity)                                    # only the last `capacity` survive, as with k pushes
            idx = (cursor + t.arange(lo, k, device=dev)) % self.capacity
            self.state[idx], self.next_state[idx] = states[lo:], next_states[lo:]
            self.action[idx] = actions[lo:]
            self.reward[idx] = t.as_tensor(rewards, dtype=t.float64, device=dev).reshape(-1)[lo:]
            self.done[idx] = t.as_tensor(dones, dtype=t.bool, device=dev).reshape(-1)[lo:]
            self.counters[:3] = t.tensor([(cursor + k) % self.capacity, min(size + k, self.capacity), pushed + k], dtype=t.int64)

    def push(self, state, action, reward, next_state, done):
        t = self._t
        self.push_batch(t.as_tensor(state)[None], [action], [reward], t.as_tensor(next_state)[None], [done])

    def sample_indices(self, batch_size):
        """batch_size distinct slots, uniformly over the transitions held (random.sample's contract, replay_buffer.py:40-43)."""
        from ._sampling import distinct_indices
        return distinct_indices(self._t, len(self), batch_size, self.device, self._gen)

    def sample_arrays(self, batch_size):
        """(states, actions, rewards, next_states, dones) as CUDA tensors.  The draw and its five gathers are enqueued under
        the lock a collector's launches take too, so no vector step lands between them: a drawn slot's fields belong to ONE
        transition even while the ring is being overwritten (the reference samples under its lock as well, :40-43)."""
        with self._guard:
            idx = self.sample_indices(batch_size)
            return self.state[idx], self.action[idx], self.reward[idx], self.next_state[idx], self.done[idx]

    def sample(self, batch_size):
        """The reference's return type - a list of (state, action, reward, next_state, done) tuples - on the host."""
        s, a, r, n, d = (x.cpu().numpy() for x in self.sample_arrays(batch_size))
        return [(s[i], int(a[i]), float(r[i]), n[i], bool(d[i])) for i in range(len(a))]

    @property
    def total_pushed(self):
        return int(self.counters[2])

    def __len__(self):
        return int(self.counters[1])


class _DeviceCollector:
    """The device-side state of a pool whose env, policy and buffer all live on the GPU, and the one call per vector step
    that advances it (gvec_pool_collect)."""

    def __init__(self, env, buffer, max_steps_per_episode, result_capacity, prior=None):
        import ctypes
        import torch
        from ._lib import CollectArgs, check
        self._C, self._t, self._check = ctypes, torch, check
        self.env, self.buffer = env, buffer
        self.L = env.engine.L
        dev = env._dev
        if buffer.device != dev:
            raise ValueError(f"replay buffer on {buffer.device}, env on {dev}")
        if buffer.capacity < env.num_envs:
            raise ValueError(f"a DeviceReplayBuffer must hold at least one vector step: capacity {buffer.capacity} < {env.num_envs} envs")
        buffer.allocate(env.single_observation_shape)
        n = env.num_envs
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        self.episode_reward, self.episode_length = z(n, torch.float64), z(n, torch.int64)
        self.result_capacity = int(result_capacity)
        if prior is not None and prior.result_capacity == self.result_capacity and prior.counters.device == dev:
            # the env was re-created (a failed step, a restart): episodes counted and results not yet read stay
            self.result_reward, self.result_length, self.result_worker, self.counters = (prior.result_reward, prior.result_length,
                                                                                         prior.result_worker, prior.counters)
        else:
            self.result_reward, self.result_length, self.result_worker = (z(self.result_capacity, torch.float64), z(self.result_capacity, torch.int32),
                                                                          z(self.result_capacity, torch.int32))
            self.counters = z(4, torch.int64)          # episodes, results held, results dropped, 0
        self.scratch = z((int(self.L.gvec_pool_collect_scratch_bytes(n)) + 7) // 8, torch.int64)
        a = self.args = CollectArgs()
        a.num_envs, a.obs_floats, a.max_steps_per_episode = n, int(np.prod(env.single_observation_shape)), int(max_steps_per_episode)
        a.capacity, a.result_capacity = buffer.capacity, self.result_capacity
        for name, tensor in (("ring_state", buffer.state), ("ring_next_state", buffer.next_state), ("ring_action", buffer.action),
                             ("ring_reward", buffer.reward), ("ring_done", buffer.done), ("ring_counters", buffer.counters),
                             ("episode_reward", self.episode_reward), ("episode_length", self.episode_length),
                             ("result_reward", self.result_reward), ("result_length", self.result_length), ("result_worker", self.result_worker),
                             ("pool_counters", self.counters), ("scratch", self.scratch)):
            setattr(a, name, tensor.data_ptr())
        self._device_index = dev.index

    def restart(self):
        self.episode_reward.zero_()
        self.episode_length.zero_()

    def collect(self, state, actions, next_state, reward, terminated, truncated, was_reset, needs_reset):
        a = self.args
        a.state, a.next_state, a.action, a.reward = state.data_ptr(), next_state.data_ptr(), actions.data_ptr(), reward.data_ptr()
        a.terminated, a.truncated, a.was_reset, a.needs_reset = terminated.data_ptr(), truncated.data_ptr(), was_reset.data_ptr(), needs_reset.data_ptr()
        stream = self._t.cuda.current_stream(self.env._dev).cuda_stream
        with self.buffer._guard:                  # ordered against a learner's sample_arrays / push_batch on the same stream
            self._check(self.L.gvec_pool_collect(self._device_index, stream, self._C.byref(a)), "gvec_pool_collect")

    def pop_results(self):
        held = int(self.counters[1])
        if held == 0:
            return []
        r, l, w = self.result_reward[:held].tolist(), self.result_length[:held].tolist(), self.result_worker[:held].tolist()
        self.counters[1] = 0
        return list(zip(r, l, w))


class ParallelVecEnvPool:
    """vector_env.py:28-192 over a vector env; see the module docstring for what is kept and what differs."""

    def __init__(self, num_envs, env_factory, action_fn, replay_buffer, max_steps_per_episode=200, max_env_retries=3, seed=42,
                 batched_actions=False, retry_sleep_s=2.0, result_capacity=1 << 20):
        # the reference's public attributes (vector_env.py:45-51)
        (self.num_envs, self.env_factory, self.action_fn, self.replay_buffer, self.max_steps_per_episode, self.max_env_retries,
         self.seed) = num_envs, env_factory, action_fn, replay_buffer, max_steps_per_episode, max_env_retries, seed
        self.batched_actions, self.retry_sleep_s = bool(batched_actions), retry_sleep_s
        # a DeviceReplayBuffer switches the pool to its resident form: env (device_outputs), policy and ring on the GPU
        self.on_device = isinstance(replay_buffer, DeviceReplayBuffer)
        self.result_capacity = int(result_capacity)
        self._dc = None
        self._step_lock = threading.Lock()      # device form: a vector step's launches vs. a reader of the result log
        # one collector instead of one thread per env
        self._halt = threading.Event()
        self._collector = None
        self._tally = threading.Lock()          # guards the three fields below
        self._episodes_done = 0
        self._live = 0
        self._finished = []                     # (episode_reward, episode_length, worker_id) since the last pop
        # private RNG per worker (vector_env.py:136-138): shared module-level RNGs would correlate exploration between workers
        self._rngs = [random.Random(self.seed * 1000 + w) for w in range(num_envs)]
        self._env = None
        self._state = self._mask = None
        self._ep_reward = np.zeros(num_envs, np.float64)
        self._ep_length = np.zeros(num_envs, np.int64)
        self._starting = np.zeros(num_envs, bool)   # the worker's next vector step is its env.reset()

    # ---- the reference's public surface (vector_env.py:62-112) ----------------------------------------------
    def start(self):
        """Starts the collector (a daemon thread) for all environments."""
        if self._collector is not None:
            raise RuntimeError("Pool already started")
        self._halt.clear()
        with self._tally:
            self._live = self.num_envs
        self._collector = threading.Thread(target=self._worker_loop, name="env-worker-vec", daemon=True)
        self._collector.start()
        logger.info("collector started for %d envs", self.num_envs)

    def stop(self, join_timeout=10.0):
        """Asks the collector to finish its current vector step and waits for it; a straggler is logged, not raised
        (the thread is a daemon)."""
        self._halt.set()
        t, self._collector = self._collector, None
        if t is not None:
            t.join(timeout=join_timeout)
            if t.is_alive():
                logger.warning("collector %s still running after %.1fs", t.name, join_timeout)

    @property
    def total_env_steps(self):
        return self.replay_buffer.total_pushed      # one transition pushed == one environment step

    @property
    def total_episodes(self):
        if self._dc is not None:
            return int(self._dc.counters[0])
        with self._tally:
            return self._episodes_done

    @property
    def alive_workers(self):
        with self._tally:
            return self._live

    def pop_episode_results(self):
        """Finished-episode results since the last call, oldest first."""
        if self._dc is not None:
            with self._step_lock:               # between two vector steps: the log is read and emptied in stream order
                return self._dc.pop_results()
        with self._tally:
            out, self._finished = self._finished, []
        return out

    # ---- collection ---------------------------------------------------------------------------------------------
    def _create_env(self, old_env=None):
        """Opens the vector env, closing a broken predecessor first; up to max_env_retries tries, retry_sleep_s apart
        (the reference's pattern, vector_env.py:114-134)."""
        if old_env is not None:
            try:
                old_env.close()
            except Exception:  # noqa: BLE001 - it is being replaced anyway
                pass
        tries = 0
        while True:
            tries += 1
            try:
                env = self._open_env()
            except Exception as e:  # noqa: BLE001
                logger.warning("opening the vector environment failed (%d of %d): %s", tries, self.max_env_retries, e)
                if tries >= self.max_env_retries:
                    raise RuntimeError("failed to create the vector environment") from e
                time.sleep(self.retry_sleep_s)
            else:
                logger.info("vector environment open")
                return env

    def _open_env(self):
        return self.env_factory(self.num_envs)

    def _begin(self):
        """Every worker's first `env.reset()` (vector_env.py:166-167)."""
        self._state, info = self._env.reset()
        self._mask = info.get("valid_actions_mask")
        if self.on_device:
            if not getattr(self._env, "device_outputs", False) or not self.batched_actions:
                raise ValueError("a DeviceReplayBuffer needs a vector env with device_outputs=True and batched_actions=True")
            if self._dc is None or self._dc.env is not self._env:
                self._dc = _DeviceCollector(self._env, self.replay_buffer, self.max_steps_per_episode, self.result_capacity, prior=self._dc)
                import torch
                self._generator = torch.Generator(device=self._env._dev)
                self._generator.manual_seed(self.seed)
            self._dc.restart()
            return
        if self._mask is None:
            self._mask = np.ones((self.num_envs, self._env.single_action_n), bool)
        self._ep_reward[:] = 0.0
        self._ep_length[:] = 0
        self._starting[:] = False

    def _actions(self):
        """The policy is asked only for the workers that will play: a worker whose next step is its `env.reset()` chooses
        nothing (and draws nothing from its RNG), as in the reference's loop."""
        acts = np.zeros(self.num_envs, np.int64)
        idx = np.flatnonzero(~self._starting)
        if len(idx) == 0:
            return acts
        if self.batched_actions:
            acts[idx] = np.asarray(self.action_fn(self._state[idx], self._mask[idx], idx, self._rngs), np.int64).reshape(len(idx))
        else:
            for w in idx:
                acts[w] = self.action_fn(self._state[w], self._mask[w], int(w), self._rngs[w])
        return acts

    def _collect_step(self):
        """One vector step = one iteration of every worker's `_run_episode` loop (vector_env.py:172-192)."""
        if self.on_device:
            return self._collect_step_device()
        state = self._state
        actions = self._actions()
        next_state, reward, terminated, truncated, info = self._env.step(actions)
        reward = np.asarray(reward, np.float64)
        done = np.asarray(terminated, bool) | np.asarray(truncated, bool)
        fresh = np.asarray(info.get("reset", np.zeros(self.num_envs, bool)), bool)   # this step was the worker's env.reset()
        live = ~fresh
        idx = np.flatnonzero(live)
        if len(idx):
            if hasattr(self.replay_buffer, "push_batch"):
                if len(idx) == self.num_envs and isinstance(self.replay_buffer, ReplayBuffer):
                    # the usual step: no worker is re-dealing, nothing to pick out - and this package's ring copies what it is
                    # given, so the env's own (soon reused) arrays can be handed over as they are
                    self.replay_buffer.push_batch(state, actions, reward, next_state, done)
                else:
                    self.replay_buffer.push_batch(np.array(state[idx]), actions[idx], reward[idx], np.array(next_state[idx]), done[idx])
            else:
                for w in idx:
                    self.replay_buffer.push(np.array(state[w]), int(actions[w]), float(reward[w]), np.array(next_state[w]), bool(done[w]))
        self._ep_reward[live] += reward[live]
        self._ep_length[live] += 1
        over = live & (done | (self._ep_length >= self.max_steps_per_episode))
        if over.any():
            ended = [(float(self._ep_reward[w]), int(self._ep_length[w]), int(w)) for w in np.flatnonzero(over)]
            with self._tally:
                self._episodes_done += len(ended)
                self._finished += ended
            cut = over & ~done
            if cut.any():
                self._env.force_reset(cut)      # the reference's next `env.reset()`: the env's own flags do not say so
            self._ep_reward[over] = 0.0
            self._ep_length[over] = 0
        self._starting = over
        self._state = next_state
        m = info.get("valid_actions_mask")
        self._mask = m if m is not None else np.ones((self.num_envs, self._env.single_action_n), bool)

    def _collect_step_device(self):
        """The same iteration with nothing on the host: the policy maps CUDA tensors to a CUDA int64 tensor of actions (it is
        asked for every worker; what it answers for a worker whose step is its `env.reset()` is ignored by the env), the gym
        kernel plays the step, gvec_pool_collect appends the transitions to the ring, keeps the episode accumulators, logs
        finished episodes and raises the env's needs_reset for episodes cut at `max_steps_per_episode`.  No synchronisation:
        the host only enqueues."""
        env = self._env
        with self._step_lock:
            state = self._state
            actions = self.action_fn(state, self._mask, None, self._generator)
            next_state, reward, terminated, truncated, info = env.step(actions)
            self._dc.collect(state, env.last_actions, next_state, reward, terminated, truncated, info["reset"], env.needs_reset_buffer())
            self._state, self._mask = next_state, info["valid_actions_mask"]

    def collect(self, steps):
        """Synchronous form for trainers and tests that own the loop: `steps` vector steps in the caller's thread."""
        if self._env is None:
            self._env = self._create_env()
            self._begin()
        for _ in range(steps):
            self._collect_step()

    def _worker_loop(self):
        """The collector thread: vector steps until stop(); a step that raises costs the running episodes (like a failed
        worker episode in the reference) and the env is reopened; when reopening fails for good the collector ends and
        alive_workers drops to 0."""
        try:
            if self._env is None:
                self._env = self._create_env()
                self._begin()
            while not self._halt.is_set():
                try:
                    self._collect_step()
                except Exception as e:  # noqa: BLE001
                    if self._halt.is_set():
                        break
                    logger.warning("vector step failed: %s", e)
                    self._env = self._create_env(old_env=self._env)
                    self._begin()
        except Exception as e:  # noqa: BLE001
            logger.error("collector ends: %s", e)
        finally:
            env, self._env = self._env, None
            if env is not None:
                try:
                    env.close()
                except Exception:  # noqa: BLE001
                    pass
            with self._tally:
                self._live = 0
