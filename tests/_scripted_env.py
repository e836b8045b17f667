"""A deterministic scripted environment in two shapes - one gym-style env per worker (what the reference's
ParallelEnvPool drives) and one vector env for all workers (what ParallelVecEnvPool drives) - defined by ONE script, so a
transition sequence recorded from the reference pool can be replayed against the vector pool.

Script (worker w, episode j, t = steps taken in the episode):
  observation  (9, 2, 2) float32, zeros except [0,0,0] = w, [0,0,1] = j, [0,1,0] = t, [0,1,1] = last action (-1 at reset)
  valid mask   12 actions, action i valid iff (i + w + j + t) % 3 != 0
  step(a)      t += 1; reward = ((7w + 5j + 3t + a) % 11 - 5) / 4; episode length L = 3 + (5w + 3j) % 9:
               terminated at t == L unless L == 7, which is a TRUNCATION at t == 7 (the env's own turn limit)"""
import numpy as np

N_ACTIONS = 12


def obs_of(w, j, t, last):
    o = np.zeros((9, 2, 2), np.float32)
    o[0, 0, 0], o[0, 0, 1], o[0, 1, 0], o[0, 1, 1] = w, j, t, last
    return o


def mask_of(w, j, t):
    return np.array([(i + w + j + t) % 3 != 0 for i in range(N_ACTIONS)])


def outcome(w, j, t_after, a):
    reward = ((7 * w + 5 * j + 3 * t_after + int(a)) % 11 - 5) / 4.0
    L = 3 + (5 * w + 3 * j) % 9
    terminated = t_after == L and L != 7
    truncated = t_after == L and L == 7
    return reward, terminated, truncated


class _Space:
    n = N_ACTIONS


class ScriptedEnv:
    """The single-env shape (reset() -> (obs, info); step(a) -> (obs, reward, terminated, truncated, info)).
    `gate(worker, episode)` is called at the top of reset(): the fixture generator parks a worker there once it has played
    its quota of episodes."""
    action_space = _Space()

    def __init__(self, worker, gate=None):
        self.w, self.j, self.t, self.gate = worker, -1, 0, gate

    def reset(self):
        self.j += 1
        self.t = 0
        if self.gate:
            self.gate(self.w, self.j)
        return obs_of(self.w, self.j, 0, -1), {"valid_actions_mask": mask_of(self.w, self.j, 0)}

    def step(self, a):
        self.t += 1
        r, term, trunc = outcome(self.w, self.j, self.t, a)
        return obs_of(self.w, self.j, self.t, a), r, term, trunc, {"valid_actions_mask": mask_of(self.w, self.j, self.t)}

    def close(self):
        pass


class ScriptedVecEnv:
    """The vector shape with GeneralsVecEnv's conventions: an env whose episode ended (by its own flags, or through
    force_reset) spends its next step starting the next episode - info["reset"] set, reward 0, the action ignored."""

    def __init__(self, num_envs):
        self.num_envs, self.single_action_n = num_envs, N_ACTIONS
        self.j = np.zeros(num_envs, np.int64)
        self.t = np.zeros(num_envs, np.int64)
        self.needs_reset = np.zeros(num_envs, bool)

    def _view(self, last):
        obs = np.stack([obs_of(w, self.j[w], self.t[w], last[w]) for w in range(self.num_envs)])
        mask = np.stack([mask_of(w, self.j[w], self.t[w]) for w in range(self.num_envs)])
        return obs, mask

    def reset(self):
        self.j[:] = 0
        self.t[:] = 0
        self.needs_reset[:] = False
        obs, mask = self._view(np.full(self.num_envs, -1))
        return obs, {"valid_actions_mask": mask}

    def force_reset(self, env_mask):
        self.needs_reset |= np.asarray(env_mask, bool)

    def step(self, actions):
        n = self.num_envs
        resetting = self.needs_reset.copy()
        reward, term, trunc, last = np.zeros(n), np.zeros(n, bool), np.zeros(n, bool), np.full(n, -1)
        for w in range(n):
            if resetting[w]:
                self.j[w] += 1
                self.t[w] = 0
            else:
                self.t[w] += 1
                reward[w], term[w], trunc[w] = outcome(w, self.j[w], self.t[w], actions[w])
                last[w] = actions[w]
        self.needs_reset = term | trunc
        obs, mask = self._view(last)
        return obs, reward, term, trunc, {"valid_actions_mask": mask, "reset": resetting}

    def close(self):
        pass


def random_action_fn(state, valid_mask, worker_id, rng):
    """The reference test's policy (python/test_parallel_env.py:76-80)."""
    valid = np.where(valid_mask)[0]
    return int(rng.choice(list(valid))) if len(valid) else 0
