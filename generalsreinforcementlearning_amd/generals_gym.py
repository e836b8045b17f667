"""`generals_gym` under its own names - for code written against the reference's Python package
(python/generals_gym/__init__.py exports GeneralsEnv, ParallelEnvPool, ReplayBuffer):

    from generalsreinforcementlearning_amd.generals_gym import GeneralsEnv, ParallelEnvPool, ReplayBuffer

is the only line `python/train_dqn_parallel.py`-style code has to change.  `GeneralsEnv` and `ReplayBuffer` are this
package's classes of those names; `ParallelEnvPool` takes the reference's constructor exactly - `env_factory(worker_id)`
returning ONE env, `action_fn(state, valid_mask, worker_id, rng) -> int` per env (vector_env.py:36-61) - and still runs
all workers as one vector env: it asks the factory for worker 0's env, reads the board configuration off it and opens
a GeneralsVecEnv of `num_envs` boards with that configuration (one HIP launch per vector step instead of one gRPC
session per worker).  Code that wants one policy call per vector step uses env_pool.ParallelVecEnvPool directly.
"""
from .env_pool import ParallelVecEnvPool, ReplayBuffer
from .vector_env import GeneralsEnv, GeneralsVecEnv

__all__ = ["GeneralsEnv", "ParallelEnvPool", "ReplayBuffer"]


class ParallelEnvPool(ParallelVecEnvPool):
    def __init__(self, num_envs, env_factory, action_fn, replay_buffer, max_steps_per_episode=200, max_env_retries=3, seed=42):
        super().__init__(num_envs, env_factory, action_fn, replay_buffer, max_steps_per_episode=max_steps_per_episode,
                         max_env_retries=max_env_retries, seed=seed, batched_actions=False)

    def _open_env(self):
        probe = self.env_factory(0)                 # worker 0's env, as the reference's worker 0 would get it
        try:
            if not isinstance(probe, GeneralsEnv):
                raise TypeError(f"env_factory returned {type(probe).__name__}: this pool fuses generals_gym.GeneralsEnv instances into one vector "
                                "env (for other env types use env_pool.ParallelVecEnvPool with a vector-env factory)")
            if probe.opponent_agent is not None:
                raise ValueError("opponent_agent acts on one env's proto state; a pool of such envs is not fused - step GeneralsVecEnv with "
                                 "other_actions=... instead")
            cfg = dict(board_width=probe.board_width, board_height=probe.board_height, max_players=probe.max_players,
                       fog_of_war=probe.fog_of_war, max_turns=probe.max_turns, device=probe.device)
        finally:
            if hasattr(probe, "close"):
                probe.close()
        return GeneralsVecEnv(self.num_envs, seed=self.seed, **cfg)
