"""GeneralsVecEnv — drop-in *vector* form of the reference's gym environment.

Reference: python/generals_gym/generals_env.py (GeneralsEnv) and vector_env.py (ParallelEnvPool,
which runs N GeneralsEnv instances from N threads, each doing two gRPC round trips and a 50 ms
sleep per step).  Here the same per-env interface is served for B boards by ONE HIP launch per
step (gvec_gym_step):

  * observation (9, H, W) float32            generals_env.py:111-116, 291-342
  * action space Discrete(board_size * 5)    generals_env.py:118-120, 344-387 (idx*5 + {up,right,down,left,half})
  * action decoding incl. the half-move quirk generals_env.py:389-441
  * reward                                    generals_env.py:499-561
  * terminated / truncated / info keys        generals_env.py:272-289

What the learner sees is the *proto* view the gRPC server would send for its player token:
fog rules of internal/grpc/gameserver/server.go:556-582 (hidden tile -> type NORMAL, owner -1,
army 0; fogged tile -> type kept, owner / army hidden) and the PlayerState fields of
server.go:526-553 (army_count = Player.ArmyCount, tile_count = len(OwnedTiles), status by Alive).

There is one execution path: the gym kernels behind the C ABI.  (The readable numpy restatement they
are tested against lives in tests/_gym_reference.py; it was a third mode of this class up to round 2.)

Deliberate differences (documented in DESIGN.md): opponents are by default the on-device random agent
(uniform over legal moves, 30 % half moves, 10 % no-op) instead of Python's `random.choice` over
full moves - or whatever the caller supplies per step (`step(actions, other_actions=...)`: an opponent
policy, self-play; GeneralsEnv's `opponent_agent`); finished / truncated envs are re-dealt on their next step ("next-step" autoreset)
because a vector env cannot wait for a per-env reset() call.
"""
import numpy as np

from .vec_engine import VecEngine
from ._lib import check


class GeneralsVecEnv:
    """B GeneralsEnv instances behind the (gymnasium-style) vector API:
    reset() -> (obs, info);  step(actions[B]) -> (obs, reward, terminated, truncated, info)."""

    def __init__(self, num_envs, board_width=15, board_height=15, max_players=2, fog_of_war=True, max_turns=500,
                 seed=0, device=0, board_pool=1024, device_outputs=False):
        """device_outputs=True  observation / mask / reward / flags are torch tensors on the GPU; `step` takes a CUDA int64
                             tensor of actions: no board state crosses PCIe, a step is one kernel launch.  Every tensor a
                             step returns lives in a buffer that the step AFTER NEXT reuses.
        default              numpy arrays in, numpy arrays out - the same kernel, its outputs copied to pinned host
                             buffers (one D2H of the observation per step)."""
        import torch
        if not torch.cuda.is_available():
            from ._lib import GvecError
            raise GvecError(-2, "GeneralsVecEnv needs a GPU: its observations, masks and rewards come from the HIP gym kernels "
                                "(there is no host path)")
        self._t = torch
        self.num_envs = num_envs
        self.board_width, self.board_height = board_width, board_height
        self.board_size = board_width * board_height
        self.max_players = max_players
        self.fog_of_war = fog_of_war
        self.max_turns = max_turns
        self.player_id = 0  # "RL_Agent" joins first (generals_env.py:163-169)
        self.single_observation_shape = (9, board_height, board_width)   # spaces.Box(0, 1, (9,H,W), float32) :111-116
        self.single_action_n = self.board_size * 5                        # spaces.Discrete(board_size*5)      :118-120
        self._seed = seed
        self._episode = 0
        self.engine = VecEngine(num_envs, board_width, board_height, max_players, fog_of_war=fog_of_war, device=device,
                                auto_reset=True)
        self._pool = board_pool
        self.valid_actions_mask = None
        self._obs_flip = 0
        self.device_outputs = bool(device_outputs)
        dev = torch.device("cuda", device)
        self._dev = dev
        self.engine.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        n = self.board_size
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        self._d_obs = [z((num_envs, 9, board_height, board_width), torch.float32) for _ in range(2)]
        self._d_mask = [z((num_envs, n * 5), torch.uint8) for _ in range(2)]
        self._d_reward, self._d_done, self._d_winner = z(num_envs, torch.float64), z(num_envs, torch.uint8), z(num_envs, torch.int8)
        self._d_turn = z(num_envs, torch.int64)
        # per-step outputs rotate through three buffer sets: what step k returns is overwritten by step k + 2
        # (needs_reset: written by step k, read by step k + 1 as `resetting` and handed out as info["reset"])
        self._d_step = [{"reward": z(num_envs, torch.float64), "winner": z(num_envs, torch.int8), "turn": z(num_envs, torch.int64),
                         "terminated": z(num_envs, torch.bool), "truncated": z(num_envs, torch.bool), "needs_reset": z(num_envs, torch.bool),
                         "played": z(num_envs, torch.bool), "invalid": z(num_envs, torch.bool), "error": z(num_envs, torch.bool)}
                        for _ in range(3)]
        self._step_no = 0
        self._arg_cache = {}
        self._acts, self.last_actions = None, None
        if not self.device_outputs:   # pinned landing buffers for the default (numpy) mode
            pin = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)
            self._h_obs = [pin((num_envs, 9, board_height, board_width), torch.float32) for _ in range(2)]
            self._h_mask = [pin((num_envs, n * 5), torch.bool) for _ in range(2)]

    # ---- the device path ---------------------------------------------------------------------------------
    def _gym_observe(self):
        self._obs_flip ^= 1
        obs, mask = self._d_obs[self._obs_flip], self._d_mask[self._obs_flip]
        e = self.engine
        check(e.L.gvec_gym_observe(e.h, self.player_id, self._d_turn.data_ptr(), self.max_turns, obs.data_ptr(), mask.data_ptr(),
                                   self._d_reward.data_ptr(), self._d_done.data_ptr(), self._d_winner.data_ptr()), "gvec_gym_observe")
        self.valid_actions_mask = mask.view(self._t.bool)
        return obs

    def _reset_device(self):
        self._d_turn.zero_()
        for b in self._d_step:
            b["needs_reset"].zero_()
        obs = self._gym_observe()          # also stores the stats the first step's reward is measured against
        return obs, {"player_id": self.player_id, "valid_actions_mask": self.valid_actions_mask, "turn": self._d_turn.clone()}

    def _step_args(self, k, flip):
        """The pointer arguments of gvec_gym_step for step number k (mod 3) writing observation buffer `flip`: computed once
        per combination - at 4,096 envs the launch itself takes ~20 us and seventeen data_ptr() calls would add half of that."""
        key = (k % 3, flip)
        a = self._arg_cache.get(key)
        if a is None:
            cur, out = self._d_step[k % 3], self._d_step[(k + 1) % 3]
            obs, mask = self._d_obs[flip], self._d_mask[flip]
            ptrs = (cur["needs_reset"].data_ptr(), self._d_turn.data_ptr(), self.max_turns, obs.data_ptr(), mask.data_ptr(),
                    out["reward"].data_ptr(), out["terminated"].data_ptr(), out["truncated"].data_ptr(), out["winner"].data_ptr(),
                    out["needs_reset"].data_ptr(), out["turn"].data_ptr(), out["played"].data_ptr(), out["invalid"].data_ptr(),
                    out["error"].data_ptr())
            info = {"turn": out["turn"], "valid_actions_mask": mask.view(self._t.bool), "invalid_action": out["invalid"], "error": out["error"],
                    "winner": out["winner"], "reset": cur["needs_reset"]}
            a = self._arg_cache[key] = (ptrs, obs, out, info)
        return a

    def _step_device(self, actions):
        t, e = self._t, self.engine
        if not (isinstance(actions, t.Tensor) and actions.is_cuda and actions.dtype == t.int64 and actions.is_contiguous()
                and actions.numel() == self.num_envs):
            if isinstance(actions, np.ndarray):
                actions = t.from_numpy(np.ascontiguousarray(actions, np.int64))
            actions = t.as_tensor(actions, dtype=t.int64).to(self._dev).reshape(self.num_envs).contiguous()
        self.last_actions = actions        # the tensor the launch reads (kept alive; a collector records it)
        k = self._step_no
        self._step_no += 1
        self._obs_flip ^= 1
        ptrs, obs, out, info = self._step_args(k, self._obs_flip)
        # ONE launch: the learner's action decoded against the resident state's mask (`resetting` = the needs_reset the previous
        # step wrote, zeroed by reset, raised by force_reset), the opponents' moves from the on-device agent, the turn, then
        # observation / mask / reward / flags of the new state (aborted turns are the opponents' business, as over gRPC)
        check(e.L.gvec_gym_step(e.h, self.player_id, self._seed + 1000 * self._episode + 1, actions.data_ptr(), *ptrs), "gvec_gym_step")
        self._episode += 1
        self.valid_actions_mask = info["valid_actions_mask"]
        return obs, out["reward"], out["terminated"], out["truncated"], dict(info)

    def _step_composed(self, actions, others):
        """The same step with the OTHER players' moves supplied by the caller - an opponent policy, self-play - instead of
        drawn by the on-device agent: `others` is [num_envs][max_players] gvec_action (numpy ACTION_DTYPE, or a CUDA uint8
        tensor [num_envs, max_players, 8]); the learner's slot is ignored.  Four launches - gvec_gym_actions (the learner's
        action decoded into its slot, a refused one makes the env sit the call out) -> gvec_step -> gvec_gym_finish_step -
        which together equal gvec_gym_step output for output when `others` are the agent's moves
        (tests/test_vector_env.py::test_gym_step_equals_the_four_call_composition)."""
        t, e, B, P = self._t, self.engine, self.num_envs, self.max_players
        if not (isinstance(actions, t.Tensor) and actions.is_cuda and actions.dtype == t.int64 and actions.is_contiguous() and actions.numel() == B):
            if isinstance(actions, np.ndarray):
                actions = t.from_numpy(np.ascontiguousarray(actions, np.int64))
            actions = t.as_tensor(actions, dtype=t.int64).to(self._dev).reshape(B).contiguous()
        if self._acts is None:
            self._acts = t.zeros((B, P, 8), dtype=t.uint8, device=self._dev)
        if isinstance(others, np.ndarray):
            from .vec_engine import ACTION_DTYPE
            others = t.from_numpy(np.ascontiguousarray(others, ACTION_DTYPE).reshape(B, P).view(np.uint8).reshape(B, P, 8))
        self._acts.copy_(others.reshape(B, P, 8))
        self.last_actions = actions
        k = self._step_no
        self._step_no += 1
        prev_mask = self._d_mask[self._obs_flip]                       # the mask of the observation the learner acted on
        self._obs_flip ^= 1
        _, obs, out, info = self._step_args(k, self._obs_flip)
        cur, mask = self._d_step[k % 3], self._d_mask[self._obs_flip]
        L, pl = e.L, self.player_id
        check(L.gvec_gym_actions(e.h, pl, actions.data_ptr(), prev_mask.data_ptr(), cur["needs_reset"].data_ptr(), self._acts.data_ptr(),
                                 out["played"].data_ptr(), out["invalid"].data_ptr(), out["error"].data_ptr()), "gvec_gym_actions")
        e.step_device(self._acts.data_ptr())
        check(L.gvec_gym_finish_step(e.h, pl, self._d_turn.data_ptr(), self.max_turns, cur["needs_reset"].data_ptr(), out["played"].data_ptr(),
                                     obs.data_ptr(), mask.data_ptr(), out["reward"].data_ptr(), out["terminated"].data_ptr(),
                                     out["truncated"].data_ptr(), out["winner"].data_ptr(), out["needs_reset"].data_ptr(),
                                     out["turn"].data_ptr()), "gvec_gym_finish_step")
        self.valid_actions_mask = info["valid_actions_mask"]
        return obs, out["reward"], out["terminated"], out["truncated"], dict(info)

    def _to_numpy(self, obs, info):
        """The device path's outputs as numpy arrays (default mode): observation and mask land in pinned buffers that
        alternate, so the arrays returned by step k stay intact until step k + 2."""
        i = self._obs_flip
        self._h_obs[i].copy_(obs, non_blocking=True)
        self._h_mask[i].copy_(info["valid_actions_mask"], non_blocking=True)
        out = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in info.items() if k != "valid_actions_mask"}
        self._t.cuda.current_stream(self._dev).synchronize()
        out["valid_actions_mask"] = self._h_mask[i].numpy()
        self.valid_actions_mask = out["valid_actions_mask"]
        return self._h_obs[i].numpy(), out

    # ---- gym API ------------------------------------------------------------------------------------
    def reset(self, seed=None):
        if seed is not None:
            self._seed = seed
        self.engine.reset_generated(self._seed * 1000003 + 17)
        self.engine.build_board_pool(self._pool, self._seed * 7919 + 5)
        obs, info = self._reset_device()
        return (obs, info) if self.device_outputs else self._to_numpy(obs, info)

    def step(self, actions, other_actions=None):
        """other_actions: None = the other players are the on-device random agent (ONE launch); else their moves for this
        step ([num_envs][max_players] gvec_action, see _step_composed)."""
        run = self._step_device if other_actions is None else (lambda a: self._step_composed(a, other_actions))
        if self.device_outputs:
            return run(actions)
        obs, reward, terminated, truncated, info = run(np.asarray(actions, np.int64))
        obs, info = self._to_numpy(obs, info)
        return obs, reward.cpu().numpy(), terminated.cpu().numpy(), truncated.cpu().numpy(), info

    def force_reset(self, env_mask):
        """Ends the running episode of the marked envs: they are re-dealt in the NEXT step (GVEC_ACT_RESET_ENV semantics),
        exactly as if that step had been preceded by terminated / truncated.  How a collector cuts an episode at its own
        length limit (ParallelEnvPool.max_steps_per_episode, vector_env.py:177) without a per-env reset() call."""
        t = self._t
        if isinstance(env_mask, t.Tensor):
            m = env_mask.to(device=self._dev, dtype=t.bool)        # a CUDA mask stays on the device: no synchronisation
        else:
            m = t.as_tensor(np.asarray(env_mask, bool)).to(self._dev)
        self.needs_reset_buffer().logical_or_(m)

    def needs_reset_buffer(self):
        """The bool[num_envs] CUDA tensor the NEXT step reads as `resetting` (written by the last step: terminated |
        truncated).  A device-side collector raises entries of it to cut episodes (gvec_pool_collect)."""
        return self._d_step[self._step_no % 3]["needs_reset"]

    def close(self):
        self.engine.close()


# --------------------------------------------------------------------------------------------------------------------
# The single-env shape: what code written against the reference's `GeneralsEnv` (and its ParallelEnvPool, whose
# env_factory makes one env per worker) instantiates.  gymnasium is not a dependency of this package: the two spaces are
# minimal stand-ins with the attributes that code reads (.shape / .dtype / .low / .high, .n / .sample() / .contains()).
# --------------------------------------------------------------------------------------------------------------------
class _Box:
    def __init__(self, low, high, shape, dtype):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())


class _Discrete:
    def __init__(self, n, seed=None):
        self.n = int(n)
        self._rng = np.random.default_rng(seed)

    def sample(self, mask=None):
        if mask is not None and np.any(mask):
            return int(self._rng.choice(np.flatnonzero(mask)))
        return int(self._rng.integers(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n


class GeneralsEnv:
    """python/generals_gym/generals_env.py:GeneralsEnv - same constructor keywords, `observation_space` (Box(0, 1,
    (9, H, W), float32), :111-116), `action_space` (Discrete(board_size * 5), :118-120), `reset(seed, options) -> (obs,
    info)` (:142-208: info keys game_id / player_id / valid_actions_mask / turn), `step(action) -> (obs, reward, terminated,
    truncated, info)` (:210-289: an action the mask rejects returns (obs, -0.1, False, False, {"invalid_action": True}) and
    the game does not advance; otherwise info carries turn / valid_actions_mask / game_status / winner), `render`, `close` -
    served by a ONE-board GeneralsVecEnv instead of a gRPC server (`server_address` is kept as an attribute only).  One board
    per launch wastes the GPU: use GeneralsVecEnv / ParallelVecEnvPool for throughput; this class is for code that wants the
    reference's object.  Opponent: the on-device random agent (the reference's default is a random opponent too, :443-497),
    or `opponent_agent` - any object with the reference's `select_action(game_state_proto) -> Action proto | None`
    (:244-255): it is handed the learner's proto GameState (wire.game_state: the reference passes `self.current_state`), its
    move is played for player 1 (GeneralsVecEnv.step(..., other_actions=...); the remaining seats of a game with more than
    two players then do not move).  `self_play` is an attribute the reference stores and never reads; so here."""
    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 4}

    def __init__(self, server_address="localhost:50051", board_width=15, board_height=15, max_players=2, fog_of_war=True,
                 render_mode=None, self_play=False, opponent_agent=None, max_turns=500, turn_time_ms=500,
                 collect_experiences=False, device=0, seed=0):
        self.server_address, self.board_width, self.board_height = server_address, board_width, board_height
        self.board_size = board_width * board_height
        self.max_players, self.fog_of_war, self.render_mode = max_players, fog_of_war, render_mode
        self.self_play, self.opponent_agent, self.max_turns = self_play, opponent_agent, max_turns
        self.turn_time_ms, self.collect_experiences, self.device = turn_time_ms, collect_experiences, device
        self._vec = GeneralsVecEnv(1, board_width, board_height, max_players, fog_of_war=fog_of_war, max_turns=max_turns, seed=seed,
                                   device=device, board_pool=4)
        self._seed, self._games = seed, 0
        self.game_id = None
        self.player_id = None
        self.turn_count = 0
        self.observation_space = _Box(0.0, 1.0, (9, board_height, board_width), np.float32)
        self.action_space = _Discrete(self.board_size * 5, seed)
        self.valid_actions_mask = None
        self._obs = None

    def reset(self, seed=None, options=None):
        if seed is not None:
            self._seed = seed
        self._games += 1
        obs, info = self._vec.reset(seed=self._seed * 7919 + self._games)      # a new game (CreateGame + JoinGame x2, :158-186)
        self.game_id, self.player_id, self.turn_count = f"vec-game-{self._games}", 0, 0
        self._obs = obs[0].copy()
        self.valid_actions_mask = info["valid_actions_mask"][0].copy()
        return self._obs, {"game_id": self.game_id, "player_id": self.player_id, "valid_actions_mask": self.valid_actions_mask,
                           "turn": self.turn_count}

    def _proto_state(self):
        """`self.current_state` of the reference (:256-262): the proto GameState the server sends for the learner's token."""
        from . import wire
        e = self._vec.engine
        st = e.game_state(fields=wire.STATE_FIELDS)
        vis, fog = e.compute_player_visibility(self.player_id)
        return wire.game_state(st, vis, fog, e.get_legal_action_mask(0, self.player_id), 0, self.player_id, game_id=self.game_id or "",
                               names=["RL_Agent", "Opponent"] + [f"player{p}" for p in range(2, self.max_players)])

    def _opponent_moves(self):
        """generals_env.py:244-255: `opponent_agent.select_action(self.current_state)` - the reference hands the opponent the
        LEARNER's state - returns a proto Action (or None = no move), submitted with the opponent's token (player 1)."""
        from .vec_engine import ACTION_DTYPE
        others = np.zeros((1, self.max_players), ACTION_DTYPE)
        act = self.opponent_agent.select_action(self._proto_state())
        if act:
            src = getattr(act, "from")
            others[0, 1] = (src.x, src.y, act.to.x, act.to.y, 1 | (2 if getattr(act, "half", False) else 0), (0, 0, 0))   # GVEC_ACT_VALID | _HALF
        return others

    def step(self, action):
        others = self._opponent_moves() if self.opponent_agent is not None else None
        obs, reward, terminated, truncated, info = self._vec.step(np.array([int(action)], np.int64), other_actions=others)
        self._obs = obs[0].copy()
        if info["invalid_action"][0]:                                           # :226-231
            return self._obs, -0.1, False, False, {"invalid_action": True}
        if info["error"][0]:                                                    # :240-243: the server refused the move
            return self._obs, -0.1, False, False, {"error": "move rejected by the engine"}
        self.turn_count = int(info["turn"][0])
        self.valid_actions_mask = info["valid_actions_mask"][0].copy()
        term, trunc = bool(terminated[0]), bool(truncated[0])
        return self._obs, float(reward[0]), term, trunc, {
            "turn": self.turn_count, "valid_actions_mask": self.valid_actions_mask,
            "game_status": "GAME_STATUS_FINISHED" if term else "GAME_STATUS_IN_PROGRESS",      # common.proto:60-64
            "winner": int(info["winner"][0]) if term else None}

    def render(self):
        """:563-597: the learner's view as text ("human" mode only), from the proto state like the reference's."""
        if self.render_mode != "human" or self._obs is None:
            return
        tiles = self._proto_state().board.tiles
        bar = "=" * (self.board_width * 4 + 1)
        print(f"\nTurn {self.turn_count}")
        print(bar)
        for y in range(self.board_height):
            cells = []
            for x in range(self.board_width):
                tile = tiles[y * self.board_width + x]
                if not tile.visible:
                    cells.append(" ? ")
                elif tile.type == 4:                                   # TILE_TYPE_MOUNTAIN (common.proto)
                    cells.append("###")
                elif tile.owner_id == self.player_id:
                    cells.append(f" {tile.army_count:2}")
                elif tile.owner_id >= 0:
                    cells.append(f"-{tile.army_count:2}")
                else:
                    cells.append(" . ")
            print("|" + "|".join(cells) + "|")
        print(bar)

    def close(self):
        self._vec.close()
