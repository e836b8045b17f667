"""GeneralsVecEnv — drop-in *vector* form of the reference's gym environment.

Reference: python/generals_gym/generals_env.py (GeneralsEnv) and vector_env.py (ParallelEnvPool,
which runs N GeneralsEnv instances from N threads, each doing two gRPC round trips and a 50 ms
sleep per step).  Here the same per-env interface is served for B boards by one HIP launch per
step through VecEngine:

  * observation (9, H, W) float32            generals_env.py:111-116, 291-342
  * action space Discrete(board_size * 5)    generals_env.py:118-120, 344-387 (idx*5 + {up,right,down,left,half})
  * action decoding incl. the half-move quirk generals_env.py:389-441
  * reward                                    generals_env.py:499-561
  * terminated / truncated / info keys        generals_env.py:272-289

What the learner sees is the *proto* view the gRPC server would send for its player token:
fog rules of internal/grpc/gameserver/server.go:556-582 (hidden tile -> type NORMAL, owner -1,
army 0; fogged tile -> type kept, owner / army hidden) and the PlayerState fields of
server.go:526-553 (army_count = Player.ArmyCount, tile_count = len(OwnedTiles), status by Alive).

Deliberate differences (documented in DESIGN.md): opponents are the on-device random agent
(uniform over legal moves, 30 % half moves, 10 % no-op) instead of Python's `random.choice` over
full moves; finished / truncated envs are re-dealt on their next step ("next-step" autoreset)
because a vector env cannot wait for a per-env reset() call.
"""
import numpy as np

from .vec_engine import ACT_HALF, ACT_SKIP_ENV, ACT_VALID, ACTION_DTYPE, VecEngine
from ._lib import check

ACT_RESET_ENV = 8  # include/generals_vec.h GVEC_ACT_RESET_ENV

# proto/common/v1/common.proto TileType as used by generals_env.py:322-329
_DIRS = ((0, -1), (1, 0), (0, 1), (-1, 0))  # up, right, down, left (generals_env.py:369, 413)


# --------------------------------------------------------------------------------------------------
# pure functions (numpy, batched over the leading axis) mirroring GeneralsEnv's private helpers
# --------------------------------------------------------------------------------------------------
def proto_view(owner, army, type_, visible, fog):
    """convertGameStateToProto's tile rules (server.go:556-582) for one player's token.
    Inputs [B, N]; visible / fog = Engine.ComputePlayerVisibility(player).  Returns the arrays the
    gym env reads from `state.board.tiles`: type (core numbering), owner_id, army_count, visible."""
    visible = visible.astype(bool)
    fogged = fog.astype(bool) & ~visible
    hidden = ~visible & ~fogged
    t = np.where(hidden, 0, type_).astype(np.int32)          # completely hidden: TILE_TYPE_NORMAL
    o = np.where(visible, owner, -1).astype(np.int32)         # hidden and fogged: owner -1
    a = np.where(visible, army, 0).astype(np.int64)           # hidden and fogged: army 0
    return {"type": t, "owner": o, "army": a, "visible": visible}


def build_observation(view, player_id, turn_count, max_turns, width, height, out=None):
    """GeneralsEnv._get_observation (generals_env.py:291-342).  view arrays [B, N] -> [B, 9, H, W].
    `out`: a float32 buffer of that shape to fill (a fresh 9*N*B-float array costs more in page faults
    than every channel below put together)."""
    B = view["owner"].shape[0]
    n = width * height
    if out is None:
        obs = np.zeros((B, 9, n), np.float32)
    else:
        obs = out.reshape(B, 9, n)
        obs[:, 8] = 0.0
    own, army, typ = view["owner"][:, :n], view["army"][:, :n], view["type"][:, :n]
    obs[:, 0] = view["visible"][:, :n]                                               # :312-314
    obs[:, 1] = np.where(own == player_id, 0.5, np.where(own >= 0, 1.0, 0.0))        # :316-322
    obs[:, 2] = np.where(army > 0, np.log(army + 1) / 10.0, 0.0)                     # :324-326 (float64 math, cast on store)
    obs[:, 3] = typ == 0                                                             # normal   :328-336
    obs[:, 4] = typ == 3                                                             # mountain
    obs[:, 5] = typ == 2                                                             # city
    obs[:, 6] = typ == 1                                                             # general
    tc = np.minimum(np.asarray(turn_count, np.float64) / max_turns, 1.0)             # :338-339
    obs[:, 7] = np.broadcast_to(np.asarray(tc, np.float64).reshape(-1, 1), (B, n))
    # channel 8 is left zero by the reference (:341-343)
    return obs.reshape(B, 9, height, width)


def valid_actions_mask(view, player_id, width, height):
    """GeneralsEnv._get_valid_actions_mask (generals_env.py:344-387) -> bool [B, board_size * 5]."""
    B = view["owner"].shape[0]
    n = width * height
    own = (view["owner"][:, :n] == player_id) & (view["army"][:, :n] > 1)            # :362-364
    not_mtn = (view["type"][:, :n] != 3).reshape(B, height, width)
    own2 = own.reshape(B, height, width)
    mask = np.zeros((B, height, width, 5), bool)
    for d, (dx, dy) in enumerate(_DIRS):                                             # :367-383
        tgt = np.zeros((B, height, width), bool)
        ys = slice(max(0, -dy), height - max(0, dy))
        xs = slice(max(0, -dx), width - max(0, dx))
        yt = slice(max(0, dy), height - max(0, -dy))
        xt = slice(max(0, dx), width - max(0, -dx))
        tgt[:, ys, xs] = not_mtn[:, yt, xt]
        mask[..., d] = own2 & tgt
    mask[..., 4] = mask[..., :4].any(-1)                                             # half move valid iff a full move is
    return mask.reshape(B, n * 5)


def decode_actions(actions, width, height):
    """GeneralsEnv._action_index_to_game_action (generals_env.py:389-441) after the mask check.
    Returns from_x, from_y, to_x, to_y, half, dir.  Half moves (move_type 4) take the first direction
    of (up, right, down, left) whose target is inside the board -- mountains are NOT checked there
    (the reference's own 'simplified' rule, :419-425)."""
    actions = np.asarray(actions, np.int64)
    from_idx, info = actions // 5, actions % 5
    fx, fy = from_idx % width, from_idx // width
    half = info == 4
    d = np.where(half, 0, info)
    if half.any():
        first = np.full(actions.shape, 3, np.int64)
        for k in (3, 2, 1, 0):
            dx, dy = _DIRS[k]
            inb = (fx + dx >= 0) & (fx + dx < width) & (fy + dy >= 0) & (fy + dy < height)
            first = np.where(inb, k, first)
        d = np.where(half, first, d)
    dxs = np.array([v[0] for v in _DIRS])[d]
    dys = np.array([v[1] for v in _DIRS])[d]
    return fx, fy, fx + dxs, fy + dys, half, d


def calculate_reward(prev, cur, player_id):
    """GeneralsEnv._calculate_reward (generals_env.py:499-561).  prev / cur: dicts with done [B],
    winner [B], alive [B,P], army_count [B,P], tile_count [B,P].  Returns float64 [B]."""
    B, P = cur["alive"].shape
    ended = cur["done"].astype(bool)                                                 # status != IN_PROGRESS
    r = (cur["tile_count"][:, player_id].astype(np.float64) - prev["tile_count"][:, player_id]) * 1.0   # :540-542
    r = r + (cur["army_count"][:, player_id].astype(np.float64) - prev["army_count"][:, player_id]) * 0.01  # :544-546
    for q in range(P):                                                               # :548-555
        if q != player_id:
            r = r + 50.0 * (prev["alive"][:, q].astype(bool) & ~cur["alive"][:, q].astype(bool))
    win = cur["winner"] == player_id
    return np.where(ended, np.where(win, 100.0, -100.0), r)                          # :520-524


# --------------------------------------------------------------------------------------------------
class GeneralsVecEnv:
    """B GeneralsEnv instances behind the (gymnasium-style) vector API:
    reset() -> (obs, info);  step(actions[B]) -> (obs, reward, terminated, truncated, info)."""

    def __init__(self, num_envs, board_width=15, board_height=15, max_players=2, fog_of_war=True, max_turns=500,
                 seed=0, device=0, board_pool=1024, device_outputs=False, numpy_reference=False):
        """Three modes, identical outputs (tests/test_vector_env.py):
        device_outputs=True  observation / mask / reward / flags are torch tensors on the GPU, produced by the gym kernels
                             (gvec_gym_actions / gvec_gym_finish_step); `step` takes a CUDA int64 tensor of actions: no
                             board state crosses PCIe, and a step is four kernel launches with no tensor glue between
                             them.  Every tensor a step returns lives in a buffer that the step AFTER NEXT reuses.
        default              numpy arrays in, numpy arrays out - the same kernels, their outputs copied to pinned host
                             buffers (one D2H of the observation per step instead of a state read-back + numpy rebuild).
        numpy_reference=True numpy arrays built on the host from a state read-back with the pure functions above: the
                             readable restatement the other two modes are tested against (also what runs when torch
                             has no GPU, e.g. on the oracle-backed engine of the CPU tests)."""
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
        self.turn_count = np.zeros(num_envs, np.int64)
        self._needs_reset = np.zeros(num_envs, bool)
        self._stats = None
        self.valid_actions_mask = None
        self._obs_bufs = [np.zeros((num_envs, 9, board_height, board_width), np.float32) for _ in range(2)]
        self._obs_flip = 0
        self.device_outputs = bool(device_outputs)
        self._via_kernels = False
        if not self.device_outputs and not numpy_reference and hasattr(self.engine, "L"):
            try:
                import torch
                self._via_kernels = torch.cuda.is_available()
            except ImportError:
                self._via_kernels = False
        if self.device_outputs or self._via_kernels:
            import torch
            self._t = torch
            dev = torch.device("cuda", device)
            self._dev = dev
            self.engine.set_stream(torch.cuda.current_stream(dev).cuda_stream)
            n = self.board_size
            z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
            self._d_obs = [z((num_envs, 9, board_height, board_width), torch.float32) for _ in range(2)]
            self._d_mask = [z((num_envs, n * 5), torch.uint8) for _ in range(2)]
            self._d_reward, self._d_done, self._d_winner = z(num_envs, torch.float64), z(num_envs, torch.uint8), z(num_envs, torch.int8)
            self._d_turn = z(num_envs, torch.int64)
            self._d_acts = z((num_envs, max_players, 8), torch.uint8)
            # per-step outputs rotate through three buffer sets: what step k returns is overwritten by step k + 2
            # (needs_reset: written by step k, read by step k + 1 as `resetting` and handed out as info["reset"])
            self._d_step = [{"reward": z(num_envs, torch.float64), "winner": z(num_envs, torch.int8), "turn": z(num_envs, torch.int64),
                             "terminated": z(num_envs, torch.bool), "truncated": z(num_envs, torch.bool), "needs_reset": z(num_envs, torch.bool),
                             "played": z(num_envs, torch.bool), "invalid": z(num_envs, torch.bool), "error": z(num_envs, torch.bool)}
                            for _ in range(3)]
            self._step_no = 0
            if self._via_kernels:   # pinned landing buffers for the default (numpy) mode
                pin = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)
                self._h_obs = [pin((num_envs, 9, board_height, board_width), torch.float32) for _ in range(2)]
                self._h_mask = [pin((num_envs, n * 5), torch.bool) for _ in range(2)]

    # ---- helpers ------------------------------------------------------------------------------------
    def _read(self):
        st = self.engine.game_state(fields=("owner", "army", "type", "done", "winner", "alive", "army_count", "tile_count"))
        vis, fog = self.engine.compute_player_visibility(self.player_id)
        view = proto_view(st["owner"], st["army"], st["type"], vis, fog)
        stats = {k: st[k] for k in ("done", "winner", "alive", "army_count", "tile_count")}
        return view, stats

    def _observe(self, view):
        # two observation buffers alternate: the array returned by step k stays intact until step k + 2
        self._obs_flip ^= 1
        obs = build_observation(view, self.player_id, self.turn_count, self.max_turns, self.board_width, self.board_height,
                                out=self._obs_bufs[self._obs_flip])
        self.valid_actions_mask = valid_actions_mask(view, self.player_id, self.board_width, self.board_height)
        return obs

    # ---- gym API ------------------------------------------------------------------------------------
    # ---- device mode ------------------------------------------------------------------------------
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

    def _step_device(self, actions):
        t, e = self._t, self.engine
        if isinstance(actions, np.ndarray):
            actions = t.from_numpy(np.ascontiguousarray(actions, np.int64))
        actions = t.as_tensor(actions, dtype=t.int64).to(self._dev).reshape(self.num_envs).contiguous()
        k = self._step_no
        self._step_no += 1
        cur, out = self._d_step[k % 3], self._d_step[(k + 1) % 3]
        resetting = cur["needs_reset"]          # written by the previous step (or zeroed by reset)
        # opponents: the on-device random agent writes every slot; the learner's slot is then overwritten
        check(e.L.gvec_agent_actions(e.h, self._seed + 1000 * self._episode + 1, 0, self._d_acts.data_ptr(), 1), "gvec_agent_actions")
        self._episode += 1
        prev_mask = self._d_mask[self._obs_flip]
        check(e.L.gvec_gym_actions(e.h, self.player_id, actions.data_ptr(), prev_mask.data_ptr(), resetting.data_ptr(), self._d_acts.data_ptr(),
                                   out["played"].data_ptr(), out["invalid"].data_ptr(), out["error"].data_ptr()), "gvec_gym_actions")
        e.step_device(self._d_acts.data_ptr())     # aborted turns are the opponents' business, as over gRPC
        # observation, mask, reward and the step's bookkeeping (turn count, terminated / truncated / needs_reset) in one launch
        self._obs_flip ^= 1
        obs, mask = self._d_obs[self._obs_flip], self._d_mask[self._obs_flip]
        check(e.L.gvec_gym_finish_step(e.h, self.player_id, self._d_turn.data_ptr(), self.max_turns, resetting.data_ptr(),
                                       out["played"].data_ptr(), obs.data_ptr(), mask.data_ptr(), out["reward"].data_ptr(),
                                       out["terminated"].data_ptr(), out["truncated"].data_ptr(), out["winner"].data_ptr(),
                                       out["needs_reset"].data_ptr(), out["turn"].data_ptr()), "gvec_gym_finish_step")
        self.valid_actions_mask = mask.view(t.bool)
        info = {"turn": out["turn"], "valid_actions_mask": self.valid_actions_mask, "invalid_action": out["invalid"], "error": out["error"],
                "winner": out["winner"], "reset": resetting}
        return obs, out["reward"], out["terminated"], out["truncated"], info

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

    def reset(self, seed=None):
        if seed is not None:
            self._seed = seed
        self.engine.reset_generated(self._seed * 1000003 + 17)
        self.engine.build_board_pool(self._pool, self._seed * 7919 + 5)
        if self.device_outputs:
            return self._reset_device()
        if self._via_kernels:
            obs, info = self._reset_device()
            return self._to_numpy(obs, info)
        self.turn_count[:] = 0
        self._needs_reset[:] = False
        view, self._stats = self._read()
        obs = self._observe(view)
        info = {"player_id": self.player_id, "valid_actions_mask": self.valid_actions_mask, "turn": self.turn_count.copy()}
        return obs, info

    def step(self, actions):
        if self.device_outputs:
            return self._step_device(actions)
        if self._via_kernels:
            obs, reward, terminated, truncated, info = self._step_device(np.asarray(actions, np.int64))
            obs, info = self._to_numpy(obs, info)
            return obs, reward.cpu().numpy(), terminated.cpu().numpy(), truncated.cpu().numpy(), info
        B, W, H = self.num_envs, self.board_width, self.board_height
        actions = np.asarray(actions, np.int64).reshape(B)
        resetting = self._needs_reset.copy()
        in_range = (actions >= 0) & (actions < self.single_action_n)
        valid = in_range & self.valid_actions_mask[np.arange(B), np.clip(actions, 0, self.single_action_n - 1)]
        fx, fy, tx, ty, half, d = decode_actions(np.where(in_range, actions, 0), W, H)
        # server-side Validate at submit time (action_validator.go:114-139): the half-move direction may hit a mountain
        from_idx = fy * W + fx
        accepted = valid & self.valid_actions_mask[np.arange(B), from_idx * 5 + d]
        played = accepted | resetting
        acts = self.engine.agent_actions(self._seed + 1000 * self._episode + 1)   # opponents (and a draft for player 0)
        self._episode += 1
        a0 = acts[:, self.player_id]
        a0["from_x"], a0["from_y"], a0["to_x"], a0["to_y"] = fx, fy, tx, ty
        a0["flags"] = np.where(played, ACT_VALID | np.where(half, ACT_HALF, 0), 0).astype(np.uint8)
        acts[:, self.player_id] = a0
        first = acts[:, 0]
        first["flags"] = np.where(played, first["flags"] & ~np.uint8(ACT_SKIP_ENV), first["flags"] | np.uint8(ACT_SKIP_ENV))
        # finished / truncated envs are re-dealt in this step (GVEC_ACT_RESET_ENV): no read-back / poke of `done`
        first["flags"] = np.where(resetting, first["flags"] | np.uint8(ACT_RESET_ENV), first["flags"])
        acts[:, 0] = first
        self.engine.step(acts)  # per-env move errors (aborted turns) are the opponents' business, as over gRPC
        prev = self._stats
        self.turn_count = np.where(resetting, 0, self.turn_count + played)
        view, self._stats = self._read()
        reward = calculate_reward(prev, self._stats, self.player_id)
        reward = np.where(resetting, 0.0, np.where(played, reward, -0.1))            # :226-241: invalid action / failed submit
        terminated = self._stats["done"].astype(bool) & played & ~resetting
        truncated = (self.turn_count >= self.max_turns) & played & ~resetting
        obs = self._observe(view)
        self._needs_reset = terminated | truncated
        info = {"turn": self.turn_count.copy(), "valid_actions_mask": self.valid_actions_mask,
                "invalid_action": ~valid & ~resetting, "error": valid & ~accepted & ~resetting,
                "winner": np.where(terminated, self._stats["winner"], -1), "reset": resetting}
        return obs, reward, terminated, truncated, info

    def close(self):
        self.engine.close()
