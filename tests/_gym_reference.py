"""TEST INFRASTRUCTURE - the readable numpy restatement of python/generals_gym/generals_env.py's private helpers and of
GeneralsEnv.step's bookkeeping, batched over the leading axis.  It is the checker the gym kernels
(gvec_gym_observe / gvec_gym_actions / gvec_gym_finish_step / gvec_gym_step) are compared with, and is itself pinned to a
scalar, line-by-line restatement in tests/test_vector_env.py.  It lived inside the package up to round 2
(vector_env.py `numpy_reference=True`); the product has ONE execution path now - the HIP kernels - and nothing under
generalsreinforcementlearning_amd/ imports this file.

Parity unpinned: the reference's GeneralsEnv cannot be imported here (gymnasium is not installed) and needs a live Go
server; nothing the reference holds backs this restatement beyond its source text (generals_env.py:111-120, 226-259,
291-441, 499-561; internal/grpc/gameserver/server.go:526-582)."""
import numpy as np

ACT_VALID, ACT_HALF, ACT_SKIP_ENV, ACT_RESET_ENV = 1, 2, 4, 8   # include/generals_vec.h
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



class NumpyReferenceVecEnv:
    """GeneralsVecEnv's contract (reset / step, same outputs) computed on the host from state read-backs with the pure
    functions above.  `engine`: anything with VecEngine's reset_generated / build_board_pool / game_state /
    compute_player_visibility / agent_actions / step (the HIP engine under -m gpu, the oracle-backed stand-in on CPU)."""

    def __init__(self, engine, num_envs, board_width, board_height, max_players=2, fog_of_war=True, max_turns=500, seed=0, board_pool=1024):
        self.engine, self.num_envs = engine, num_envs
        self.board_width, self.board_height, self.board_size = board_width, board_height, board_width * board_height
        self.max_players, self.fog_of_war, self.max_turns = max_players, fog_of_war, max_turns
        self.player_id = 0
        self.single_observation_shape = (9, board_height, board_width)
        self.single_action_n = self.board_size * 5
        self._seed, self._episode, self._pool = seed, 0, board_pool
        self.turn_count = np.zeros(num_envs, np.int64)
        self._needs_reset = np.zeros(num_envs, bool)
        self._stats = None
        self.valid_actions_mask = None

    def _read(self):
        st = self.engine.game_state(fields=("owner", "army", "type", "done", "winner", "alive", "army_count", "tile_count"))
        vis, fog = self.engine.compute_player_visibility(self.player_id)
        view = proto_view(st["owner"], st["army"], st["type"], vis, fog)
        return view, {k: st[k] for k in ("done", "winner", "alive", "army_count", "tile_count")}

    def _observe(self, view):
        obs = build_observation(view, self.player_id, self.turn_count, self.max_turns, self.board_width, self.board_height)
        self.valid_actions_mask = valid_actions_mask(view, self.player_id, self.board_width, self.board_height)
        return obs

    def reset(self, seed=None):
        if seed is not None:
            self._seed = seed
        self.engine.reset_generated(self._seed * 1000003 + 17)
        self.engine.build_board_pool(self._pool, self._seed * 7919 + 5)
        self.turn_count[:] = 0
        self._needs_reset[:] = False
        view, self._stats = self._read()
        obs = self._observe(view)
        return obs, {"player_id": self.player_id, "valid_actions_mask": self.valid_actions_mask, "turn": self.turn_count.copy()}

    def force_reset(self, env_mask):
        self._needs_reset |= np.asarray(env_mask, bool)

    def step(self, actions):
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
