"""VecEngine — host mirror of the reference's ``*game.Engine`` method set for B boards.

Reference seam (internal/game/engine.go): NewEngine :62, Step :75, GameState :197,
IsGameOver :198, GetWinner :248, GetLegalActionMask :271, GetChangedTiles :283,
GetVisibilityChangedTiles :292, ComputePlayerVisibility (visibility.go:153).
Every method is the batched form of the Go method it is named after; arrays are
env-major numpy arrays (host) or torch-ROCm tensors passed by device pointer.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Config, GvecError, RolloutStats, StateView, check

TILE_NORMAL, TILE_GENERAL, TILE_CITY, TILE_MOUNTAIN = 0, 1, 2, 3  # core/board.go:20-26
ACT_VALID, ACT_HALF, ACT_SKIP_ENV = 1, 2, 4
ERR_NAMES = {0: None, 1: "ErrInvalidCoordinates", 2: "ErrNotAdjacent", 3: "ErrNotOwned", 4: "ErrInsufficientArmy",
             5: "ErrGameOver", 6: "ErrInvalidPlayer", 7: "ErrMoveToSelf", 8: "ErrTargetIsMountain"}  # core/errors.go:8-17

ACTION_DTYPE = np.dtype([("from_x", "i1"), ("from_y", "i1"), ("to_x", "i1"), ("to_y", "i1"), ("flags", "u1"),
                         ("reserved", "u1", (3,))])
MEM_HOST, MEM_DEVICE = 0, 1

_STATE_SPEC = {"army": (np.int32, "tile"), "owner": (np.int8, "tile"), "type": (np.uint8, "tile"), "visible": (np.uint8, "tile"),
               "listed": (np.int8, "tile"), "changed": (np.uint8, "tile"), "vis_changed": (np.uint8, "tile"),
               "turn": (np.int32, "env"), "done": (np.uint8, "env"), "winner": (np.int8, "env"), "width": (np.int32, "env"),
               "height": (np.int32, "env"), "players": (np.int32, "env"), "alive": (np.uint8, "player"),
               "army_count": (np.int32, "player"), "tile_count": (np.int32, "player"), "general_idx": (np.int32, "player")}


def make_actions(num_envs, max_players, moves=()):
    """moves: iterable of (env, player, from_x, from_y, to_x, to_y, move_all) -> [B][P] action array.
    Mirrors core.MoveAction (core/action.go:23-36); clamps coordinates like the cgo shim does."""
    a = np.zeros((num_envs, max_players), ACTION_DTYPE)
    for env, p, fx, fy, tx, ty, move_all in moves:
        c = lambda v: int(max(-128, min(127, v)))
        a[env, p] = (c(fx), c(fy), c(tx), c(ty), ACT_VALID | (0 if move_all else ACT_HALF), (0, 0, 0))
    return a


def unpack_legal_bits(bits, width, height):
    """[..., mask_bytes] packed mask -> [..., W*H*4] bool in Engine.GetLegalActionMask order
    (index (y*W+x)*4+d, d = 0 up, 1 right, 2 down, 3 left; rules/legal_moves.go:13-18).

    Packed form (include/generals_vec.h): four direction bit-planes of mask_bytes/4 bytes each;
    bit t (LSB first) of plane d is action (t, d), t = y*W + x."""
    bits = np.ascontiguousarray(bits)
    plane = bits.shape[-1] // 4
    u = np.unpackbits(bits.reshape(bits.shape[:-1] + (4, plane)), axis=-1, bitorder="little")  # [..., d, t]
    u = u[..., : width * height]
    return np.moveaxis(u, -2, -1).reshape(bits.shape[:-1] + (width * height * 4,)).astype(bool)


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a.data_ptr()  # torch tensor


class VecEngine:
    """B independent engines stepped by one HIP launch (one wavefront per board)."""

    def __init__(self, num_envs, width, height, players, fog_of_war=True, device=0, production=(1, 1, 1),
                 normal_growth_interval=25, auto_reset=False, stream=None, lib=None, devices=None):
        """devices: a list of HIP device ordinals -> ONE engine over several GPUs (gvec_create_sharded): the boards are split
        into contiguous shards, one per listed device; every host-array method works unchanged on all B boards, device-pointer
        methods belong to one device (`shard(i)`).  The batch plays the same games whatever the number of shards."""
        self.L = lib if lib is not None else _lib.load()
        cfg = Config()
        check(self.L.gvec_config_default(C.byref(cfg)))
        cfg.num_envs, cfg.max_width, cfg.max_height, cfg.max_players = num_envs, width, height, players
        cfg.device, cfg.fog_of_war, cfg.auto_reset = device, int(bool(fog_of_war)), int(bool(auto_reset))
        cfg.prod_general, cfg.prod_city, cfg.prod_normal = production
        cfg.normal_growth_interval = normal_growth_interval
        self.h = C.c_void_p()
        self._owned = True
        if devices is not None:
            devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            check(self.L.gvec_create_sharded(C.byref(cfg), devs, len(devices), C.byref(self.h)), "gvec_create_sharded")
        else:
            check(self.L.gvec_create(C.byref(cfg), C.byref(self.h)), "gvec_create")
        self.B, self.max_w, self.max_h, self.max_p = num_envs, width, height, players
        self.device = int(device)          # the device of a plain handle (a sharded one: see `devices`)
        self.stride = self.L.gvec_tile_stride(self.h)
        self.mask_bytes = self.L.gvec_mask_bytes(self.h)
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            if getattr(self, "_owned", True):      # a shard view belongs to its sharded engine
                self.L.gvec_destroy(self.h)
            self.h = C.c_void_p()
        for p in getattr(self, "_pinned", []):
            self.L.gvec_host_free(p)
        self._pinned = []

    # ---- sharded engines (gvec_create_sharded) ---------------------------------------------------------
    def num_shards(self):
        return self.L.gvec_num_shards(self.h)

    def shard(self, i):
        """(view, env_begin, num_envs, device): a VecEngine view of shard i for the device-pointer methods; it must not
        outlive this engine."""
        child, begin, n, dev = C.c_void_p(), C.c_int32(), C.c_int32(), C.c_int32()
        check(self.L.gvec_shard(self.h, i, C.byref(child), C.byref(begin), C.byref(n), C.byref(dev)), "gvec_shard")
        v = object.__new__(VecEngine)
        v.L, v.h, v._owned = self.L, child, False
        v.B, v.max_w, v.max_h, v.max_p, v.stride, v.mask_bytes = n.value, self.max_w, self.max_h, self.max_p, self.stride, self.mask_bytes
        return v, begin.value, n.value, dev.value

    def gather_experience_records(self, n, shard_env_begin=0, env_id_base=0, dst_device_ptr=None, dst_device=0):
        """Sharded engines: every shard's records of ITS envs [shard_env_begin, +n) to one place.  dst_device_ptr None ->
        a host uint8 array [num_shards * n * record_bytes] is returned (each device copies over its own PCIe link);
        else the slabs travel GPU-to-GPU into that buffer on dst_device."""
        k = self.num_shards()
        if dst_device_ptr is None:
            out = np.empty(k * n * self.experience_record_bytes(), np.uint8)
            check(self.L.gvec_gather_experience_records(self.h, shard_env_begin, n, env_id_base, MEM_HOST, 0, out.ctypes.data),
                  "gvec_gather_experience_records")
            return out
        check(self.L.gvec_gather_experience_records(self.h, shard_env_begin, n, env_id_base, MEM_DEVICE, dst_device, C.c_void_p(int(dst_device_ptr))),
              "gvec_gather_experience_records")
        return None

    __del__ = close

    def set_stream(self, hip_stream):
        check(self.L.gvec_set_stream(self.h, C.c_void_p(int(hip_stream))))

    def synchronize(self):
        check(self.L.gvec_synchronize(self.h))

    # ---- NewEngine / EngineInitializer (engine_initializer.go:34-87) -------------------------
    def reset(self, army, owner, type, width=None, height=None, players=None, env_ids=None):
        """Upload boards ([n][max_w*max_h] planes, index y*W+x) and run performInitialSetup."""
        n = len(army)
        full = lambda v, d: np.full(n, d, np.int32) if v is None else np.ascontiguousarray(v, np.int32)
        w, h, p = full(width, self.max_w), full(height, self.max_h), full(players, self.max_p)
        ids = None if env_ids is None else np.ascontiguousarray(env_ids, np.int32)
        army = np.ascontiguousarray(army, np.int32).reshape(n, self.stride)
        owner = np.ascontiguousarray(owner, np.int8).reshape(n, self.stride)
        type = np.ascontiguousarray(type, np.uint8).reshape(n, self.stride)
        check(self.L.gvec_reset(self.h, _ptr(ids), n, _ptr(army), _ptr(owner), _ptr(type), _ptr(w), _ptr(h), _ptr(p), MEM_HOST),
              "gvec_reset")

    def reset_generated(self, seed, width=None, height=None, players=None):
        cv = lambda v: None if v is None else np.ascontiguousarray(v, np.int32)
        w, h, p = cv(width), cv(height), cv(players)
        check(self.L.gvec_reset_generated(self.h, seed, _ptr(w), _ptr(h), _ptr(p)), "gvec_reset_generated")

    def reset_go_seeded(self, seeds, width=None, height=None, players=None):
        """Env i starts from the board the Go engine builds from GameConfig.Rng = rand.New(rand.NewSource(seeds[i]))
        (Go's math/rand restated on the device; include/generals_vec.h gvec_reset_go_seeded)."""
        cv = lambda v: None if v is None else np.ascontiguousarray(v, np.int32)
        w, h, p = cv(width), cv(height), cv(players)
        sd = np.ascontiguousarray(seeds, np.int64)
        assert sd.shape == (self.B,)
        check(self.L.gvec_reset_go_seeded(self.h, _ptr(sd), _ptr(w), _ptr(h), _ptr(p)), "gvec_reset_go_seeded")

    def build_board_pool(self, pool_size, seed, width=None, height=None, players=None):
        cv = lambda v: None if v is None else np.ascontiguousarray(v, np.int32)
        w, h, p = cv(width), cv(height), cv(players)
        check(self.L.gvec_build_board_pool(self.h, pool_size, seed, _ptr(w), _ptr(h), _ptr(p)), "gvec_build_board_pool")

    def pinned(self, shape, dtype):
        """A numpy array over page-locked host memory (gvec_host_alloc): host-array calls copy it at full PCIe rate.
        Freed with the engine (keep no reference past close())."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        p = C.c_void_p()
        check(self.L.gvec_host_alloc(max(n, 1), C.byref(p)), "gvec_host_alloc")
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
        a = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
        a[...] = 0
        return a

    # ---- Engine.Step (engine.go:75) --------------------------------------------------------------
    def step(self, actions, want_mask=False, pinned=False):
        """actions: [B][max_players] ACTION_DTYPE.  Returns err[B] (sentinel codes, 0 = nil error)
        and, if want_mask, the packed post-step legal masks [B][max_players][mask_bytes].
        pinned=True: err / masks land in page-locked arrays owned by the engine and REUSED by the next such call (4x the
        PCIe rate of fresh pageable arrays; pass `actions` allocated with `pinned()` for the same on the way in)."""
        actions = np.ascontiguousarray(actions, ACTION_DTYPE).reshape(self.B, self.max_p)
        if pinned:
            if not hasattr(self, "_p_err"):
                self._p_err = self.pinned((self.B,), np.int32)
                self._p_bits = None
            if want_mask and self._p_bits is None:
                self._p_bits = self.pinned((self.B, self.max_p, self.mask_bytes), np.uint8)
            err, bits = self._p_err, (self._p_bits if want_mask else None)
            check(self.L.gvec_step(self.h, _ptr(actions), _ptr(err), _ptr(bits), MEM_HOST), "gvec_step")
            return (err, bits) if want_mask else err
        err = np.zeros(self.B, np.int32)
        bits = np.zeros((self.B, self.max_p, self.mask_bytes), np.uint8) if want_mask else None
        check(self.L.gvec_step(self.h, _ptr(actions), _ptr(err), _ptr(bits), MEM_HOST), "gvec_step")
        return (err, bits) if want_mask else err

    def step_device(self, actions_ptr, err_ptr=None, legal_ptr=None):
        """Zero-copy form: device pointers (ints / torch tensors); enqueued, not synchronised."""
        p = lambda v: None if v is None else (v if isinstance(v, int) else v.data_ptr())
        check(self.L.gvec_step(self.h, p(actions_ptr), p(err_ptr), p(legal_ptr), MEM_DEVICE), "gvec_step")

    # ---- Engine.GetLegalActionMask (engine.go:271-280) -----------------------------------------
    def legal_action_mask_bits(self):
        bits = np.zeros((self.B, self.max_p, self.mask_bytes), np.uint8)
        check(self.L.gvec_legal_mask(self.h, _ptr(bits), MEM_HOST), "gvec_legal_mask")
        return bits

    def get_legal_action_mask(self, env, player_id):
        """[]bool of size W*H*4 for one engine, exactly Engine.GetLegalActionMask(playerID)."""
        st = self.game_state(env, 1, fields=("width", "height"))
        w, h = int(st["width"][0]), int(st["height"][0])
        if player_id < 0 or player_id >= self.max_p:
            return np.zeros(w * h * 4, bool)  # engine.go:273-276
        return unpack_legal_bits(self.legal_action_mask_bits()[env, player_id], w, h)

    # ---- Engine.GameState / IsGameOver / GetWinner / GetChangedTiles (engine.go:197-298) ----------
    def game_state(self, env_begin=0, n=None, fields=None):
        n = self.B - env_begin if n is None else n
        out, view = {}, StateView()
        for name, (dt, kind) in _STATE_SPEC.items():
            if fields is not None and name not in fields:
                continue
            shape = {"tile": (n, self.stride), "env": (n,), "player": (n, self.max_p)}[kind]
            out[name] = np.zeros(shape, dt)
            setattr(view, name, out[name].ctypes.data)
        check(self.L.gvec_read_state(self.h, env_begin, n, C.byref(view), MEM_HOST), "gvec_read_state")
        return out

    def write_state(self, arrays, env_begin=0):
        """Raw poke of engine state (what the Go tests do with e.gs.*); no init pass."""
        view, keep = StateView(), []
        n = None
        for name, a in arrays.items():
            dt, _ = _STATE_SPEC[name]
            a = np.ascontiguousarray(a, dt)
            keep.append(a)
            n = len(a) if n is None else n
            setattr(view, name, a.ctypes.data)
        check(self.L.gvec_write_state(self.h, env_begin, n, C.byref(view), MEM_HOST), "gvec_write_state")

    def is_game_over(self):
        return self.game_state(fields=("done",))["done"].astype(bool)

    def get_winner(self):
        return self.game_state(fields=("winner",))["winner"].astype(np.int32)

    def get_changed_tiles(self):
        return self.game_state(fields=("changed",))["changed"].astype(bool)

    def get_visibility_changed_tiles(self):
        return self.game_state(fields=("vis_changed",))["vis_changed"].astype(bool)

    # ---- Engine.ComputePlayerVisibility (visibility.go:153) --------------------------------------
    def compute_player_visibility(self, player_id):
        vis = np.zeros((self.B, self.stride), np.uint8)
        fog = np.zeros((self.B, self.stride), np.uint8)
        check(self.L.gvec_player_visibility(self.h, player_id, _ptr(vis), _ptr(fog), MEM_HOST), "gvec_player_visibility")
        return vis.astype(bool), fog.astype(bool)

    # ---- createStreamUpdate's deltas (server.go:632-777) -----------------------------------------
    def stream_deltas(self, player):
        """-> kind[B] (1 delta / 2 full state), count[B], updates[B][cap] uint64 (include/generals_vec.h gvec_stream_deltas):
        what a turn's broadcast needs of the boards, a few bytes per env."""
        cap = self.L.gvec_stream_delta_cap(self.h)
        kind, count = np.zeros(self.B, np.uint8), np.zeros(self.B, np.int32)
        upd = np.zeros((self.B, cap), np.uint64)
        check(self.L.gvec_stream_deltas(self.h, player, _ptr(kind), _ptr(count), _ptr(upd), MEM_HOST), "gvec_stream_deltas")
        return kind, count, upd

    def stream_deltas_packed(self, player, full_tiles=False):
        """-> kind[B], offset[B + 1], updates[total]: env e's updates are updates[offset[e]:offset[e + 1]] - only the updates
        that exist cross PCIe (gvec_stream_deltas_packed).  full_tiles: an env the server would send a full state for
        (kind 2) contributes all its tiles, fog rules applied, so no board is ever read back.  The landing buffers are
        page-locked and reused by the next call."""
        cap = self.stride if full_tiles else self.L.gvec_stream_delta_cap(self.h)
        key = "_sd_full" if full_tiles else "_sd"
        if not hasattr(self, key):
            setattr(self, key, (self.pinned((self.B,), np.uint8), self.pinned((self.B + 1,), np.int64), self.pinned((self.B * cap,), np.uint64)))
        kind, off, upd = getattr(self, key)
        total = C.c_int64()
        check(self.L.gvec_stream_deltas_packed(self.h, player, int(bool(full_tiles)), _ptr(kind), _ptr(off), _ptr(upd), self.B * cap, C.byref(total)),
              "gvec_stream_deltas_packed")
        return kind, off, upd[: total.value]

    # ---- synthetic random-agent rollouts ----------------------------------------------------------
    def set_agent_mix(self, noop_per_65536=6554, half_per_65536=19661):
        """Random-agent mix for rollout / agent_actions: P(no-op) = noop/65536, P(half move) = half/65536.
        (45875, 19661) are the rates of the reference's game.GenerateRandomActions (demo_helpers.go:20,44);
        (0, 0) always plays a full move, uniform over the legal ones."""
        check(self.L.gvec_set_agent_mix(self.h, int(noop_per_65536), int(half_per_65536)), "gvec_set_agent_mix")

    def counters(self):
        """Lifetime counters summed over all envs: turns actually played, aborted turns (H5), games finished."""
        st = RolloutStats()
        check(self.L.gvec_counters(self.h, C.byref(st)), "gvec_counters")
        return {"env_steps": st.env_steps, "aborted_turns": st.aborted_turns, "games_finished": st.games_finished}

    def step_traffic_bytes(self):
        """Bytes one env-step must move by construction of the resident layout: dict(read, write, mask, rare_extra)."""
        out = (C.c_int64 * 4)()
        check(self.L.gvec_step_traffic_bytes(self.h, out), "gvec_step_traffic_bytes")
        return {"read": out[0], "write": out[1], "mask": out[2], "rare_extra": out[3]}

    def agent_actions(self, seed, invalid_permille=0):
        acts = np.zeros((self.B, self.max_p), ACTION_DTYPE)
        check(self.L.gvec_agent_actions(self.h, seed, invalid_permille, _ptr(acts), MEM_HOST), "gvec_agent_actions")
        return acts

    def rollout(self, turns, seed, invalid_permille=0, fused=True, want_stats=True):
        st = RolloutStats()
        check(self.L.gvec_rollout(self.h, turns, seed, invalid_permille, int(bool(fused)), C.byref(st) if want_stats else None),
              "gvec_rollout")
        return {"env_steps": st.env_steps, "aborted_turns": st.aborted_turns, "games_finished": st.games_finished} if want_stats else None

    def rollout_range(self, env_begin, n, turns, seed, invalid_permille=0):
        """Per-turn rollout of envs [env_begin, env_begin + n) only (gvec_rollout_range); enqueued, not synchronised."""
        check(self.L.gvec_rollout_range(self.h, env_begin, n, turns, seed, invalid_permille), "gvec_rollout_range")

    # ---- internal/experience side channel (serializer.go, rewards.go) --------------------------------
    def experience_begin(self):
        """TurnProcessor.captureStateForExperience (turn_processor.go:116-121): snapshot before the step."""
        check(self.L.gvec_experience_begin(self.h), "gvec_experience_begin")

    def experience_begin_range(self, env_begin, n):
        check(self.L.gvec_experience_begin_range(self.h, env_begin, n), "gvec_experience_begin_range")

    def experience_record_layout(self):
        """dict(record_dw, mp, fd, ns, max_players, stride): how to parse gvec_experience_records' output."""
        out = (C.c_int32 * 8)()
        check(self.L.gvec_experience_record_layout(self.h, out), "gvec_experience_record_layout")
        return {"record_dw": out[0], "mp": out[1], "fd": out[2], "ns": out[3], "max_players": out[4], "stride": out[5]}

    def experience_record_bytes(self):
        return check(self.L.gvec_experience_record_bytes(self.h), "gvec_experience_record_bytes")

    def experience_records(self, dst_device_ptr, actions=None, env_begin=0, n=None, env_id_base=0):
        """Writes the compact experience records of envs [env_begin, env_begin+n) to device memory.
        actions: the [B][max_players] array that was stepped (host numpy), or None = the handle's action buffer."""
        n = self.B - env_begin if n is None else n
        a = None if actions is None else np.ascontiguousarray(actions, ACTION_DTYPE).reshape(self.B, self.max_p)
        check(self.L.gvec_experience_records(self.h, _ptr(a), MEM_HOST, env_begin, n, env_id_base, C.c_void_p(int(dst_device_ptr))),
              "gvec_experience_records")

    def record_agent_actions(self, on=True):
        check(self.L.gvec_record_agent_actions(self.h, int(bool(on))), "gvec_record_agent_actions")

    def experience_rewards(self):
        """CalculateReward(prev snapshot, current, player) -> (rewards[B][P] float32, done[B] bool)."""
        r = np.zeros((self.B, self.max_p), np.float32)
        d = np.zeros(self.B, np.uint8)
        check(self.L.gvec_experience_rewards(self.h, _ptr(r), _ptr(d), MEM_HOST), "gvec_experience_rewards")
        return r, d.astype(bool)

    def observe(self, player=-1):
        """Serializer.StateToTensor: [B][9*stride] for one player, [B][P][9*stride] for player=-1."""
        shape = (self.B, self.max_p, 9 * self.stride) if player < 0 else (self.B, 9 * self.stride)
        out = np.zeros(shape, np.float32)
        check(self.L.gvec_observe(self.h, player, _ptr(out), MEM_HOST), "gvec_observe")
        return out

    def serializer_mask_bits(self):
        """Serializer.GenerateActionMask for all envs/players, packed like legal_action_mask_bits."""
        bits = np.zeros((self.B, self.max_p, self.mask_bytes), np.uint8)
        check(self.L.gvec_serializer_mask(self.h, _ptr(bits), MEM_HOST), "gvec_serializer_mask")
        return bits

    # ---- experience gather support -----------------------------------------------------------------
    def state_bytes_per_env(self):
        return self.L.gvec_state_bytes_per_env(self.h)

    def export_records(self, dst_device_ptr, env_begin=0, n=None):
        n = self.B - env_begin if n is None else n
        check(self.L.gvec_export_records(self.h, env_begin, n, C.c_void_p(int(dst_device_ptr))), "gvec_export_records")

    def import_records(self, src_device_ptr, env_begin=0, n=None):
        n = self.B - env_begin if n is None else n
        check(self.L.gvec_import_records(self.h, env_begin, n, C.c_void_p(int(src_device_ptr))), "gvec_import_records")

    def device_buffer(self, which):
        return self.L.gvec_device_buffer(self.h, which)

    def recorded_actions(self, env_begin=0, n=None):
        """[n][max_players] ACTION_DTYPE: what the device agent played in the last per-turn rollout launch
        (record_agent_actions on)."""
        n = self.B - env_begin if n is None else n
        out = np.zeros((n, self.max_p), ACTION_DTYPE)
        per = self.max_p * ACTION_DTYPE.itemsize
        check(self.L.gvec_read_buffer(self.h, 4, env_begin * per, n * per, _ptr(out)), "gvec_read_buffer")
        return out

    def last_errors(self, env_begin=0, n=None):
        """[n] int32: the per-env codes of the last host-mode step or recorded per-turn rollout launch."""
        n = self.B - env_begin if n is None else n
        out = np.zeros(n, np.int32)
        check(self.L.gvec_read_buffer(self.h, 5, env_begin * 4, n * 4, _ptr(out)), "gvec_read_buffer")
        return out
