"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Builds the library with `make -C oracle` on first use.  Nothing in the product
package imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ODIR = os.path.join(_ROOT, "oracle")
_LIB = None

i32p = C.POINTER(C.c_int32)
i8p = C.POINTER(C.c_int8)
u8p = C.POINTER(C.c_uint8)


class Tile(C.Structure):
    _fields_ = [("owner", C.c_int32), ("army", C.c_int64), ("type", C.c_int32),
                ("visible", C.c_uint32), ("discovered", C.c_uint32)]


class Board(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("t", C.POINTER(Tile))]


class Move(C.Structure):
    _fields_ = [("player_id", C.c_int32), ("from_x", C.c_int32), ("from_y", C.c_int32),
                ("to_x", C.c_int32), ("to_y", C.c_int32), ("move_all", C.c_int32)]


class Capture(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("tile_type", C.c_int32),
                ("capturing_player", C.c_int32), ("previous_owner", C.c_int32), ("previous_army", C.c_int64)]


class Elimination(C.Structure):
    _fields_ = [("eliminated", C.c_int32), ("new_owner", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("fog_of_war", C.c_int32), ("prod_general", C.c_int32), ("prod_city", C.c_int32),
                ("prod_normal", C.c_int32), ("normal_growth_interval", C.c_int32)]


ACTION_DTYPE = np.dtype([("from_x", "i1"), ("from_y", "i1"), ("to_x", "i1"), ("to_y", "i1"),
                         ("flags", "u1"), ("reserved", "u1", (3,))])
assert ACTION_DTYPE.itemsize == 8

STATE_FIELDS = [("army", np.int32, "tile"), ("owner", np.int8, "tile"), ("type", np.uint8, "tile"),
                ("visible", np.uint8, "tile"), ("listed", np.int8, "tile"), ("changed", np.uint8, "tile"),
                ("vis_changed", np.uint8, "tile"), ("turn", np.int32, "env"), ("done", np.uint8, "env"),
                ("winner", np.int8, "env"), ("width", np.int32, "env"), ("height", np.int32, "env"),
                ("players", np.int32, "env"), ("alive", np.uint8, "player"), ("army_count", np.int32, "player"),
                ("tile_count", np.int32, "player"), ("general_idx", np.int32, "player")]


class StateView(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _, _ in STATE_FIELDS]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_ODIR, "liboracle.so")
    src = [os.path.join(_ODIR, f) for f in ("generals_oracle.c", "generals_oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _ODIR, "-s"])
    L = C.CDLL(so)
    L.ora_board_new.restype = C.POINTER(Board)
    L.ora_board_new.argtypes = [C.c_int32, C.c_int32]
    L.ora_board_free.argtypes = [C.POINTER(Board)]
    L.ora_validate.argtypes = [C.POINTER(Board), C.POINTER(Move), C.c_int32]
    L.ora_apply_move.argtypes = [C.POINTER(Board), C.POINTER(Move), u8p, C.POINTER(Capture), i32p]
    L.ora_process_captures.argtypes = [C.POINTER(Capture), C.c_int32, C.POINTER(Elimination)]
    L.ora_tile_set_visible.argtypes = [C.POINTER(Tile), C.c_int32, C.c_int32]
    L.ora_tile_is_visible_to.argtypes = [C.POINTER(Tile), C.c_int32]
    L.ora_params_default.argtypes = [C.POINTER(Params)]
    L.ora_engine_new.restype = C.c_void_p
    L.ora_engine_new.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(Params), i32p, i8p, u8p]
    for f in ("ora_engine_free", "ora_engine_initial_setup", "ora_engine_update_player_stats", "ora_engine_update_fog",
              "ora_engine_process_production", "ora_engine_check_game_over"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = None
    L.ora_engine_step.argtypes = [C.c_void_p, C.POINTER(Move), C.c_int32]
    L.ora_engine_legal_mask.argtypes = [C.c_void_p, C.c_int32, u8p]
    L.ora_engine_player_visibility.argtypes = [C.c_void_p, C.c_int32, u8p, u8p]
    L.ora_engine_is_game_over.argtypes = [C.c_void_p]
    L.ora_engine_winner.argtypes = [C.c_void_p]
    L.ora_engine_board.restype = C.POINTER(Board)
    L.ora_engine_board.argtypes = [C.c_void_p]
    L.ora_engine_turn.argtypes = [C.c_void_p]
    L.ora_engine_set_turn.argtypes = [C.c_void_p, C.c_int32]
    L.ora_engine_set_game_over.argtypes = [C.c_void_p, C.c_int32]
    L.ora_engine_set_fog.argtypes = [C.c_void_p, C.c_int32]
    L.ora_engine_num_players.argtypes = [C.c_void_p]
    L.ora_player_alive.argtypes = [C.c_void_p, C.c_int32]
    L.ora_player_set_alive.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.ora_player_army_count.restype = C.c_int64
    L.ora_player_army_count.argtypes = [C.c_void_p, C.c_int32]
    L.ora_player_general_idx.argtypes = [C.c_void_p, C.c_int32]
    L.ora_player_set_general_idx.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.ora_player_num_owned.argtypes = [C.c_void_p, C.c_int32]
    L.ora_player_owned.restype = i32p
    L.ora_player_owned.argtypes = [C.c_void_p, C.c_int32]
    L.ora_player_set_owned.argtypes = [C.c_void_p, C.c_int32, i32p, C.c_int32]
    L.ora_engine_changed_count.argtypes = [C.c_void_p]
    L.ora_engine_vis_changed_count.argtypes = [C.c_void_p]
    L.ora_engine_changed.restype = u8p
    L.ora_engine_changed.argtypes = [C.c_void_p]
    L.ora_engine_vis_changed.restype = u8p
    L.ora_engine_vis_changed.argtypes = [C.c_void_p]
    L.ora_batch_new.restype = C.c_void_p
    L.ora_batch_new.argtypes = [C.c_int32] * 4 + [C.POINTER(Params)]
    L.ora_batch_free.argtypes = [C.c_void_p]
    L.ora_batch_reset.argtypes = [C.c_void_p, i32p, C.c_int32, i32p, i8p, u8p, i32p, i32p, i32p]
    L.ora_batch_step.argtypes = [C.c_void_p, C.c_void_p, i32p, u8p, C.c_int32]
    L.ora_batch_legal_mask.argtypes = [C.c_void_p, u8p, C.c_int32]
    L.ora_batch_engine.restype = C.c_void_p
    L.ora_batch_engine.argtypes = [C.c_void_p, C.c_int32]
    L.ora_batch_read_state.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(StateView)]
    L.ora_batch_write_state.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(StateView)]
    L.ora_fmix32.restype = C.c_uint32
    L.ora_fmix32.argtypes = [C.c_uint32]
    L.ora_batch_set_agent_mix.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.ora_batch_agent_actions.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_int32]
    L.ora_mapgen.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, i32p, i8p, u8p]
    L.ora_batch_set_pool.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, i32p, i32p, i32p]
    L.ora_batch_rollout.restype = C.c_int64
    L.ora_batch_rollout.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_int32, C.c_int32]
    L.ora_batch_rollout_masks.restype = C.c_int64
    L.ora_batch_rollout_masks.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_int32, C.c_int32, u8p]
    L.ora_batch_next_fog_legacy.argtypes = [C.c_void_p, u8p, u8p]
    L.ora_engine_clone.restype = C.c_void_p
    L.ora_engine_clone.argtypes = [C.c_void_p]
    L.ora_state_to_tensor.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float)]
    L.ora_serializer_mask.argtypes = [C.c_void_p, C.c_int32, u8p]
    L.ora_calculate_reward.restype = C.c_float
    L.ora_calculate_reward.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.ora_army_advantage.restype = C.c_float
    L.ora_army_advantage.argtypes = [C.c_void_p, C.c_int32]
    L.ora_city_changes.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, i32p, i32p]
    L.ora_batch_experience_begin.argtypes = [C.c_void_p]
    L.ora_batch_rewards.argtypes = [C.c_void_p, C.POINTER(C.c_float), u8p]
    L.ora_batch_observe.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float)]
    L.ora_batch_serializer_mask.argtypes = [C.c_void_p, u8p]
    _LIB = L
    return L


def _ptr(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def params(fog=True, prod=(1, 1, 1), interval=25):
    p = Params()
    lib().ora_params_default(C.byref(p))
    p.fog_of_war = int(bool(fog))
    p.prod_general, p.prod_city, p.prod_normal = prod
    p.normal_growth_interval = interval
    return p


def planes_from_tiles(w, h, tiles, stride=None):
    """tiles: list of dicts x,y,owner,army,type -> (army, owner, type) row-major planes."""
    n = stride or w * h
    army = np.zeros(n, np.int32)
    owner = np.full(n, -1, np.int8)
    typ = np.zeros(n, np.uint8)
    for t in tiles:
        i = t["y"] * w + t["x"]
        army[i] = t.get("army", 0)
        owner[i] = t.get("owner", -1)
        typ[i] = t.get("type", 0)
    return army, owner, typ


class OracleEngine:
    """One Engine (internal/game/engine.go) on the oracle."""

    def __init__(self, w, h, players, tiles=(), army=None, owner=None, type=None, fog=True, setup=True, prm=None):
        self.L = lib()
        self.w, self.h, self.p = w, h, players
        if army is None:
            army, owner, type = planes_from_tiles(w, h, tiles)
        self._prm = prm if prm is not None else params(fog=fog)
        self.e = self.L.ora_engine_new(w, h, players, C.byref(self._prm), _ptr(np.ascontiguousarray(army, np.int32), i32p),
                                       _ptr(np.ascontiguousarray(owner, np.int8), i8p),
                                       _ptr(np.ascontiguousarray(type, np.uint8), u8p))
        self._own = True
        if setup:
            self.L.ora_engine_initial_setup(self.e)

    @classmethod
    def wrap(cls, handle, w, h, p):
        self = cls.__new__(cls)
        self.L = lib()
        self.e, self.w, self.h, self.p, self._own = handle, w, h, p, False
        return self

    def __del__(self):
        if getattr(self, "_own", False) and self.e:
            self.L.ora_engine_free(self.e)
            self.e = None

    def tile(self, x, y):
        return self.L.ora_engine_board(self.e).contents.t[y * self.w + x]

    def step(self, moves):
        """moves: list of (player, fx, fy, tx, ty, move_all). Returns error code."""
        arr = (Move * max(1, len(moves)))()
        for i, m in enumerate(moves):
            arr[i] = Move(*[int(v) for v in m])
        return self.L.ora_engine_step(self.e, arr, len(moves))

    def legal_mask(self, player):
        m = np.zeros(self.w * self.h * 4, np.uint8)
        self.L.ora_engine_legal_mask(self.e, player, _ptr(m, u8p))
        return m

    def player_visibility(self, player):
        v = np.zeros(self.w * self.h, np.uint8)
        f = np.zeros(self.w * self.h, np.uint8)
        self.L.ora_engine_player_visibility(self.e, player, _ptr(v, u8p), _ptr(f, u8p))
        return v, f

    def state_to_tensor(self, player):
        out = np.zeros(9 * self.w * self.h, np.float32)
        self.L.ora_state_to_tensor(self.e, player, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def serializer_mask(self, player):
        m = np.zeros(self.w * self.h * 4, np.uint8)
        self.L.ora_serializer_mask(self.e, player, _ptr(m, u8p))
        return m

    def owned(self, p):
        n = self.L.ora_player_num_owned(self.e, p)
        ptr = self.L.ora_player_owned(self.e, p)
        return [ptr[i] for i in range(n)]

    def set_owned(self, p, tiles):
        a = np.ascontiguousarray(tiles, np.int32)
        self.L.ora_player_set_owned(self.e, p, _ptr(a, i32p), len(a))

    @property
    def turn(self):
        return self.L.ora_engine_turn(self.e)

    @property
    def game_over(self):
        return bool(self.L.ora_engine_is_game_over(self.e))

    @property
    def winner(self):
        return self.L.ora_engine_winner(self.e)

    def alive(self, p):
        return bool(self.L.ora_player_alive(self.e, p))

    def army_count(self, p):
        return self.L.ora_player_army_count(self.e, p)

    def general_idx(self, p):
        return self.L.ora_player_general_idx(self.e, p)


def alloc_state(n, stride, max_p, fields=None):
    out = {}
    for name, dt, kind in STATE_FIELDS:
        if fields is not None and name not in fields:
            continue
        shape = {"tile": (n, stride), "env": (n,), "player": (n, max_p)}[kind]
        out[name] = np.zeros(shape, dt)
    return out


def make_view(cls, arrays):
    v = cls()
    for name, _, _ in STATE_FIELDS:
        a = arrays.get(name)
        setattr(v, name, None if a is None else a.ctypes.data)
    return v


class OracleBatch:
    """B engines behind the plane formats of include/generals_vec.h."""

    def __init__(self, num_envs, max_w, max_h, max_p, fog=True, prod=(1, 1, 1), interval=25):
        self.L = lib()
        self.B, self.max_w, self.max_h, self.max_p = num_envs, max_w, max_h, max_p
        self.stride = max_w * max_h
        self.mask_bytes = int(self.L.ora_mask_bytes(self.stride))  # = gvec_mask_bytes()
        self._prm = params(fog, prod, interval)
        self.b = self.L.ora_batch_new(num_envs, max_w, max_h, max_p, C.byref(self._prm))

    def __del__(self):
        if getattr(self, "b", None):
            self.L.ora_batch_free(self.b)
            self.b = None

    def reset(self, army, owner, type, w, h, p, env_ids=None):
        n = len(w)
        ids = None if env_ids is None else np.ascontiguousarray(env_ids, np.int32)
        rc = self.L.ora_batch_reset(self.b, _ptr(ids, i32p), n, _ptr(np.ascontiguousarray(army, np.int32), i32p),
                                    _ptr(np.ascontiguousarray(owner, np.int8), i8p),
                                    _ptr(np.ascontiguousarray(type, np.uint8), u8p),
                                    _ptr(np.ascontiguousarray(w, np.int32), i32p),
                                    _ptr(np.ascontiguousarray(h, np.int32), i32p),
                                    _ptr(np.ascontiguousarray(p, np.int32), i32p))
        assert rc == 0, rc

    def step(self, actions, want_mask=False, threads=1):
        actions = np.ascontiguousarray(actions, ACTION_DTYPE).reshape(self.B, self.max_p)
        err = np.zeros(self.B, np.int32)
        bits = np.zeros((self.B, self.max_p, self.mask_bytes), np.uint8) if want_mask else None
        self.L.ora_batch_step(self.b, actions.ctypes.data, _ptr(err, i32p), _ptr(bits, u8p), threads)
        return (err, bits) if want_mask else err

    def legal_mask(self, threads=1):
        bits = np.zeros((self.B, self.max_p, self.mask_bytes), np.uint8)
        self.L.ora_batch_legal_mask(self.b, _ptr(bits, u8p), threads)
        return bits

    def set_agent_mix(self, noop_per_65536=6554, half_per_65536=19661):
        self.L.ora_batch_set_agent_mix(self.b, int(noop_per_65536), int(half_per_65536))

    def agent_actions(self, seed, invalid_permille=0, threads=1):
        acts = np.zeros((self.B, self.max_p), ACTION_DTYPE)
        self.L.ora_batch_agent_actions(self.b, seed, invalid_permille, acts.ctypes.data, threads)
        return acts

    def read_state(self, env_begin=0, n=None, fields=None):
        n = self.B - env_begin if n is None else n
        arrays = alloc_state(n, self.stride, self.max_p, fields)
        v = make_view(StateView, arrays)
        rc = self.L.ora_batch_read_state(self.b, env_begin, n, C.byref(v))
        assert rc == 0, rc
        return arrays

    def write_state(self, arrays, env_begin=0):
        n = len(next(iter(arrays.values())))
        arrays = {k: np.ascontiguousarray(a) for k, a in arrays.items()}
        v = make_view(StateView, arrays)
        rc = self.L.ora_batch_write_state(self.b, env_begin, n, C.byref(v))
        assert rc == 0, rc

    def next_fog_legacy(self):
        """(visible, discovered) [B][stride]: what the LEGACY fog update (visibility.go:19-144) makes of the
        state the next step starts from."""
        v = np.zeros((self.B, self.stride), np.uint8)
        d = np.zeros((self.B, self.stride), np.uint8)
        self.L.ora_batch_next_fog_legacy(self.b, _ptr(v, u8p), _ptr(d, u8p))
        return v, d

    def experience_begin(self):
        self.L.ora_batch_experience_begin(self.b)

    def rewards(self):
        r = np.zeros((self.B, self.max_p), np.float32)
        d = np.zeros(self.B, np.uint8)
        rc = self.L.ora_batch_rewards(self.b, r.ctypes.data_as(C.POINTER(C.c_float)), _ptr(d, u8p))
        assert rc == 0
        return r, d

    def observe(self, player):
        out = np.zeros((self.B, 9 * self.stride), np.float32)
        self.L.ora_batch_observe(self.b, player, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def serializer_mask(self):
        bits = np.zeros((self.B, self.max_p, self.mask_bytes), np.uint8)
        self.L.ora_batch_serializer_mask(self.b, _ptr(bits, u8p))
        return bits

    def set_pool(self, pool_size, seed, w=None, h=None, p=None):
        cv = lambda a: None if a is None else np.ascontiguousarray(a, np.int32)
        w, h, p = cv(w), cv(h), cv(p)
        self.L.ora_batch_set_pool(self.b, pool_size, seed, _ptr(w, i32p), _ptr(h, i32p), _ptr(p, i32p))

    def rollout(self, turns, seed, invalid_permille=0, threads=1, legal_bits=None):
        """legal_bits: a uint8 [B][max_p][mask_bytes] array -> every env also packs its legal masks after every turn
        (the device's per-turn mask emission: the same work for the timed CPU baseline)."""
        if legal_bits is None:
            return self.L.ora_batch_rollout(self.b, turns, seed, invalid_permille, threads)
        assert legal_bits.dtype == np.uint8 and legal_bits.size == self.B * self.max_p * self.mask_bytes and legal_bits.flags.c_contiguous
        return self.L.ora_batch_rollout_masks(self.b, turns, seed, invalid_permille, threads, legal_bits.ctypes.data_as(u8p))

    def engine(self, env):
        e = self.L.ora_batch_engine(self.b, env)
        return OracleEngine.wrap(e, self.L.ora_engine_board(e).contents.w, self.L.ora_engine_board(e).contents.h,
                                 self.L.ora_engine_num_players(e))


def mapgen_go(seed, w, h, players, cfg=None, stride=None):
    """mapgen.NewGenerator(cfg, rand.New(rand.NewSource(seed))).GenerateMap() on the oracle's restatement of Go's math/rand.
    cfg None = DefaultMapConfig; else [veins, min_len, max_len, city_ratio, city_army, spacing, stages]."""
    n = stride or w * h
    army, owner, typ = np.zeros(n, np.int32), np.full(n, -1, np.int8), np.zeros(n, np.uint8)
    L = lib()
    L.ora_mapgen_go.restype = C.c_int32
    L.ora_mapgen_go.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, i32p, i32p, i8p, u8p]
    c = None if cfg is None else np.ascontiguousarray(cfg, np.int32)
    rc = L.ora_mapgen_go(seed, w, h, players, _ptr(c, i32p), _ptr(army, i32p), _ptr(owner, i8p), _ptr(typ, u8p))
    return rc, army, owner, typ


class GoRand:
    """rand.New(rand.NewSource(seed)) as the oracle restates it."""

    def __init__(self, seed):
        self.L = lib()
        self.L.ora_gorand_new.restype = C.c_void_p
        self.L.ora_gorand_new.argtypes = [C.c_int64]
        self.L.ora_gorand_intn.restype = C.c_int32
        self.L.ora_gorand_intn.argtypes = [C.c_void_p, C.c_int32]
        self.L.ora_gorand_int63.restype = C.c_int64
        self.L.ora_gorand_int63.argtypes = [C.c_void_p]
        self.L.ora_gorand_free.argtypes = [C.c_void_p]
        self.r = self.L.ora_gorand_new(seed)

    def intn(self, n):
        return int(self.L.ora_gorand_intn(self.r, n))

    def int63(self):
        return int(self.L.ora_gorand_int63(self.r))

    def __del__(self):
        if getattr(self, "r", None):
            self.L.ora_gorand_free(self.r)
            self.r = None


def mapgen(seed, env, w, h, players, stride=None):
    n = stride or w * h
    army = np.zeros(n, np.int32)
    owner = np.full(n, -1, np.int8)
    typ = np.zeros(n, np.uint8)
    rc = lib().ora_mapgen(seed, env, w, h, players, _ptr(army, i32p), _ptr(owner, i8p), _ptr(typ, u8p))
    return rc, army, owner, typ
