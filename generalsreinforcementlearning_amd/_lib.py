"""ctypes binding of libgvec_hip.so (the C ABI of include/generals_vec.h)."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class GvecError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gvec error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("num_envs", C.c_int32), ("max_width", C.c_int32), ("max_height", C.c_int32),
                ("max_players", C.c_int32), ("device", C.c_int32), ("fog_of_war", C.c_int32), ("prod_general", C.c_int32),
                ("prod_city", C.c_int32), ("prod_normal", C.c_int32), ("normal_growth_interval", C.c_int32),
                ("auto_reset", C.c_int32), ("reserved", C.c_int32 * 4)]


STATE_FIELDS = ["army", "owner", "type", "visible", "listed", "changed", "vis_changed", "turn", "done", "winner", "width",
                "height", "players", "alive", "army_count", "tile_count", "general_idx"]


class StateView(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in STATE_FIELDS]


class RolloutStats(C.Structure):
    _fields_ = [("env_steps", C.c_int64), ("aborted_turns", C.c_int64), ("games_finished", C.c_int64), ("reserved", C.c_int64)]


class CollectArgs(C.Structure):
    """gvec_collect_args (include/generals_vec.h): every pointer is device memory."""
    _fields_ = ([(n, C.c_int32) for n in ("num_envs", "obs_floats", "max_steps_per_episode", "reserved")]
                + [(n, C.c_int64) for n in ("capacity", "result_capacity")]
                + [(n, C.c_void_p) for n in ("state", "next_state", "action", "reward", "terminated", "truncated", "was_reset", "needs_reset",
                                             "ring_state", "ring_next_state", "ring_action", "ring_reward", "ring_done", "ring_counters",
                                             "episode_reward", "episode_length", "result_reward", "result_length", "result_worker",
                                             "pool_counters", "scratch")])


# every symbol include/generals_vec.h declares: (restype, argtypes)
_vp, _i32, _u64 = C.c_void_p, C.c_int32, C.c_uint64
SYMBOLS = {
    "gvec_abi_version": (_i32, []),
    "gvec_config_default": (_i32, [C.POINTER(Config)]),
    "gvec_create": (_i32, [C.POINTER(Config), C.POINTER(_vp)]),
    "gvec_destroy": (_i32, [_vp]),
    "gvec_create_sharded": (_i32, [C.POINTER(Config), C.POINTER(C.c_int32), _i32, C.POINTER(_vp)]),
    "gvec_num_shards": (_i32, [_vp]),
    "gvec_shard": (_i32, [_vp, _i32, C.POINTER(_vp), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gvec_gather_experience_records": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gvec_set_stream": (_i32, [_vp, _vp]),
    "gvec_synchronize": (_i32, [_vp]),
    "gvec_last_error": (C.c_char_p, []),
    "gvec_host_alloc": (_i32, [_u64, C.POINTER(_vp)]),
    "gvec_host_free": (_i32, [_vp]),
    "gvec_num_envs": (_i32, [_vp]),
    "gvec_tile_stride": (_i32, [_vp]),
    "gvec_mask_bytes": (_i32, [_vp]),
    "gvec_state_bytes_per_env": (C.c_int64, [_vp]),
    "gvec_reset": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32]),
    "gvec_reset_generated": (_i32, [_vp, _u64, _vp, _vp, _vp]),
    "gvec_reset_go_seeded": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "gvec_build_board_pool": (_i32, [_vp, _i32, _u64, _vp, _vp, _vp]),
    "gvec_step": (_i32, [_vp, _vp, _vp, _vp, _i32]),
    "gvec_legal_mask": (_i32, [_vp, _vp, _i32]),
    "gvec_player_visibility": (_i32, [_vp, _i32, _vp, _vp, _i32]),
    "gvec_read_state": (_i32, [_vp, _i32, _i32, C.POINTER(StateView), _i32]),
    "gvec_write_state": (_i32, [_vp, _i32, _i32, C.POINTER(StateView), _i32]),
    "gvec_rollout": (_i32, [_vp, _i32, _u64, _i32, _i32, C.POINTER(RolloutStats)]),
    "gvec_rollout_range": (_i32, [_vp, _i32, _i32, _i32, _u64, _i32]),
    "gvec_set_agent_mix": (_i32, [_vp, _i32, _i32]),
    "gvec_agent_actions": (_i32, [_vp, _u64, _i32, _vp, _i32]),
    "gvec_counters": (_i32, [_vp, C.POINTER(RolloutStats)]),
    "gvec_step_traffic_bytes": (_i32, [_vp, C.POINTER(C.c_int64)]),
    "gvec_experience_begin": (_i32, [_vp]),
    "gvec_experience_begin_range": (_i32, [_vp, _i32, _i32]),
    "gvec_experience_rewards": (_i32, [_vp, _vp, _vp, _i32]),
    "gvec_experience_record_layout": (_i32, [_vp, C.POINTER(C.c_int32)]),
    "gvec_experience_record_bytes": (_i32, [_vp]),
    "gvec_experience_records": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "gvec_record_agent_actions": (_i32, [_vp, _i32]),
    "gvec_expand_experience_records": (_i32, [_i32, _vp, C.POINTER(C.c_int32), _vp, _i32, _vp, _vp, _vp, _vp]),
    "gvec_pool_collect": (_i32, [_i32, _vp, C.POINTER(CollectArgs)]),
    "gvec_pool_collect_scratch_bytes": (_u64, [_i32]),
    "gvec_gym_observe": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "gvec_gym_finish_step": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gvec_gym_actions": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gvec_gym_step": (_i32, [_vp, _i32, _u64] + [_vp] * 3 + [_i32] + [_vp] * 11),
    "gvec_stream_delta_cap": (_i32, [_vp]),
    "gvec_stream_deltas": (_i32, [_vp, _i32, _vp, _vp, _vp, _i32]),
    "gvec_stream_deltas_packed": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, C.c_int64, C.POINTER(C.c_int64)]),
    "gvec_observe": (_i32, [_vp, _i32, _vp, _i32]),
    "gvec_serializer_mask": (_i32, [_vp, _vp, _i32]),
    "gvec_export_records": (_i32, [_vp, _i32, _i32, _vp]),
    "gvec_import_records": (_i32, [_vp, _i32, _i32, _vp]),
    "gvec_device_buffer": (_vp, [_vp, _i32]),
    "gvec_read_buffer": (_i32, [_vp, _i32, _u64, _u64, _vp]),
    "gvec_selftest": (_i32, [_i32]),
}


def lib_path():
    """The in-tree build; GVEC_LIB names another build of the same library (A/B measurements of two builds)."""
    return os.environ.get("GVEC_LIB") or os.path.join(_PKG, "libgvec_hip.so")


def load_from(path):
    """Loads one build of the HIP library from `path` (not cached: A/B benchmarking of two builds
    in one process uses this directly)."""
    if not os.path.exists(path):
        raise GvecError(-2, f"{path} not found: build it with `python generalsreinforcementlearning_amd/csrc/build.py` "
                            "(hipcc, gfx950). This package has no CPU fallback.")
    L = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if L.gvec_abi_version() != 2:
        raise GvecError(-1, "ABI version mismatch")
    return L


def load():
    """Loads the HIP library.  Fails loudly if it has not been built: there is no fallback."""
    global _LIB
    if _LIB is None:
        _LIB = load_from(lib_path())
    return _LIB


def lib():
    return load()


def check(rc, what="", L=None):
    if rc < 0:
        msg = (L or load()).gvec_last_error()
        raise GvecError(rc, f"{what}: {msg.decode() if msg else ''}")
    return rc
