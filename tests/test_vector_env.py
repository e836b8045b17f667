"""The drop-in vector env.  Two layers of checker, both test infrastructure:
  1. a scalar, line-by-line restatement of python/generals_gym/generals_env.py's helpers and of the server's proto fog
     rules (internal/grpc/gameserver/server.go:526-582) - this file;
  2. its batched numpy form + GeneralsEnv.step's bookkeeping (tests/_gym_reference.py: NumpyReferenceVecEnv), pinned to (1)
     here on CPU with states from the oracle.
The product (GeneralsVecEnv: the HIP gym kernels, one launch per step) is compared with (2) output for output under -m gpu.
The reference env itself cannot run here (needs gymnasium and a live Go server): parity unpinned beyond its source text."""
import math

import numpy as np
import pytest

import _harness as H
import _oracle as O
import _gym_reference as G

DIRS = [(0, -1), (1, 0), (0, 1), (-1, 0)]


class Tile:  # gamev1.Tile as the gym env reads it
    def __init__(self, type_, owner_id, army_count, visible):
        self.type, self.owner_id, self.army_count, self.visible = type_, owner_id, army_count, visible


def ref_proto_tiles(owner, army, typ, vis, fog):
    """server.go:556-582, one env."""
    out = []
    for i in range(len(owner)):
        t = Tile(int(typ[i]), int(owner[i]), int(army[i]), bool(vis[i]))
        fow = bool(fog[i])
        if not t.visible and not fow:
            t.type, t.owner_id, t.army_count = 0, -1, 0
        elif fow and not t.visible:
            t.owner_id, t.army_count = -1, 0
        out.append(t)
    return out


def ref_get_observation(tiles, w, h, player_id, turn_count, max_turns):
    """generals_env.py:291-342"""
    obs = np.zeros((9, h, w), dtype=np.float32)
    for y in range(h):
        for x in range(w):
            tile = tiles[y * w + x]
            if tile.visible:
                obs[0, y, x] = 1.0
            if tile.owner_id == player_id:
                obs[1, y, x] = 0.5
            elif tile.owner_id >= 0:
                obs[1, y, x] = 1.0
            else:
                obs[1, y, x] = 0.0
            if tile.army_count > 0:
                obs[2, y, x] = np.log(tile.army_count + 1) / 10.0
            if tile.type == 0:
                obs[3, y, x] = 1.0
            elif tile.type == 3:
                obs[4, y, x] = 1.0
            elif tile.type == 2:
                obs[5, y, x] = 1.0
            elif tile.type == 1:
                obs[6, y, x] = 1.0
    obs[7, :, :] = min(turn_count / max_turns, 1.0)
    return obs


def ref_valid_mask(tiles, w, h, player_id):
    """generals_env.py:344-387"""
    mask = np.zeros(w * h * 5, dtype=bool)
    for y in range(h):
        for x in range(w):
            idx = y * w + x
            tile = tiles[idx]
            if tile.owner_id != player_id or tile.army_count <= 1:
                continue
            for direction, (dx, dy) in enumerate(DIRS):
                nx, ny = x + dx, y + dy
                if 0 <= nx < w and 0 <= ny < h:
                    if tiles[ny * w + nx].type != 3:
                        mask[idx * 5 + direction] = True
                        mask[idx * 5 + 4] = True
    return mask


def ref_decode(action_idx, w, h):
    """generals_env.py:402-425"""
    from_idx, move_info = action_idx // 5, action_idx % 5
    from_x, from_y = from_idx % w, from_idx // w
    is_half = bool(move_info == 4)
    if move_info < 4:
        dx, dy = DIRS[move_info]
        to_x, to_y = from_x + dx, from_y + dy
    else:
        for dx, dy in DIRS:
            to_x, to_y = from_x + dx, from_y + dy
            if 0 <= to_x < w and 0 <= to_y < h:
                break
    return from_x, from_y, to_x, to_y, is_half


def ref_reward(prev, cur, e, player_id):
    """generals_env.py:499-561 on PlayerState fields (server.go:526-553)."""
    if cur["done"][e]:
        return 100.0 if cur["winner"][e] == player_id else -100.0
    reward = 0.0
    reward += (int(cur["tile_count"][e, player_id]) - int(prev["tile_count"][e, player_id])) * 1.0
    reward += (int(cur["army_count"][e, player_id]) - int(prev["army_count"][e, player_id])) * 0.01
    for q in range(cur["alive"].shape[1]):
        if q != player_id and prev["alive"][e, q] and not cur["alive"][e, q]:
            reward += 50.0
    return reward


def _views(ora, B, w, h, player):
    st = ora.read_state()
    vis = np.zeros((B, w * h), np.uint8)
    fog = np.zeros((B, w * h), np.uint8)
    for e in range(B):
        v, f = ora.engine(e).player_visibility(player)
        vis[e], fog[e] = v, f
    return st, vis, fog


def test_pure_functions_match_scalar_restatement():
    B, w, h, P = 24, 9, 7, 3
    sizes = [(w, h, P)] * B
    army, owner, typ, ws, hs, ps = H.gen_boards(5, sizes, w, h)
    ora = O.OracleBatch(B, w, h, P)
    ora.reset(army, owner, typ, ws, hs, ps)
    prev = None
    for k in range(80):
        ora.step(ora.agent_actions(3, 5))
        if k % 8 != 0:
            continue
        st, vis, fog = _views(ora, B, w, h, 0)
        view = G.proto_view(st["owner"], st["army"], st["type"], vis, fog)
        tc = np.full(B, k + 1)
        obs = G.build_observation(view, 0, tc, 50, w, h)
        mask = G.valid_actions_mask(view, 0, w, h)
        for e in range(B):
            tiles = ref_proto_tiles(st["owner"][e], st["army"][e], st["type"][e], vis[e], fog[e])
            assert np.array_equal(obs[e], ref_get_observation(tiles, w, h, 0, k + 1, 50)), (k, e)
            assert np.array_equal(mask[e], ref_valid_mask(tiles, w, h, 0)), (k, e)
        stats = {f: st[f] for f in ("done", "winner", "alive", "army_count", "tile_count")}
        if prev is not None:
            r = G.calculate_reward(prev, stats, 0)
            for e in range(B):
                assert r[e] == ref_reward(prev, stats, e, 0), (k, e)
        prev = stats
    assert obs.dtype == np.float32 and obs.shape == (B, 9, h, w) and mask.shape == (B, w * h * 5)


def test_decode_actions_matches_reference_quirk():
    w, h = 6, 4
    acts = np.arange(w * h * 5)
    fx, fy, tx, ty, half, d = G.decode_actions(acts, w, h)
    for a in acts:
        assert (fx[a], fy[a], tx[a], ty[a], bool(half[a])) == ref_decode(int(a), w, h), a


class OracleBackedEngine:
    """The VecEngine surface GeneralsVecEnv uses, served by the CPU oracle (same plane formats)."""

    def __init__(self, num_envs, width, height, players, fog_of_war=True, device=0, auto_reset=False):
        self.B, self.w, self.h, self.p = num_envs, width, height, players
        self.ora = O.OracleBatch(num_envs, width, height, players, fog=fog_of_war)

    def reset_generated(self, seed):
        army, owner, typ, ws, hs, ps = H.gen_boards(seed, [(self.w, self.h, self.p)] * self.B, self.w, self.h)
        self.ora.reset(army, owner, typ, ws, hs, ps)

    def build_board_pool(self, n, seed):
        self.ora.set_pool(n, seed)

    def game_state(self, fields=None):
        return self.ora.read_state(fields=fields)

    def write_state(self, arrays):
        self.ora.write_state(arrays)

    def compute_player_visibility(self, player):
        vis = np.zeros((self.B, self.w * self.h), bool)
        fog = np.zeros((self.B, self.w * self.h), bool)
        for e in range(self.B):
            v, f = self.ora.engine(e).player_visibility(player)
            vis[e], fog[e] = v, f
        return vis, fog

    def agent_actions(self, seed):
        return self.ora.agent_actions(seed)

    def step(self, acts):
        return self.ora.step(acts)

    def close(self):
        pass


def _episode_flow(env):
    assert env.single_observation_shape == (9, 8, 8) and env.single_action_n == 8 * 8 * 5
    obs, info = env.reset()
    assert obs.shape == (64, 9, 8, 8) and obs.dtype == np.float32 and obs.min() >= 0.0 and obs.max() <= 1.0
    assert info["valid_actions_mask"].shape == (64, 320) and info["valid_actions_mask"].any(1).all()
    rng = np.random.default_rng(0)
    seen = {"invalid": 0, "terminated": 0, "truncated": 0, "reset": 0}
    total_reward = np.zeros(64)
    for k in range(130):
        mask = info["valid_actions_mask"]
        acts = np.array([rng.choice(np.flatnonzero(m)) if m.any() else 0 for m in mask])
        if k % 7 == 3:
            acts[:4] = [int(np.flatnonzero(~m)[0]) for m in mask[:4]]  # deliberately invalid
        turn_before = info["turn"].copy()
        obs, reward, terminated, truncated, info = env.step(acts)
        inval = info["invalid_action"]
        if k % 7 == 3:
            assert (inval[:4] | info["reset"][:4]).all() and (reward[:4][~info["reset"][:4]] == -0.1).all()
            assert (info["turn"][:4][~info["reset"][:4]] == turn_before[:4][~info["reset"][:4]]).all()  # game did not advance
        seen["invalid"] += int(inval.sum())
        seen["terminated"] += int(terminated.sum())
        seen["truncated"] += int(truncated.sum())
        seen["reset"] += int(info["reset"].sum())
        assert (reward[terminated] != 0).all() and set(np.unique(np.abs(reward[terminated]))) <= {100.0}
        assert (info["turn"][info["reset"]] == 0).all()
        total_reward += reward
    assert seen["invalid"] > 0 and seen["truncated"] + seen["terminated"] > 0 and seen["reset"] > 0
    env.close()


def test_reference_env_episode_flow_on_oracle_engine():
    """The checker's own episode flow, on the CPU oracle."""
    _episode_flow(G.NumpyReferenceVecEnv(OracleBackedEngine(64, 8, 8, 2), 64, 8, 8, max_players=2, max_turns=40, seed=3))


def _hip_reference_env(num_envs, board_width, board_height, max_players=2, fog_of_war=True, max_turns=500, seed=0, board_pool=1024):
    """NumpyReferenceVecEnv over a HIP VecEngine: same boards, same pool, same opponents as GeneralsVecEnv with these arguments."""
    import generalsreinforcementlearning_amd as g
    eng = g.VecEngine(num_envs, board_width, board_height, max_players, fog_of_war=fog_of_war, auto_reset=True)
    return G.NumpyReferenceVecEnv(eng, num_envs, board_width, board_height, max_players=max_players, fog_of_war=fog_of_war, max_turns=max_turns,
                                  seed=seed, board_pool=board_pool)


@pytest.mark.gpu
def test_vector_env_api_and_episode_flow():
    from generalsreinforcementlearning_amd.vector_env import GeneralsVecEnv
    _episode_flow(GeneralsVecEnv(64, board_width=8, board_height=8, max_players=2, max_turns=40, seed=3))


def test_vector_env_needs_a_gpu_and_has_no_host_path():
    """There is one execution path; without a GPU construction fails loudly (no silent numpy fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from generalsreinforcementlearning_amd import GvecError
    from generalsreinforcementlearning_amd.vector_env import GeneralsVecEnv
    with pytest.raises(GvecError):
        GeneralsVecEnv(4, board_width=8, board_height=8)
    import generalsreinforcementlearning_amd.vector_env as V
    for name in ("proto_view", "build_observation", "valid_actions_mask", "decode_actions", "calculate_reward"):
        assert not hasattr(V, name), f"{name}: the numpy checker must not live in the product"


@pytest.mark.gpu
@pytest.mark.parametrize("W,Hh,P", [(10, 10, 2), (15, 15, 2), (7, 5, 2), (25, 25, 4), (21, 13, 3)],
                         ids=["10x10", "15x15_odd_planes", "7x5_odd_planes", "25x25_odd_planes", "21x13_odd_planes"])
def test_vector_env_matches_scalar_restatement_on_hip_states(W, Hh, P):
    """Observation and mask of the gym kernels against the scalar restatement of generals_env.py - also on boards whose
    planes (W*H floats) start on no 16-byte boundary: those leave through the aligned-window stores of gym_emit."""
    from generalsreinforcementlearning_amd.vector_env import GeneralsVecEnv
    env = GeneralsVecEnv(32, board_width=W, board_height=Hh, max_players=P, max_turns=100, seed=9)
    obs, info = env.reset()
    rng = np.random.default_rng(1)
    for k in range(40 if W * Hh < 300 else 12):
        acts = np.array([rng.choice(np.flatnonzero(m)) if m.any() else 0 for m in info["valid_actions_mask"]])
        obs, reward, terminated, truncated, info = env.step(acts)
        st = env.engine.game_state()
        vis, fog = env.engine.compute_player_visibility(0)
        for e in range(0, 32, 5):
            tiles = ref_proto_tiles(st["owner"][e], st["army"][e], st["type"][e], vis[e], fog[e])
            assert np.array_equal(obs[e], ref_get_observation(tiles, W, Hh, 0, int(info["turn"][e]), 100)), (k, e)
            assert np.array_equal(info["valid_actions_mask"][e], ref_valid_mask(tiles, W, Hh, 0)), (k, e)
    env.close()


# ---- serializer action index (internal/experience/serializer.go:179-223), reference vectors ----
def _kats(kind):
    import json
    import os
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json")))
    return [pytest.param(c, id=c["name"]) for c in d["cases"] if c["kind"] == kind]


def _one_action(fx, fy, tx, ty):
    from generalsreinforcementlearning_amd.vec_engine import ACTION_DTYPE
    a = np.zeros(1, ACTION_DTYPE)
    a["from_x"], a["from_y"], a["to_x"], a["to_y"], a["flags"] = fx, fy, tx, ty, 1
    return a


@pytest.mark.parametrize("c", _kats("action_to_index"))
def test_golden_action_to_index(c):
    from generalsreinforcementlearning_amd.experience import action_to_index
    assert int(action_to_index(_one_action(*c["from"], *c["to"]), c["w"])[0]) == c["expect"]


@pytest.mark.parametrize("c", _kats("index_to_action"))
def test_golden_index_to_action(c):
    from generalsreinforcementlearning_amd.experience import action_to_index, index_to_action
    fx, fy, tx, ty = (int(v) for v in index_to_action(c["index"], c["w"], c["h"]))
    assert [fx, fy] == c["expect"]["from"] and [tx, ty] == c["expect"]["to"]
    assert int(action_to_index(_one_action(fx, fy, tx, ty), c["w"])[0]) == c["index"]  # round trip


@pytest.mark.parametrize("c", _kats("action_indices_distinct"))
def test_golden_action_indices_distinct(c):
    from generalsreinforcementlearning_amd.experience import action_to_index
    idx = {int(action_to_index(_one_action(*m), c["w"])[0]) for m in c["moves"]}
    assert len(idx) == c["expect_distinct"]


# ---- device mode: the gym kernels (gvec_gym_observe / gvec_gym_actions) ------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("fog,P", [(True, 2), (True, 4), (False, 3)], ids=["fog_p2", "fog_p4", "nofog_p3"])
def test_device_vector_env_equals_host_vector_env(fog, P):
    """device_outputs=True (observation / mask / reward / flags as CUDA tensors from the gym kernels, actions decoded on
    the device) against the host-numpy mode, which the tests above pin to the scalar restatement of generals_env.py:
    every output of every step, bit for bit - invalid actions, half moves, terminations, truncations and re-deals included."""
    import torch
    kw = dict(board_width=9, board_height=8, max_players=P, fog_of_war=fog, max_turns=30, seed=5, board_pool=64)
    from generalsreinforcementlearning_amd.vector_env import GeneralsVecEnv
    host = _hip_reference_env(48, **kw)
    dev = GeneralsVecEnv(48, device_outputs=True, **kw)
    ho, hi = host.reset()
    do, di = dev.reset()
    assert do.is_cuda and do.dtype == torch.float32 and di["valid_actions_mask"].dtype == torch.bool
    assert np.array_equal(do.cpu().numpy().view(np.uint32), ho.view(np.uint32)) and np.array_equal(di["valid_actions_mask"].cpu().numpy(), hi["valid_actions_mask"])
    rng = np.random.default_rng(2)
    seen = {"term": 0, "trunc": 0, "invalid": 0, "error": 0, "half": 0}
    for k in range(150):
        mask = hi["valid_actions_mask"]
        acts = np.array([rng.choice(np.flatnonzero(m)) if m.any() else 0 for m in mask])
        if k % 5 == 2:
            acts[:6] = [int(np.flatnonzero(~m)[rng.integers(0, 20)]) for m in mask[:6]]      # invalid
            acts[6] = -3
            acts[7] = 9 * 8 * 5 + 4                                                          # out of range
        if k % 3 == 0:                                                                        # half moves (index 4) where legal
            for e in range(8, 20):
                hm = np.flatnonzero(mask[e][4::5])
                if len(hm):
                    acts[e] = int(hm[rng.integers(0, len(hm))]) * 5 + 4
                    seen["half"] += 1
        ho, hr, hterm, htrunc, hi = host.step(acts)
        do, dr, dterm, dtrunc, di = dev.step(torch.from_numpy(acts).cuda())
        assert np.array_equal(do.cpu().numpy().view(np.uint32), ho.view(np.uint32)), k
        assert np.array_equal(di["valid_actions_mask"].cpu().numpy(), hi["valid_actions_mask"]), k
        assert dr.dtype == torch.float64 and np.array_equal(dr.cpu().numpy().view(np.uint64), np.asarray(hr, np.float64).view(np.uint64)), k
        assert np.array_equal(dterm.cpu().numpy(), hterm) and np.array_equal(dtrunc.cpu().numpy(), htrunc)
        for f in ("turn", "invalid_action", "error", "winner", "reset"):
            assert np.array_equal(di[f].cpu().numpy(), np.asarray(hi[f])), (k, f)
        seen["term"] += int(hterm.sum()); seen["trunc"] += int(htrunc.sum())
        seen["invalid"] += int(hi["invalid_action"].sum()); seen["error"] += int(hi["error"].sum())
    assert seen["trunc"] > 0 and seen["invalid"] > 0 and seen["half"] > 0
    # the two engines went through identical states
    H.assert_states_equal(dev.engine.game_state(), host.engine.game_state(), "device vs host vector env")
    host.close(); dev.close()


@pytest.mark.gpu
def test_default_numpy_mode_runs_on_the_gym_kernels_and_equals_the_reference_env():
    """The default mode (numpy in / numpy out) is the device path plus pinned D2H copies: same values, dtypes and
    shapes as the numpy reference env (tests/_gym_reference.py), step by step."""
    kw = dict(board_width=10, board_height=10, max_players=3, max_turns=25, seed=11, board_pool=32)
    from generalsreinforcementlearning_amd.vector_env import GeneralsVecEnv
    ref = _hip_reference_env(40, **kw)
    fast = GeneralsVecEnv(40, **kw)
    ro, ri = ref.reset()
    fo, fi = fast.reset()
    rng = np.random.default_rng(4)
    for k in range(80):
        assert isinstance(fo, np.ndarray) and fo.dtype == ro.dtype and fo.shape == ro.shape and np.array_equal(fo.view(np.uint32), ro.view(np.uint32)), k
        for f in ri:
            if f == "player_id":
                continue
            a, b = np.asarray(fi[f]), np.asarray(ri[f])
            assert a.dtype == b.dtype and np.array_equal(a, b), (k, f)
        mask = ri["valid_actions_mask"]
        acts = np.array([rng.choice(np.flatnonzero(m)) if m.any() else 0 for m in mask])
        if k % 4 == 1:
            acts[:5] = [int(np.flatnonzero(~m)[3]) for m in mask[:5]]
        ro, rr, rt, ru, ri = ref.step(acts)
        fo, fr, ft, fu, fi = fast.step(acts)
        for a, b in ((fr, rr), (ft, rt), (fu, ru)):
            a, b = np.asarray(a), np.asarray(b)
            assert a.dtype == b.dtype and np.array_equal(a, b), k
    ref.close(); fast.close()


@pytest.mark.gpu
def test_gym_observation_log_channel_is_numpys_float64_log():
    """Channel 2 = float32(np.log(army + 1) / 10.0) with the log in float64 (generals_env.py:324-326): checked for
    every army 1 .. 2^18 and for 2^18 samples up to 2^31 - 2, on a board where every tile is visible (fog off)."""
    import torch
    import generalsreinforcementlearning_amd as g
    from generalsreinforcementlearning_amd._lib import check
    B, w, h = 256, 32, 32
    eng = g.VecEngine(B, w, h, 2, fog_of_war=False, stream=torch.cuda.current_stream().cuda_stream)
    eng.reset_generated(4)
    st = eng.game_state(fields=("army", "type"))
    rng = np.random.default_rng(3)
    for values in (np.arange(1, B * w * h + 1, dtype=np.int64), rng.integers(1, 2 ** 31 - 1, B * w * h, dtype=np.int64)):
        army = values.astype(np.int32).reshape(B, w * h)
        eng.write_state({"army": army})
        obs = torch.zeros((B, 9, w * h), dtype=torch.float32, device="cuda")
        mask = torch.zeros((B, w * h * 5), dtype=torch.uint8, device="cuda")
        tc = torch.zeros(B, dtype=torch.int64, device="cuda")
        check(eng.L.gvec_gym_observe(eng.h, 0, tc.data_ptr(), 10, obs.data_ptr(), mask.data_ptr(), None, None, None), "gvec_gym_observe")
        got = obs[:, 2].cpu().numpy()
        want = (np.log(army.astype(np.int64) + 1) / 10.0).astype(np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int((got != want).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,P,fog", [(9, 8, 2, True), (20, 20, 4, True), (12, 13, 3, False), (25, 25, 4, True), (32, 32, 8, True),
                                       (5, 5, 2, True), (16, 16, 2, True), (21, 21, 8, True), (25, 24, 8, False), (30, 32, 2, True), (11, 11, 4, True)],
                         ids=["9x8_p2", "20x20_p4", "12x13_p3_nofog", "25x25_p4", "32x32_p8", "5x5_p2", "16x16_p2", "21x21_p8", "25x24_p8_nofog",
                              "30x32_p2", "11x11_p4"])
def test_gym_step_equals_the_four_call_composition(w, h, P, fog):
    """gvec_gym_step (ONE launch) == gvec_agent_actions -> gvec_gym_actions -> gvec_step -> gvec_gym_finish_step on a twin
    engine: every output of every step bit for bit, and the two engines' states at the end - over every register layout of
    the step kernel, with invalid / out-of-range / half-move actions, terminations, truncations and re-deals."""
    import torch
    import generalsreinforcementlearning_amd as g
    from generalsreinforcementlearning_amd._lib import check
    B, max_turns, n = 64, 25, w * h
    dev = torch.device("cuda")
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)

    class Side:
        def __init__(self):
            self.e = g.VecEngine(B, w, h, P, fog_of_war=fog, auto_reset=True, stream=torch.cuda.current_stream().cuda_stream)
            self.e.reset_generated(77)
            self.e.build_board_pool(16, 5)
            self.turn, self.obs, self.mask = z(B, torch.int64), z((B, 9, n), torch.float32), z((B, n * 5), torch.uint8)
            self.out = {k: z(B, dt) for k, dt in (("reward", torch.float64), ("terminated", torch.uint8), ("truncated", torch.uint8),
                                                  ("winner", torch.int8), ("needs_reset", torch.uint8), ("turn_out", torch.int64),
                                                  ("played", torch.uint8), ("invalid", torch.uint8), ("error", torch.uint8))}
            self.resetting = z(B, torch.uint8)
            self.acts = z((B, P, 8), torch.uint8)
            e = self.e
            check(e.L.gvec_gym_observe(e.h, 0, self.turn.data_ptr(), max_turns, self.obs.data_ptr(), self.mask.data_ptr(), None, None, None))

        def outputs(self):
            torch.cuda.synchronize()
            d = {k: v.cpu().numpy().copy() for k, v in self.out.items()}
            d["obs"], d["mask"], d["turn"] = self.obs.cpu().numpy().view(np.uint32).copy(), self.mask.cpu().numpy().copy(), self.turn.cpu().numpy().copy()
            return d

    one, four = Side(), Side()
    rng = np.random.default_rng(8)
    seen = {"term": 0, "trunc": 0, "invalid": 0, "error": 0, "reset": 0}
    for k in range(120):
        mask = one.mask.cpu().numpy().astype(bool)
        acts = np.array([rng.choice(np.flatnonzero(m)) if m.any() else 0 for m in mask], np.int64)
        if k % 4 == 1:
            acts[:5] = [int(np.flatnonzero(~m)[rng.integers(0, 10)]) for m in mask[:5]]
            acts[5], acts[6] = -7, n * 5 + 3
        if k % 3 == 0:
            for e_ in range(8, 24):
                hm = np.flatnonzero(mask[e_][4::5])
                if len(hm):
                    acts[e_] = int(hm[rng.integers(0, len(hm))]) * 5 + 4
        ta = torch.from_numpy(acts).to(dev)
        seed = 1000 * k + 3
        o, e = one.out, one.e
        prev_mask_four = four.mask.clone()
        check(e.L.gvec_gym_step(e.h, 0, seed, ta.data_ptr(), one.resetting.data_ptr(), one.turn.data_ptr(), max_turns, one.obs.data_ptr(),
                                one.mask.data_ptr(), o["reward"].data_ptr(), o["terminated"].data_ptr(), o["truncated"].data_ptr(),
                                o["winner"].data_ptr(), o["needs_reset"].data_ptr(), o["turn_out"].data_ptr(), o["played"].data_ptr(),
                                o["invalid"].data_ptr(), o["error"].data_ptr()), "gvec_gym_step")
        o, e = four.out, four.e
        check(e.L.gvec_agent_actions(e.h, seed, 0, four.acts.data_ptr(), 1))
        check(e.L.gvec_gym_actions(e.h, 0, ta.data_ptr(), prev_mask_four.data_ptr(), four.resetting.data_ptr(), four.acts.data_ptr(),
                                   o["played"].data_ptr(), o["invalid"].data_ptr(), o["error"].data_ptr()))
        e.step_device(four.acts.data_ptr())
        check(e.L.gvec_gym_finish_step(e.h, 0, four.turn.data_ptr(), max_turns, four.resetting.data_ptr(), o["played"].data_ptr(),
                                       four.obs.data_ptr(), four.mask.data_ptr(), o["reward"].data_ptr(), o["terminated"].data_ptr(),
                                       o["truncated"].data_ptr(), o["winner"].data_ptr(), o["needs_reset"].data_ptr(), o["turn_out"].data_ptr()))
        a, b = one.outputs(), four.outputs()
        for f in a:
            x, y = a[f], b[f]
            if f == "reward":
                x, y = x.view(np.uint64), y.view(np.uint64)
            assert np.array_equal(x, y), (k, f, np.flatnonzero((x != y).reshape(B, -1).any(1))[:8])
        seen["term"] += int(a["terminated"].sum()); seen["trunc"] += int(a["truncated"].sum())
        seen["invalid"] += int(a["invalid"].sum()); seen["error"] += int(a["error"].sum()); seen["reset"] += int(one.resetting.sum().item())
        one.resetting.copy_(one.out["needs_reset"])
        four.resetting.copy_(four.out["needs_reset"])
    assert seen["trunc"] > 0 and seen["invalid"] > 0 and seen["reset"] > 0
    H.assert_states_equal(one.e.game_state(), four.e.game_state(), "gym_step vs composition")


@pytest.mark.gpu
def test_single_env_facade_has_the_reference_env_s_shape():
    """GeneralsEnv (the single-env object code written for the reference instantiates): constructor keywords, spaces,
    reset / step tuples and info keys of generals_env.py:48-289; an episode loop like python/test_gym_env.py's.  (The
    reference-shaped env_factory(worker_id) driving a pool of them: tests/test_env_pool.py, generals_gym.ParallelEnvPool.)"""
    from generalsreinforcementlearning_amd.vector_env import GeneralsEnv
    env = GeneralsEnv(server_address="localhost:50051", board_width=6, board_height=5, max_players=2, fog_of_war=False, max_turns=30)
    assert env.observation_space.shape == (9, 5, 6) and env.observation_space.dtype == np.float32 and env.action_space.n == 150
    obs, info = env.reset()
    assert obs.shape == (9, 5, 6) and obs.dtype == np.float32 and env.observation_space.contains(obs)
    assert set(info) == {"game_id", "player_id", "valid_actions_mask", "turn"} and info["turn"] == 0 and info["valid_actions_mask"].shape == (150,)
    mask = info["valid_actions_mask"]
    o2, r, term, trunc, inf = env.step(int(np.flatnonzero(~mask)[0]))                    # an action the mask rejects
    assert (r, term, trunc, inf) == (-0.1, False, False, {"invalid_action": True}) and np.array_equal(o2, obs)
    steps, ended = 0, False
    for episode in range(3):
        obs, info = env.reset()
        for _ in range(40):
            a = env.action_space.sample(info["valid_actions_mask"])
            obs, r, term, trunc, info = env.step(a)
            steps += 1
            assert isinstance(r, float) and isinstance(term, bool) and isinstance(trunc, bool)
            assert set(info) == {"turn", "valid_actions_mask", "game_status", "winner"}
            assert info["game_status"] == ("GAME_STATUS_FINISHED" if term else "GAME_STATUS_IN_PROGRESS") and (info["winner"] is None) == (not term)
            if term or trunc:
                ended = True
                assert trunc == (info["turn"] >= 30) or term
                break
    assert ended and steps > 30
    env.render()
    env.close()


@pytest.mark.gpu
def test_step_with_supplied_opponent_moves_equals_the_one_launch_step():
    """GeneralsVecEnv.step(actions, other_actions=...) - the other players' moves supplied by the caller - with exactly the
    moves the on-device agent would have drawn equals the one-launch step, output for output and state for state."""
    from generalsreinforcementlearning_amd.vector_env import GeneralsVecEnv
    B = 96
    mk = lambda: GeneralsVecEnv(B, board_width=9, board_height=8, max_players=3, max_turns=20, seed=4, board_pool=16)
    one, two = mk(), mk()
    (oa, ia), (ob, ib) = one.reset(), two.reset()
    assert np.array_equal(oa, ob)
    rng = np.random.default_rng(2)
    for k in range(60):
        m = ia["valid_actions_mask"]
        acts = np.array([rng.choice(np.flatnonzero(r)) if r.any() else 0 for r in m], np.int64)
        if k % 5 == 2:
            acts[:6] = [int(np.flatnonzero(~r)[0]) for r in m[:6]]                       # refused: those envs sit the step out
        others = two.engine.agent_actions(4 + 1000 * k + 1, 0)                            # what step k's launch draws (vector_env.py: seed formula)
        a = one.step(acts)
        b = two.step(acts, other_actions=others)
        for x, y, name in zip(a[:4], b[:4], ("obs", "reward", "terminated", "truncated")):
            assert np.array_equal(x, y), (k, name)
        ia, ib = a[4], b[4]
        assert set(ia) == set(ib)
        for f in ia:
            assert np.array_equal(ia[f], ib[f]), (k, f)
    H.assert_states_equal(one.engine.game_state(), two.engine.game_state(), "supplied opponents")
    one.close(); two.close()


@pytest.mark.gpu
def test_single_env_facade_plays_an_opponent_agent(capsys):
    """GeneralsEnv(opponent_agent=...): the agent is handed the learner's proto GameState (as the reference does) and its
    Action is played for player 1 - checked against the oracle engine stepped with the same two moves."""
    import types
    import _oracle as O
    from generalsreinforcementlearning_amd import wire
    from generalsreinforcementlearning_amd.vec_engine import ACTION_DTYPE
    from generalsreinforcementlearning_amd.vector_env import GeneralsEnv
    W, Hh = 7, 6
    ora = O.OracleBatch(1, W, Hh, 2, fog=True)
    rng = np.random.default_rng(9)
    seen = {"states": 0, "moves": 0, "none": 0}

    class Opponent:                                        # picks among player 1's TRUE legal moves (it knows the oracle), or passes
        def select_action(self, state):
            assert type(state).__name__ == "GameState" and state.board.width == W and len(state.board.tiles) == W * Hh
            assert state.turn == int(ora.read_state()["turn"][0]) and [p.name for p in state.players] == ["RL_Agent", "Opponent"]
            seen["states"] += 1
            legal = np.flatnonzero(ora.engine(0).legal_mask(1))
            if len(legal) == 0 or rng.random() < 0.2:
                seen["none"] += 1
                self.last = None
                return None
            m = int(rng.choice(legal))
            t, d = m // 4, m % 4
            fx, fy = t % W, t // W
            dx, dy = [(0, -1), (1, 0), (0, 1), (-1, 0)][d]                   # Engine.GetLegalActionMask: up, right, down, left
            seen["moves"] += 1
            self.last = (fx, fy, fx + dx, fy + dy, bool(rng.random() < 0.3))
            NS = types.SimpleNamespace
            return NS(**{"from": NS(x=fx, y=fy), "to": NS(x=fx + dx, y=fy + dy), "half": self.last[4]})

    opp = Opponent()
    env = GeneralsEnv(board_width=W, board_height=Hh, max_players=2, fog_of_war=True, max_turns=60, opponent_agent=opp, seed=3)
    obs, info = env.reset()
    st = env._vec.engine.game_state()
    ora.reset(st["army"], st["owner"], st["type"], st["width"], st["height"], st["players"])
    H.assert_states_equal(env._vec.engine.game_state(), ora.read_state(), "after reset")
    for k in range(50):
        mask = info["valid_actions_mask"]
        full = np.flatnonzero(mask.reshape(-1, 5)[:, :4].reshape(-1))                      # full moves only: tile * 4 + d
        if len(full) == 0:
            break
        m = int(rng.choice(full))
        t, d = m // 4, m % 4
        fx, fy = t % W, t // W
        dx, dy = [(0, -1), (1, 0), (0, 1), (-1, 0)][d]
        obs, r, term, trunc, info = env.step(t * 5 + d)
        acts = np.zeros((1, 2), ACTION_DTYPE)
        acts[0, 0] = (fx, fy, fx + dx, fy + dy, 1, (0, 0, 0))
        if opp.last is not None:
            acts[0, 1] = opp.last[:4] + (1 | (2 if opp.last[4] else 0), (0, 0, 0))
        ora.step(acts)
        H.assert_states_equal(env._vec.engine.game_state(), ora.read_state(), f"turn {k}")
        if term or trunc:
            break
    assert seen["states"] >= 10 and seen["moves"] > 5 and seen["none"] > 0
    env.render_mode = "human"
    capsys.readouterr()
    env.render()                                            # the learner's view as text, like _print_board (:568-597)
    text = capsys.readouterr().out.splitlines()
    assert text[1].startswith("Turn ") and text[2] == "=" * (W * 4 + 1) and len(text) == Hh + 4 and all(len(r) == W * 4 + 1 for r in text[3:3 + Hh])
    env.close()
