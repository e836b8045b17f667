"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the
reference's golden vectors.  Bar: bit-exact (all integer / byte / index work)."""
import json
import os

import numpy as np
import pytest

import _harness as H
import _oracle as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_kats.json")) as f:
    CASES = json.load(f)["cases"]


def by_kind(kind):
    return [pytest.param(c, id=c["name"]) for c in CASES if c["kind"] == kind]


@pytest.fixture(scope="module")
def g():
    import generalsreinforcementlearning_amd as g
    g.load()
    return g


def test_wave_primitives_selftest(g):
    assert g.lib().gvec_selftest(0) == 0, g.lib().gvec_last_error()


# ------------------------------------------------------------------ golden vectors on the HIP path
def _raw_engine(g, w, h, players, tiles, fog=True):
    """createTestEngineForActionMask-style engine (action_mask_test.go:16-55): players alive,
    OwnedTiles empty, Turn 0, nothing initialised."""
    # production (0,0,0): these vectors pin core-level moves / masks, not the production phase
    eng = g.VecEngine(1, w, h, players, fog_of_war=fog, production=(0, 0, 0))
    army, owner, typ = O.planes_from_tiles(w, h, tiles)
    eng.reset(army[None], owner[None], typ[None])
    eng.write_state({"listed": np.full((1, w * h), -1, np.int8), "alive": np.ones((1, players), np.uint8),
                     "done": np.zeros(1, np.uint8), "general_idx": np.full((1, players), -1, np.int32),
                     "visible": np.zeros((1, w * h), np.uint8)})
    return eng


def _check_tiles(st, w, expect_tiles):
    for t in expect_tiles:
        i = t["y"] * w + t["x"]
        for k in ("owner", "army", "type"):
            if k in t:
                assert st[k][0, i] == t[k], (t, k, st[k][0, i])


@pytest.mark.parametrize("c", by_kind("apply_move"))
def test_golden_apply_move(g, c):
    players = 4
    eng = _raw_engine(g, c["w"], c["h"], players, c["tiles"])
    a = c["action"]
    acts = g.make_actions(1, players, [(0, a["player"], a["from"][0], a["from"][1], a["to"][0], a["to"][1], a["move_all"])])
    before = eng.game_state()
    err = eng.step(acts)
    e = c["expect"]
    assert err[0] == e["err"]
    st = eng.game_state()
    _check_tiles(st, c["w"], e.get("tiles", []))
    if e["err"]:
        for f in ("army", "owner"):
            assert np.array_equal(st[f], before[f])
    for x, y in e.get("changed", []):
        assert st["changed"][0, y * c["w"] + x] == 1
    if "capture" in e or e.get("captured"):
        t = a["to"][1] * c["w"] + a["to"][0]
        assert st["vis_changed"][0, t] == 1 and st["owner"][0, t] == a["player"]
    elif "captured" in e:
        assert st["vis_changed"].sum() == 0
    for victim, new_owner in e.get("eliminations", []):
        assert st["alive"][0, victim] == 0


@pytest.mark.parametrize("c", by_kind("validate"))
def test_golden_validate(g, c):
    players = 4
    eng = _raw_engine(g, c["w"], c["h"], players, c["tiles"])
    a = c["action"]
    acts = g.make_actions(1, players, [(0, c["player"], a["from"][0], a["from"][1], a["to"][0], a["to"][1], a["move_all"])])
    assert eng.step(acts)[0] == c["expect"]["err"]


class _HipEngineAdapter:
    """Gives one env of a VecEngine the surface test_oracle_golden.run_engine_script expects."""

    def __init__(self, g, c):
        self.g, self.w, self.h, self.p = g, c["w"], c["h"], c["players"]
        self.eng = g.VecEngine(1, self.w, self.h, self.p)
        army, owner, typ = O.planes_from_tiles(self.w, self.h, c["tiles"])
        self.eng.reset(army[None], owner[None], typ[None])

    def st(self, *f):
        return self.eng.game_state(fields=f)

    def step(self, moves):
        acts = self.g.make_actions(1, self.p, [(0, m[0], m[1], m[2], m[3], m[4], bool(m[5])) for m in moves])
        return int(self.eng.step(acts)[0])

    turn = property(lambda s: int(s.st("turn")["turn"][0]))
    game_over = property(lambda s: bool(s.st("done")["done"][0]))
    winner = property(lambda s: int(s.st("winner")["winner"][0]))

    def alive(self, p):
        return bool(self.st("alive")["alive"][0, p])

    def army_count(self, p):
        return int(self.st("army_count")["army_count"][0, p])

    def general_idx(self, p):
        return int(self.st("general_idx")["general_idx"][0, p])


@pytest.mark.parametrize("c", by_kind("board_idx"))
def test_golden_board_idx(g, c):
    """core/board_test.go TestBoard_Idx through the ABI: a gvec_step move FROM (x, y) is accepted exactly when the player's
    army lies at the tile index the reference expects for (x, y), and is ErrNotOwned for every other index."""
    w, h = c["w"], c["h"]
    n = w * h
    for x, y, idx in c["xy_idx"]:
        eng = g.VecEngine(n, w, h, 2, fog_of_war=False)          # env `at`: the army sits at index `at`
        army, owner, typ = np.zeros((n, n), np.int32), np.full((n, n), -1, np.int8), np.zeros((n, n), np.uint8)
        for at in range(n):
            army[at, at], owner[at, at], typ[at, at] = 5, 0, 1      # both players hold a general: the game is live
            far = n - 1 if at != n - 1 else 0
            army[at, far], owner[at, far], typ[at, far] = 3, 1, 1
        eng.reset(army, owner, typ)
        tx, ty = (x + 1, y) if x + 1 < w else (x - 1, y)
        err = eng.step(g.make_actions(n, 2, [(at, 0, x, y, tx, ty, True) for at in range(n)]))
        want = np.full(n, 3, np.int32)
        want[idx] = 0
        assert np.array_equal(err, want), (x, y, idx, err)


@pytest.mark.parametrize("c", by_kind("board_xy"))
def test_golden_board_xy(g, c):
    """core/board_test.go TestBoard_XY as the fog update uses it: what a player sees around its only tile, at index idx, is
    the 3x3 centred on the reference's (x, y)."""
    from test_oracle_golden import fog_square, xy_board
    w, h = c["w"], c["h"]
    for idx, x, y in c["idx_xy"]:
        army, owner, typ = O.planes_from_tiles(w, h, xy_board(c, idx))
        eng = g.VecEngine(1, w, h, 2, fog_of_war=True)
        eng.reset(army[None], owner[None], typ[None])
        vis, _ = eng.compute_player_visibility(0)
        assert sorted(np.flatnonzero(vis[0])) == fog_square(x, y, w, h), (idx, x, y)


@pytest.mark.parametrize("c", [p for p in by_kind("engine") if "production_turn" not in p.id])
def test_golden_engine(g, c):
    a = _HipEngineAdapter(g, c)
    remembered = {}
    for s in c["script"]:
        if s["op"] == "poke":
            upd = {}
            if "tiles" in s:
                st = a.st("army", "owner", "type")
                for t in s["tiles"]:
                    i = t["y"] * a.w + t["x"]
                    for k in ("owner", "army", "type"):
                        if k in t:
                            st[k][0, i] = t[k]
                upd.update(st)
            if "turn" in s:
                upd["turn"] = np.array([s["turn"]], np.int32)
            if "game_over" in s:
                upd["done"] = np.array([int(s["game_over"])], np.uint8)
            if "alive" in s:
                al = a.st("alive")["alive"]
                for p, v in s["alive"].items():
                    al[0, int(p)] = int(v)
                upd["alive"] = al
            if "general_idx" in s:
                gi = a.st("general_idx")["general_idx"]
                for p, v in s["general_idx"].items():
                    gi[0, int(p)] = v
                upd["general_idx"] = gi
            a.eng.write_state(upd)
        elif s["op"] == "remember_army_count":
            remembered[s["player"]] = a.army_count(s["player"])
        elif s["op"] == "step":
            moves = [(m["player"], m["from"][0], m["from"][1], m["to"][0], m["to"][1], int(m["move_all"])) for m in s["actions"]]
            assert a.step(moves) == s["expect_err"]
        elif s["op"] == "expect":
            if "turn" in s:
                assert a.turn == s["turn"]
            if "game_over" in s:
                assert a.game_over == s["game_over"]
            if "winner" in s:
                assert a.winner == s["winner"]
            for p, v in s.get("alive", {}).items():
                assert a.alive(int(p)) == v
            for p, v in s.get("general_idx", {}).items():
                assert a.general_idx(int(p)) == v
            for p, d in s.get("army_count_delta", {}).items():
                assert a.army_count(int(p)) == remembered[int(p)] + d
            _check_tiles(a.eng.game_state(), a.w, s.get("tiles", []))


def test_golden_production_turn_25_vs_24(g):
    """game/engine_test.go:104-182 calls processTurnProduction directly at Turn 25 and 24; through
    Step the same production runs with Turn = previous + 1, so poke Turn 24 / 23 and step once."""
    c = next(x for x in CASES if x["name"] == "engine/production_turn_25_vs_24")
    for start_turn, expect in ((24, [3, 6, 3]), (23, [3, 6, 2])):
        a = _HipEngineAdapter(g, c)
        a.eng.write_state({"turn": np.array([start_turn], np.int32)})
        assert a.step([]) == 0
        st = a.eng.game_state()
        got = [int(st["army"][0, 2 * 5 + 2]), int(st["army"][0, 0]), int(st["army"][0, 1])]
        assert got == expect, (start_turn, got)


@pytest.mark.parametrize("c", by_kind("legal_mask"))
def test_golden_legal_mask(g, c):
    from test_oracle_golden import check_mask_expect
    w, h, P = c["w"], c["h"], c["players"]
    eng = _raw_engine(g, w, h, P, c["tiles"])
    listed = np.full((1, w * h), -1, np.int8)
    for p, tiles in c["owned"].items():
        listed[0, tiles] = int(p)
    alive = np.ones((1, P), np.uint8)
    for p, v in c["alive"].items():
        alive[0, int(p)] = int(v)
    eng.write_state({"listed": listed, "alive": alive})
    m = eng.get_legal_action_mask(0, c["query_player"])
    check_mask_expect(m.astype(np.uint8), c["expect"])


# ------------------------------------------------------------------ randomised lock-step parity
CONFIGS = [
    # name, B, (w, h, p) pattern, fog, turns, invalid_permille
    ("10x10_p2_fog_off", 512, [(10, 10, 2)], False, 300, 0),
    ("15x15_p2_fog_on", 256, [(15, 15, 2)], True, 300, 5),
    ("20x20_p4_fog_on", 192, [(20, 20, 4)], True, 300, 5),
    ("mixed_padded", 192, [(10, 10, 2), (15, 15, 3), (20, 20, 4)], True, 250, 10),
    ("tiny_boards", 64, [(3, 3, 2), (5, 5, 2), (8, 8, 3), (7, 5, 2)], True, 120, 20),
    ("wide_25x25_p8", 32, [(25, 25, 8), (32, 32, 5), (32, 17, 6)], True, 150, 5),
    # the remaining register layouts of the step kernel (gvec_packed.hpp): <4,10> two players per 32-lane
    # row pair, <8,7> two registers of four 16-lane rows, <8,10> four registers of two rows
    ("25x25_p4", 48, [(25, 25, 4), (22, 24, 3)], True, 150, 5),
    ("20x20_p8", 48, [(20, 20, 8), (16, 21, 6), (21, 21, 5)], True, 150, 5),
    ("25x25_p8", 32, [(25, 25, 8), (24, 26, 7)], True, 120, 5),
]


@pytest.mark.parametrize("name,B,pattern,fog,turns,inv", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_lockstep_vs_oracle(g, name, B, pattern, fog, turns, inv):
    sizes = [pattern[i % len(pattern)] for i in range(B)]
    mw, mh, mp = max(s[0] for s in sizes), max(s[1] for s in sizes), max(s[2] for s in sizes)
    army, owner, typ, w, h, p = H.gen_boards(1234, sizes, mw, mh)
    eng = g.VecEngine(B, mw, mh, mp, fog_of_war=fog)
    ora = O.OracleBatch(B, mw, mh, mp, fog=fog)
    eng.reset(army, owner, typ, w, h, p)
    ora.reset(army, owner, typ, w, h, p)
    # start turns offset per env (growth-interval divergence stress, BASELINE config 5)
    t0 = (np.arange(B) % 25).astype(np.int32)
    eng.write_state({"turn": t0})
    ora.write_state({"turn": t0})
    H.assert_states_equal(eng.game_state(), ora.read_state(), f"{name} after reset")
    assert np.array_equal(eng.legal_action_mask_bits(), ora.legal_mask())
    H.run_lockstep(eng, ora, turns, seed=7, invalid_permille=inv, check_every=1, ctx=name)
    # something must actually have happened
    st = ora.read_state()
    assert st["changed"].sum() > 0 and (st["owner"] >= 0).sum() > B * mp


def test_player_visibility_matches_oracle(g):
    B, sizes = 64, [(12, 9, 3)] * 64
    army, owner, typ, w, h, p = H.gen_boards(5, sizes, 12, 9)
    for fog in (True, False):
        eng = g.VecEngine(B, 12, 9, 3, fog_of_war=fog)
        ora = O.OracleBatch(B, 12, 9, 3, fog=fog)
        eng.reset(army, owner, typ, w, h, p)
        ora.reset(army, owner, typ, w, h, p)
        H.run_lockstep(eng, ora, 60, seed=3, check_every=20, want_mask=False, ctx="vis")
        for player in (0, 1, 2, 5, -1):
            hv, hf = eng.compute_player_visibility(player)
            for e in range(B):
                ov, of = ora.engine(e).player_visibility(player)
                assert np.array_equal(hv[e, :108], ov.astype(bool)) and np.array_equal(hf[e, :108], of.astype(bool)), (fog, player, e)


# ------------------------------------------------------------------ synthetic-input generators
def test_device_agent_matches_oracle_agent(g):
    B, sizes = 256, [(15, 15, 2), (20, 20, 4), (10, 10, 3)]
    sizes = [sizes[i % 3] for i in range(B)]
    army, owner, typ, w, h, p = H.gen_boards(77, sizes, 20, 20)
    eng = g.VecEngine(B, 20, 20, 4)
    ora = O.OracleBatch(B, 20, 20, 4)
    eng.reset(army, owner, typ, w, h, p)
    ora.reset(army, owner, typ, w, h, p)
    for k in range(120):
        inv = 30 if k % 3 == 0 else 0
        oa = ora.agent_actions(99, inv)
        ha = eng.agent_actions(99, inv)
        assert np.array_equal(ha.view(np.uint64), oa.view(np.uint64)), f"turn {k}: agent actions differ in envs {np.unique(np.argwhere(ha.view(np.uint64) != oa.view(np.uint64))[:, 0])[:8]}"
        assert np.array_equal(eng.step(oa), ora.step(oa))


def test_device_mapgen_matches_oracle_mapgen(g):
    B = 300
    sizes = [[(10, 10, 2), (15, 15, 2), (20, 20, 4), (8, 8, 3), (25, 20, 5)][i % 5] for i in range(B)]
    army, owner, typ, w, h, p = H.gen_boards(4242, sizes, 25, 20)
    eng = g.VecEngine(B, 25, 20, 5)
    eng.reset_generated(4242, w, h, p)
    st = eng.game_state()
    assert np.array_equal(st["army"], army) and np.array_equal(st["owner"], owner) and np.array_equal(st["type"], typ)
    ora = O.OracleBatch(B, 25, 20, 5)
    ora.reset(army, owner, typ, w, h, p)
    H.assert_states_equal(st, ora.read_state(), "generated reset")
    # ratios of mapgen/generator.go:25-47 (distribution-level parity with the Go generator)
    for i in (2, 7):
        n = w[i] * h[i]
        assert (typ[i] == 2).sum() == n // 20 and (typ[i] == 1).sum() == p[i]
        assert 0 < (typ[i] == 3).sum() <= (n // 50) * max(3, w[i] // 4)


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "per_turn_launch"])
def test_rollout_matches_oracle_rollout(g, fused):
    B, K = 384, 160
    sizes = [[(20, 20, 4), (15, 15, 2), (10, 10, 2)][i % 3] for i in range(B)]
    army, owner, typ, w, h, p = H.gen_boards(11, sizes, 20, 20)
    eng = g.VecEngine(B, 20, 20, 4)
    ora = O.OracleBatch(B, 20, 20, 4)
    eng.reset(army, owner, typ, w, h, p)
    ora.reset(army, owner, typ, w, h, p)
    stats = eng.rollout(K, seed=2024, invalid_permille=10, fused=fused)
    steps = ora.rollout(K, 2024, 10)
    H.assert_states_equal(eng.game_state(), ora.read_state(), "rollout")
    assert stats["env_steps"] == steps
    assert stats["aborted_turns"] > 0  # the H5 path was exercised
    assert np.array_equal(eng.legal_action_mask_bits(), ora.legal_mask())


@pytest.mark.parametrize("sizes", [[(25, 25, 4), (22, 24, 3)], [(32, 32, 8), (25, 25, 8), (32, 17, 6)], [(20, 20, 8), (21, 21, 5)],
                                   [(25, 25, 2), (32, 32, 2)]],
                         ids=["25x25_p4", "32x32_p8", "20x20_p8", "large_p2"])
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "per_turn_launch"])
def test_rollout_matches_oracle_on_every_register_layout(g, fused, sizes):
    """The fused and per-turn rollouts of the larger kernel variants (packed layouts of gvec_packed.hpp,
    two mask words per player) against the oracle's rollout, auto-reset included."""
    B, K = 96, 120
    per_env = [sizes[i % len(sizes)] for i in range(B)]
    mw, mh, mp = max(s[0] for s in per_env), max(s[1] for s in per_env), max(s[2] for s in per_env)
    army, owner, typ, w, h, p = H.gen_boards(77, per_env, mw, mh)
    eng = g.VecEngine(B, mw, mh, mp, auto_reset=True)
    ora = O.OracleBatch(B, mw, mh, mp)
    eng.reset(army, owner, typ, w, h, p)
    ora.reset(army, owner, typ, w, h, p)
    eng.build_board_pool(13, 99)
    ora.set_pool(13, 99)
    stats = eng.rollout(K, seed=4242, invalid_permille=10, fused=fused)
    steps = ora.rollout(K, 4242, 10)
    H.assert_states_equal(eng.game_state(), ora.read_state(), f"rollout fused={fused}")
    assert stats["env_steps"] == steps
    assert np.array_equal(eng.legal_action_mask_bits(), ora.legal_mask())


def test_auto_reset_matches_oracle(g):
    # tiny boards so games actually finish; every finished env is re-dealt from the pool
    B, K, pool = 256, 400, 37
    sizes = [(6, 6, 2)] * B
    army, owner, typ, w, h, p = H.gen_boards(3, sizes, 6, 6)
    eng = g.VecEngine(B, 6, 6, 2, auto_reset=True)
    ora = O.OracleBatch(B, 6, 6, 2)
    eng.reset(army, owner, typ, w, h, p)
    ora.reset(army, owner, typ, w, h, p)
    eng.build_board_pool(pool, 555)
    ora.set_pool(pool, 555)
    stats = eng.rollout(K, seed=8, fused=True)
    steps = ora.rollout(K, 8, 0)
    assert stats["games_finished"] > 0, "no game finished: the auto-reset path was not exercised"
    assert stats["env_steps"] == steps
    H.assert_states_equal(eng.game_state(), ora.read_state(), "auto-reset rollout")
    # and through the per-turn API
    for k in range(50):
        acts = ora.agent_actions(9)
        assert np.array_equal(eng.step(acts), ora.step(acts))
    H.assert_states_equal(eng.game_state(), ora.read_state(), "auto-reset steps")


def test_game_over_env_is_frozen(g):
    eng = g.VecEngine(2, 5, 5, 2)
    tiles = [dict(x=0, y=0, owner=0, army=20, type=0), dict(x=0, y=1, owner=1, army=1, type=1), dict(x=4, y=4, owner=0, army=2, type=1)]
    army, owner, typ = O.planes_from_tiles(5, 5, tiles)
    eng.reset(np.stack([army, army]), np.stack([owner, owner]), np.stack([typ, typ]))
    acts = g.make_actions(2, 2, [(0, 0, 0, 0, 0, 1, True)])
    assert list(eng.step(acts)) == [0, 0]
    st1 = eng.game_state()
    assert list(st1["done"]) == [1, 0] and st1["winner"][0] == 0
    assert list(eng.step(g.make_actions(2, 2))) == [5, 0]  # ErrGameOver for the finished env only
    st2 = eng.game_state()
    for f in H.TILE_FIELDS + ("turn",):
        assert np.array_equal(st1[f][0], st2[f][0]), f
    assert st2["turn"][1] == st1["turn"][1] + 1


def test_api_misuse_is_reported(g):
    eng = g.VecEngine(2, 5, 5, 2)
    army, owner, typ = O.planes_from_tiles(5, 5, [dict(x=0, y=0, owner=3, army=1, type=1)])
    with pytest.raises(g.GvecError) as ei:
        eng.reset(army[None], owner[None], typ[None])
    assert ei.value.code == -5  # GVEC_E_BOARD: owner >= players
    with pytest.raises(g.GvecError):
        eng.game_state(1, 5)
    with pytest.raises(g.GvecError):
        g.VecEngine(1, 40, 5, 2)


def test_record_export_import_roundtrip(g):
    import torch
    B = 64
    sizes = [(15, 15, 2)] * B
    army, owner, typ, w, h, p = H.gen_boards(21, sizes, 15, 15)
    a = g.VecEngine(B, 15, 15, 2)
    b = g.VecEngine(B, 15, 15, 2)
    a.reset(army, owner, typ, w, h, p)
    a.rollout(40, seed=1)
    buf = torch.empty(B * a.state_bytes_per_env(), dtype=torch.uint8, device="cuda")
    a.export_records(buf.data_ptr())
    a.synchronize()
    b.import_records(buf.data_ptr())
    b.synchronize()
    sa, sb = a.game_state(), b.game_state()
    for f in sa:
        assert np.array_equal(sa[f], sb[f]), f
    assert np.array_equal(a.legal_action_mask_bits(), b.legal_action_mask_bits())


# ------------------------------------------------------------------ full BASELINE sizes
@pytest.mark.parametrize("B,w,h,p,fog", [(65536, 15, 15, 2, True), (262144, 20, 20, 4, True), (4096, 10, 10, 2, False)],
                         ids=["cfg3_65536x15x15", "cfg4_262144x20x20", "cfg2_4096x10x10"])
def test_full_size_subset_and_invariants(g, B, w, h, p, fog):
    """At BASELINE.json's full sizes: (1) envs are independent and keyed by env id, so the first
    2,048 envs of the big batch must equal an oracle run of just those envs; (2) fused K-turn
    rollouts and K single-turn launches give identical state (checksum over all envs);
    (3) structural invariants over every env."""
    K, sub = 60, 2048
    eng = g.VecEngine(B, w, h, p, fog_of_war=fog)
    eng.reset_generated(31337)
    first = eng.game_state(0, sub)
    ora = O.OracleBatch(sub, w, h, p, fog=fog)
    ora.reset(first["army"], first["owner"], first["type"], first["width"], first["height"], first["players"])
    stats = eng.rollout(K, seed=5, invalid_permille=5, fused=True)
    ora.rollout(K, 5, 5, threads=8)
    H.assert_states_equal(eng.game_state(0, sub), ora.read_state(), "full-size subset")
    assert stats["env_steps"] == B * K - 0 or stats["games_finished"] > 0
    eng2 = g.VecEngine(B, w, h, p, fog_of_war=fog)
    eng2.reset_generated(31337)
    eng2.rollout(K, seed=5, invalid_permille=5, fused=False, want_stats=False)
    for lo in range(0, B, 32768):
        n = min(32768, B - lo)
        s1, s2 = eng.game_state(lo, n), eng2.game_state(lo, n)
        for f in s1:
            assert np.array_equal(s1[f], s2[f]), (f, lo)
        N = w * h
        assert (s1["army"] >= 0).all()
        assert ((s1["owner"] >= -1) & (s1["owner"] < p)).all()
        assert (s1["type"][:, :N] == 3).sum() > 0 and not ((s1["type"] == 3) & (s1["owner"] >= 0)).any()  # mountains stay neutral
        assert ((s1["listed"] >= -1) & (s1["listed"] < p)).all()
        assert ((s1["turn"] == K) | (s1["done"] == 1)).all()
        tc = np.stack([(s1["listed"] == q).sum(1) for q in range(p)], 1)
        assert np.array_equal(tc, s1["tile_count"])
        ac = np.stack([np.where(s1["listed"] == q, s1["army"], 0).sum(1) for q in range(p)], 1)
        # ArmyCount is as of the last stats pass: exact unless the env's last turn was aborted (H5)
        assert (ac == s1["army_count"]).all(1).mean() > 0.9


# ------------------------------------------------------------------ every compiled variant of the kernels
@pytest.mark.parametrize("maxp", [2, 4, 8])
@pytest.mark.parametrize("slots,parity", sorted(H.VARIANT_DIMS), ids=H.VARIANT_IDS)
def test_every_kernel_variant_against_the_oracle(g, maxp, slots, parity):
    """Per-turn lock-step (host actions, masks, err codes every turn), then the device agent's per-turn and fused rollouts
    with auto-reset, on the board limits that select this <players, slots, parity> instantiation (tests/_harness.py)."""
    B = 24
    mw, mh, sizes = H.variant_batch(maxp, slots, parity, B)
    army, owner, typ, w, h, p = H.gen_boards(900 + slots, sizes, mw, mh)
    eng = g.VecEngine(B, mw, mh, maxp, fog_of_war=True, auto_reset=True)
    ora = O.OracleBatch(B, mw, mh, maxp, fog=True)
    eng.reset(army, owner, typ, w, h, p)
    ora.reset(army, owner, typ, w, h, p)
    t0 = (np.arange(B) % 25).astype(np.int32)
    eng.write_state({"turn": t0})
    ora.write_state({"turn": t0})
    ctx = f"variant <{maxp},{slots},{parity}>"
    H.run_lockstep(eng, ora, 40, seed=11, invalid_permille=10, check_every=1, ctx=ctx)
    pw, ph, pp = w[:5], h[:5], p[:5]                             # pool boards of the batch's own (ragged) shapes
    eng.build_board_pool(5, 31, pw, ph, pp)
    ora.set_pool(5, 31, pw, ph, pp)
    for fused in (False, True):
        stats = eng.rollout(30, seed=77 + fused, invalid_permille=10, fused=fused)
        steps = ora.rollout(30, 77 + fused, 10)
        H.assert_states_equal(eng.game_state(), ora.read_state(), f"{ctx} rollout fused={fused}")
        assert stats["env_steps"] == steps
        assert np.array_equal(eng.legal_action_mask_bits(), ora.legal_mask())
