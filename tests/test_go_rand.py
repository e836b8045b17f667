"""Go's math/rand, restated (oracle/generals_oracle.c "Go's math/rand"; table: scripts/gen_go_rand_cooked.py), and the
reference's map generator on it - which makes seed -> board parity with the Go engine checkable without a Go toolchain:

  * Seed(1) -> Intn(100) x 10 = 81 87 47 59 81 18 25 40 56 0, Go's well-known default sequence;
  * the generator script's derivation re-run here (jump-ahead == the committed table; both .inc files identical);
  * mapgen/generator_test.go's two seed-12345 assertions (22 mountains; 58 mountains + 20 cities + 4 spaced generals),
    the only places where the reference pins its RNG stream (tests/golden/reference_kats.json, kind mapgen_seeded);
  * the RNG-independent assertions of generator_test.go on Go-seeded boards (seeds 0..49 on 5x5 never fail, :27-46)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import _oracle as O
from test_oracle_golden import by_kind

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_go_rand_default_sequence():
    r = O.GoRand(1)
    assert [r.intn(100) for _ in range(10)] == [81, 87, 47, 59, 81, 18, 25, 40, 56, 0]
    # Int31n's two branches: powers of two mask, everything else rejects and reduces
    r = O.GoRand(42)
    assert all(0 <= r.intn(n) < n for n in (1, 2, 3, 7, 8, 20, 25, 1000, 1 << 20, (1 << 31) - 1) for _ in range(50))
    assert all(0 <= O.GoRand(s).int63() < 1 << 63 for s in (0, -5, 1 << 40, 2147483647, 2147483648))
    # Seed reduces mod 2^31 - 1 and maps 0 to 89482311 (rng.go Seed)
    assert O.GoRand(0).int63() == O.GoRand(2147483647).int63() == O.GoRand(89482311).int63()
    assert O.GoRand(5).int63() == O.GoRand(5 + 2147483647).int63() != O.GoRand(6).int63()


def test_cooked_table_is_what_the_script_derives():
    a = open(os.path.join(ROOT, "oracle", "go_rand_cooked.inc")).read()
    b = open(os.path.join(ROOT, "generalsreinforcementlearning_amd", "csrc", "go_rand_cooked.inc")).read()
    assert a == b
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import gen_go_rand_cooked as G
    cooked = G.advance(G.seed_vector(1, 20, 10), 7_800_000_000_000)
    words = [int(w.rstrip("ull,"), 16) for w in a.split() if w.startswith("0x")]
    assert words == cooked and len(words) == 607


@pytest.mark.parametrize("c", by_kind("mapgen_seeded"))
def test_reference_seeded_vectors(c):
    w, h, e = c["w"], c["h"], c["expect"]
    rc, army, owner, typ = O.mapgen_go(c["seed"], w, h, c["players"], c["cfg"])
    assert rc == 0
    assert int((typ == 3).sum()) == e["mountains"] and int((typ == 2).sum()) == e["cities"] and int((typ == 1).sum()) == e["generals"]
    assert (army[typ == 3] == 0).all() and (owner[typ == 3] == -1).all()                    # generator_test.go:75-79
    if e["cities"]:
        assert (army[typ == 2] == e["city_army"]).all() and (owner[typ == 2] == -1).all()   # :428-431
    if e["generals"]:
        g = np.flatnonzero(typ == 1)
        assert (army[g] == e["general_army"]).all() and sorted(owner[g]) == list(range(e["generals"]))   # :421-426
        for i in range(len(g)):
            for j in range(i + 1, len(g)):
                assert abs(g[i] % w - g[j] % w) + abs(g[i] // w - g[j] // w) >= e["min_spacing"]          # :446-455
    assert (army[(typ == 0)] == 0).all() and (owner[typ == 0] == -1).all()                  # :436-439


def test_default_config_boards_from_go_seeds():
    """DefaultMapConfig boards (what game.NewEngine builds from GameConfig.Rng): generator_test.go:27-46 - a 5x5 2-player
    board never fails, whatever the seed - and the default ratios on the BASELINE sizes."""
    for seed in range(50):
        rc, army, owner, typ = O.mapgen_go(seed, 5, 5, 2)
        assert rc == 0 and int((typ == 1).sum()) == 2
    for (w, h, p) in ((10, 10, 2), (15, 15, 2), (20, 20, 4)):
        for seed in (1, 2, 3, 12345):
            rc, army, owner, typ = O.mapgen_go(seed, w, h, p)
            assert rc == 0 and int((typ == 1).sum()) == p and int((typ == 2).sum()) == (w * h) // 20
            assert 0 < int((typ == 3).sum()) <= ((w * h) // 50) * max(3, w // 4)
    a = O.mapgen_go(7, 20, 20, 4)
    b = O.mapgen_go(7, 20, 20, 4)
    assert all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))
    c = O.mapgen_go(8, 20, 20, 4)
    assert not np.array_equal(a[3], c[3])


# ---- the device generator on Go's math/rand (gvec_reset_go_seeded) ---------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [[(20, 20, 4)], [(10, 10, 2)], [(5, 5, 2), (15, 15, 3), (20, 20, 4), (25, 25, 4), (32, 32, 8), (7, 30, 5)]],
                         ids=["20x20_p4", "10x10_p2", "mixed_padded"])
def test_device_boards_from_go_seeds_equal_the_oracle_s(sizes):
    """gvec_reset_go_seeded: board i is what the oracle's restatement of Go's math/rand + generator builds from seeds[i] -
    which the reference's seed-12345 vectors pin - and the engines then start identically (performInitialSetup)."""
    import generalsreinforcementlearning_amd as g
    import _harness as H
    B = 300
    per = [sizes[i % len(sizes)] for i in range(B)]
    mw, mh, mp = max(s[0] for s in per), max(s[1] for s in per), max(s[2] for s in per)
    rng = np.random.default_rng(5)
    seeds = np.concatenate([[12345, 0, 1, -7, 2147483647, 2147483648, 1 << 40], rng.integers(-2 ** 62, 2 ** 62, B - 7)]).astype(np.int64)
    ws, hs, ps = (np.array([s[k] for s in per], np.int32) for k in range(3))
    eng = g.VecEngine(B, mw, mh, mp)
    eng.reset_go_seeded(seeds, ws, hs, ps)
    st = eng.game_state()
    army = np.zeros((B, mw * mh), np.int32)
    owner = np.full((B, mw * mh), -1, np.int8)
    typ = np.zeros((B, mw * mh), np.uint8)
    for i, (w, h, p) in enumerate(per):
        rc, a, o, t = O.mapgen_go(int(seeds[i]), w, h, p, stride=mw * mh)
        assert rc == 0
        army[i], owner[i], typ[i] = a, o, t
    for f, want in (("army", army), ("owner", owner), ("type", typ)):
        assert np.array_equal(st[f], want), (f, np.flatnonzero((st[f] != want).any(1))[:8])
    ora = O.OracleBatch(B, mw, mh, mp)
    ora.reset(army, owner, typ, ws, hs, ps)
    H.assert_states_equal(st, ora.read_state(), "after gvec_reset_go_seeded")
    H.run_lockstep(eng, ora, 30, seed=2, invalid_permille=5, check_every=10, ctx="go-seeded boards")
    # sharded handles hand every shard its slice of the seeds
    many = g.VecEngine(B, mw, mh, mp, devices=[0, 0, 0])
    many.reset_go_seeded(seeds, ws, hs, ps)
    sm = many.game_state()
    for f in ("army", "owner", "type", "visible", "listed"):
        assert np.array_equal(sm[f], st[f]), f
    many.close()
    eng.close()


# ---- engine_test.go on the boards the Go tests themselves build (NewEngine with Rng = rand.NewSource(12345)) ---------
def _new_engine_checks(st, P):
    """TestNewEngine (game/engine_test.go:24-63) on a state dict of one env."""
    assert not st["done"][0] and st["turn"][0] == 0                       # :46-47
    for i in range(P):
        assert st["alive"][0, i]                                          # :52
        gi = int(st["general_idx"][0, i])
        assert gi != -1 and st["owner"][0, gi] == i and st["type"][0, gi] == 1   # :53-58
        assert st["army_count"][0, i] >= 1                                # :60
    assert int((st["type"][0] == 1).sum()) == P                           # :62


def test_new_engine_on_the_go_tests_own_board_oracle():
    rc, army, owner, typ = O.mapgen_go(12345, 8, 8, 2)
    assert rc == 0
    ora = O.OracleBatch(1, 8, 8, 2)
    ora.reset(army[None], owner[None], typ[None], [8], [8], [2])
    _new_engine_checks(ora.read_state(), 2)
    # TestEngine_Step_BasicTurn (:65-86): NewEngine(5x5, 1 player, seed 12345), one Step without actions
    rc, army, owner, typ = O.mapgen_go(12345, 5, 5, 1)
    one = O.OracleBatch(1, 5, 5, 1)
    one.reset(army[None], owner[None], typ[None], [5], [5], [1])
    s0 = one.read_state()
    acts = np.zeros((1, 1), O.ACTION_DTYPE)
    assert one.step(acts)[0] == 0
    s1 = one.read_state()
    assert s1["turn"][0] == s0["turn"][0] + 1 and s1["army_count"][0, 0] == s0["army_count"][0, 0] + 1 and not s1["done"][0]   # :82-85


@pytest.mark.gpu
def test_new_engine_on_the_go_tests_own_board_hip():
    import generalsreinforcementlearning_amd as g
    eng = g.VecEngine(1, 8, 8, 2)
    eng.reset_go_seeded([12345])
    _new_engine_checks(eng.game_state(), 2)
    one = g.VecEngine(1, 5, 5, 1)
    one.reset_go_seeded([12345])
    s0 = one.game_state()
    assert one.step(g.make_actions(1, 1))[0] == 0
    s1 = one.game_state()
    assert s1["turn"][0] == s0["turn"][0] + 1 and s1["army_count"][0, 0] == s0["army_count"][0, 0] + 1 and not s1["done"][0]
