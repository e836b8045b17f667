"""Pins the CPU oracle against the known-answer vectors the reference's own Go
tests hold (tests/golden/reference_kats.json, transcribed in make_reference_kats.py)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import _oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_kats.json")) as f:
    CASES = json.load(f)["cases"]


def by_kind(kind):
    return [pytest.param(c, id=c["name"]) for c in CASES if c["kind"] == kind]


def make_board(c):
    L = O.lib()
    b = L.ora_board_new(c["w"], c["h"])
    for t in c["tiles"]:
        tl = b.contents.t[t["y"] * c["w"] + t["x"]]
        tl.owner, tl.army, tl.type = t.get("owner", -1), t.get("army", 0), t.get("type", 0)
    return b


def mk_move(a):
    return O.Move(a["player"], a["from"][0], a["from"][1], a["to"][0], a["to"][1], int(a["move_all"]))


def check_tiles(get_tile, w, expect_tiles):
    for t in expect_tiles:
        tl = get_tile(t["x"], t["y"])
        for k in ("owner", "army", "type"):
            if k in t:
                assert getattr(tl, k) == t[k], (t, k, getattr(tl, k))


def test_regenerated_json_is_current():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(HERE, "golden", "make_reference_kats.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    assert json.loads(json.dumps(mk.cases)) == CASES


@pytest.mark.parametrize("c", by_kind("apply_move"))
def test_apply_move(c):
    L = O.lib()
    b = make_board(c)
    mv = mk_move(c["action"])
    changed = np.zeros(c["w"] * c["h"], np.uint8)
    cap = O.Capture()
    captured = C.c_int32(0)
    err = L.ora_apply_move(b, C.byref(mv), changed.ctypes.data_as(O.u8p), C.byref(cap), C.byref(captured))
    e = c["expect"]
    if e.get("err_nonzero"):
        assert err != 0
    assert err == e["err"]
    if "capture" in e:
        assert captured.value == 1
        x = e["capture"]
        assert (cap.x, cap.y, cap.tile_type, cap.capturing_player, cap.previous_owner, cap.previous_army) == \
            (x["x"], x["y"], x["tile_type"], x["capturer"], x["prev_owner"], x["prev_army"])
    if "captured" in e:
        assert bool(captured.value) == e["captured"]
    check_tiles(lambda x, y: b.contents.t[y * c["w"] + x], c["w"], e.get("tiles", []))
    for x, y in e.get("changed", []):
        assert changed[y * c["w"] + x] == 1
    if "eliminations" in e:
        out = (O.Elimination * 4)()
        n = L.ora_process_captures(C.byref(cap), 1, out)
        assert [[out[i].eliminated, out[i].new_owner] for i in range(n)] == e["eliminations"]
    L.ora_board_free(b)


@pytest.mark.parametrize("c", by_kind("validate"))
def test_validate(c):
    L = O.lib()
    b = make_board(c)
    mv = mk_move(c["action"])
    assert L.ora_validate(b, C.byref(mv), c["player"]) == c["expect"]["err"]
    L.ora_board_free(b)


@pytest.mark.parametrize("c", by_kind("process_captures"))
def test_process_captures(c):
    L = O.lib()
    caps = (O.Capture * max(1, len(c["captures"])))()
    for i, x in enumerate(c["captures"]):
        caps[i] = O.Capture(x["x"], x["y"], x["tile_type"], x["capturer"], x["prev_owner"], x["prev_army"])
    out = (O.Elimination * 8)()
    n = L.ora_process_captures(caps, len(c["captures"]), out)
    assert [[out[i].eliminated, out[i].new_owner] for i in range(n)] == c["expect"]["eliminations"]


@pytest.mark.parametrize("c", by_kind("bitfield"))
def test_bitfield(c):
    L = O.lib()
    t = O.Tile()
    for op in c["ops"]:
        if op[0] == "set":
            L.ora_tile_set_visible(C.byref(t), op[1], op[2])
        elif op[0] == "raw":
            t.visible = op[1]
        else:
            assert L.ora_tile_is_visible_to(C.byref(t), op[1]) == op[2], op


def run_engine_script(c, eng):
    """Shared interpreter for 'engine' cases; `eng` is any object with the OracleEngine surface."""
    remembered = {}
    for s in c["script"]:
        op = s["op"]
        if op == "poke":
            eng_poke(eng, s)
        elif op == "remember_army_count":
            remembered[s["player"]] = eng.army_count(s["player"])
        elif op == "step":
            moves = [(a["player"], a["from"][0], a["from"][1], a["to"][0], a["to"][1], int(a["move_all"])) for a in s["actions"]]
            assert eng.step(moves) == s["expect_err"]
        elif op == "production":
            eng.L.ora_engine_process_production(eng.e)
        elif op == "expect":
            if "turn" in s:
                assert eng.turn == s["turn"]
            if "game_over" in s:
                assert eng.game_over == s["game_over"]
            if "winner" in s:
                assert eng.winner == s["winner"]
            for p, v in s.get("alive", {}).items():
                assert eng.alive(int(p)) == v
            for p, v in s.get("general_idx", {}).items():
                assert eng.general_idx(int(p)) == v
            for p, d in s.get("army_count_delta", {}).items():
                assert eng.army_count(int(p)) == remembered[int(p)] + d
            check_tiles(eng.tile, eng.w, s.get("tiles", []))


def eng_poke(eng, s):
    L = eng.L
    for t in s.get("tiles", []):
        tl = eng.tile(t["x"], t["y"])
        for k in ("owner", "army", "type"):
            if k in t:
                setattr(tl, k, t[k])
    if "turn" in s:
        L.ora_engine_set_turn(eng.e, s["turn"])
    if "game_over" in s:
        L.ora_engine_set_game_over(eng.e, int(s["game_over"]))
    for p, v in s.get("alive", {}).items():
        L.ora_player_set_alive(eng.e, int(p), int(v))
    for p, v in s.get("general_idx", {}).items():
        L.ora_player_set_general_idx(eng.e, int(p), v)


@pytest.mark.parametrize("c", by_kind("engine"))
def test_engine(c):
    eng = O.OracleEngine(c["w"], c["h"], c["players"], c["tiles"], setup=(c["setup"] == "new_engine"))
    run_engine_script(c, eng)


@pytest.mark.parametrize("c", by_kind("legal_mask"))
def test_legal_mask(c):
    eng = O.OracleEngine(c["w"], c["h"], c["players"], c["tiles"], setup=False)
    for p, tiles in c["owned"].items():
        eng.set_owned(int(p), tiles)
    for p, v in c["alive"].items():
        eng.L.ora_player_set_alive(eng.e, int(p), int(v))
    m = eng.legal_mask(c["query_player"])
    check_mask_expect(m, c["expect"])


def check_mask_expect(m, e):
    true_idx = [int(i) for i in np.nonzero(m)[0]]
    if "size" in e:
        assert len(m) == e["size"]
    if "true_indices" in e:
        assert true_idx == e["true_indices"]
    for i in e.get("true_indices_subset", []):
        assert m[i]
    for i in e.get("false_indices", []):
        assert not m[i]
    if "any_true_in" in e:
        assert any(m[i] for i in e["any_true_in"])
    if "count_gt" in e:
        assert len(true_idx) > e["count_gt"]
    if "count_lt" in e:
        assert len(true_idx) < e["count_lt"]


# ------------------------------------------------------------------ internal/experience vectors
REWARD_CONFIG = {"WinGame": 1.0, "LoseGame": -1.0, "CaptureCity": 0.1, "LoseCity": -0.1, "CaptureGeneral": 0.5, "LoseGeneral": -0.5,
                 "TerritoryGained": 0.01, "TerritoryLost": -0.01, "ArmyGained": 0.001, "ArmyLost": -0.001, "ArmyAdvantage": 0.05}


def f32_expr(v):
    """'a/b' -> float32(a)/float32(b) like the Go tests compute their expectations."""
    if isinstance(v, str):
        a, b = v.split("/")
        return np.float32(np.float32(float(a)) / np.float32(float(b)))
    return np.float32(v)


def experience_engine(c, tiles, fog=False, alive=None):
    """createTestGameState-style hand-built state: no init pass, Turn 1, both players alive."""
    eng = O.OracleEngine(c["w"], c["h"], c.get("players", 2), tiles, fog=fog, setup=False)
    eng.L.ora_engine_set_turn(eng.e, 1)
    for p, v in (alive or {}).items():
        eng.L.ora_player_set_alive(eng.e, int(p), int(v))
    return eng


def set_visibility(eng, c):
    n = c["w"] * c["h"]
    if c.get("visible_all"):
        for t in range(n):
            eng.tile(t % c["w"], t // c["w"]).visible = (1 << c.get("players", 2)) - 1
    for p, tiles in c.get("visible_tiles", {}).items():
        for t in tiles:
            eng.tile(t % c["w"], t // c["w"]).visible |= 1 << int(p)


@pytest.mark.parametrize("c", by_kind("tensor"))
def test_state_to_tensor(c):
    eng = experience_engine(c, c["tiles"], fog=c["fog"])
    set_visibility(eng, c)
    out = eng.state_to_tensor(c["player"])
    e = c["expect"]
    if "size" in e:
        assert len(out) == e["size"]
    for i, v in e["values"]:
        assert out[i] == f32_expr(v), (i, out[i], v)


@pytest.mark.parametrize("c", by_kind("ser_mask"))
def test_serializer_mask(c):
    eng = experience_engine(c, c["tiles"])
    m = eng.serializer_mask(c["player"])
    assert len(m) == c["expect"]["size"]
    assert all(m[i] for i in c["expect"]["true"]) and not any(m[i] for i in c["expect"]["false"])


def expected_reward(c, prev, cur):
    L = O.lib()
    total = np.float32(0)
    for name, mult in c["expect"]["terms"]:
        if name == "ArmyAdvantageCur":
            total += np.float32(L.ora_army_advantage(cur.e, c["player"])) * np.float32(REWARD_CONFIG["ArmyAdvantage"])
        elif name == "ArmyAdvantageDelta":
            total += (np.float32(L.ora_army_advantage(cur.e, c["player"])) - np.float32(L.ora_army_advantage(prev.e, c["player"]))) * np.float32(REWARD_CONFIG["ArmyAdvantage"])
        else:
            total += np.float32(mult) * np.float32(REWARD_CONFIG[name])
    return total


@pytest.mark.parametrize("c", by_kind("reward"))
def test_calculate_reward(c):
    prev, cur = experience_engine(c, c["prev"]), experience_engine(c, c["cur"], alive=c.get("alive_cur"))
    r = np.float32(O.lib().ora_calculate_reward(prev.e, cur.e, c["player"]))
    exp = expected_reward(c, prev, cur)
    if c["expect"]["delta"] == 0:
        assert r == exp, (r, exp)
    else:
        assert abs(float(r) - float(exp)) <= c["expect"]["delta"], (r, exp)


@pytest.mark.parametrize("c", by_kind("army_advantage"))
def test_army_advantage(c):
    eng = experience_engine(c, c["tiles"])
    for p in ("0", "1"):
        assert abs(float(O.lib().ora_army_advantage(eng.e, int(p))) - float(f32_expr(c["expect"][p]))) <= c["expect"]["delta"]


@pytest.mark.parametrize("c", by_kind("city_changes"))
def test_city_changes(c):
    prev, cur = experience_engine(c, c["prev"]), experience_engine(c, c["cur"])
    g, l = C.c_int32(), C.c_int32()
    O.lib().ora_city_changes(prev.e, cur.e, c["player"], C.byref(g), C.byref(l))
    assert (g.value, l.value) == (c["expect"]["gained"], c["expect"]["lost"])


def _neighbour(x, y, w, h):
    return (x + 1, y) if x + 1 < w else (x - 1, y)


@pytest.mark.parametrize("c", by_kind("board_idx"))
def test_board_idx(c):
    """Board.Idx (core/board.go:108): a move FROM (x, y) reads the tile at Idx(x, y) - it validates only when the player's army
    sits at the index the reference test expects (core/action.go:82-89), and fails ErrNotOwned anywhere else."""
    L, w, h = O.lib(), c["w"], c["h"]
    for x, y, idx in c["xy_idx"]:
        for at in range(w * h):
            b = L.ora_board_new(w, h)
            b.contents.t[at].owner, b.contents.t[at].army = 0, 5
            tx, ty = _neighbour(x, y, w, h)
            rc = L.ora_validate(b, C.byref(O.Move(0, x, y, tx, ty, 1)), 0)
            assert rc == (0 if at == idx else 3), (x, y, idx, at, rc)
            L.ora_board_free(b)


def fog_square(x, y, w, h):
    return sorted(yy * w + xx for yy in range(max(0, y - 1), min(h, y + 2)) for xx in range(max(0, x - 1), min(w, x + 2)))


def xy_board(c, idx):
    """P0's general at tile `idx`, P1's in the corner farthest from it."""
    w, h = c["w"], c["h"]
    far = (0 if (idx % w) * 2 >= w else w - 1) + w * (0 if (idx // w) * 2 >= h else h - 1)
    return [{"x": idx % w, "y": idx // w, "owner": 0, "army": 5, "type": 1}, {"x": far % w, "y": far // w, "owner": 1, "army": 5, "type": 1}]


@pytest.mark.parametrize("c", by_kind("board_xy"))
def test_board_xy(c):
    """Board.XY (core/board.go:111-113) as the fog update uses it (visibility_optimized.go:101,120,133,154): the 3x3 a
    player sees around its only tile, at index idx, is centred on the (x, y) the reference test expects."""
    for idx, x, y in c["idx_xy"]:
        eng = O.OracleEngine(c["w"], c["h"], 2, xy_board(c, idx), fog=True)
        vis, _ = eng.player_visibility(0)
        assert sorted(np.flatnonzero(vis)) == fog_square(x, y, c["w"], c["h"]), (idx, x, y)


@pytest.mark.parametrize("c", by_kind("collector"))
def test_collector_vectors(c):
    """SimpleCollector.OnStateTransition's per-experience fields from the oracle's pieces (collector.go:41-56)."""
    prev = experience_engine(c, c["prev"])
    cur = experience_engine(c, c["cur"], alive=c.get("alive_cur"))
    if "cur_turn" in c:
        cur.L.ora_engine_set_turn(cur.e, c["cur_turn"])
    e, n = c["expect"], c["w"] * c["h"]
    exps = []
    for a in c["actions"]:                               # one experience per player that submitted an action (:33-37)
        p = a["player"]
        exps.append({"player_id": p, "turn": cur.turn, "state": prev.state_to_tensor(p), "next_state": cur.state_to_tensor(p),
                     "mask": prev.serializer_mask(p), "reward": np.float32(O.lib().ora_calculate_reward(prev.e, cur.e, p)),
                     "done": sum(cur.alive(q) for q in range(2)) <= 1})   # GameState.IsGameOver (state.go:73-82)
    assert len(exps) == e["count"] and sorted(x["player_id"] for x in exps) == e["player_ids"]
    x = exps[0]
    if "turn" in e:
        assert x["turn"] == e["turn"]
    if "done" in e:
        assert x["done"] == e["done"]
    if "tensor_shape" in e:
        assert len(x["state"]) == len(x["next_state"]) == int(np.prod(e["tensor_shape"])) and len(x["mask"]) == e["mask_len"]
    if "reward" in e:
        assert x["reward"] == np.float32(e["reward"])                      # assert.Equal(float32(1.0), ...): exact


def test_oracle_is_clean_under_sanitizers():
    """address/UB sanitizers on the CPU build (GPU ASan is not available on this pool)."""
    import shutil, subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    r = subprocess.run(["make", "-C", os.path.join(os.path.dirname(HERE), "oracle"), "-s", "sanitize"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sanitizer run OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_rollout_with_mask_packing_plays_the_same_games():
    """bench.py's CPU baseline packs the legal masks after every turn (the GPU leg's work): the games are the same, and the
    masks it leaves are the state's."""
    import _harness as H
    B = 48
    army, owner, typ, ws, hs, ps = H.gen_boards(1, [(20, 20, 4)] * B, 20, 20)
    a, b = O.OracleBatch(B, 20, 20, 4), O.OracleBatch(B, 20, 20, 4)
    for x in (a, b):
        x.reset(army, owner, typ, ws, hs, ps)
    bits = np.zeros((B, 4, a.mask_bytes), np.uint8)
    assert a.rollout(40, 3, 5) == b.rollout(40, 3, 5, legal_bits=bits)
    sa, sb = a.read_state(), b.read_state()
    assert all(np.array_equal(sa[f], sb[f]) for f in sa) and np.array_equal(bits, a.legal_mask())
