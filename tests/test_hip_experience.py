"""GPU parity of the internal/experience side channel: observation tensors,
serializer masks and rewards from the HIP path vs the CPU oracle.  float32 values must be
bit-identical: same operation order, no fused multiply-add (tolerance 0)."""
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


def _hand_built(g, c, tiles, fog=False, visible=None):
    """createTestGameState-style state on the HIP engine: reset, then poke Turn 1 / alive / visibility."""
    w, h, P = c["w"], c["h"], c.get("players", 2)
    eng = g.VecEngine(1, w, h, P, fog_of_war=fog)
    army, owner, typ = O.planes_from_tiles(w, h, tiles)
    eng.reset(army[None], owner[None], typ[None])
    upd = {"turn": np.array([1], np.int32), "alive": np.ones((1, P), np.uint8), "done": np.zeros(1, np.uint8)}
    if visible is not None:
        upd["visible"] = visible[None]
    eng.write_state(upd)
    return eng


def _visible_plane(c):
    n = c["w"] * c["h"]
    v = np.zeros(n, np.uint8)
    if c.get("visible_all"):
        v[:] = (1 << c.get("players", 2)) - 1
    for p, tiles in c.get("visible_tiles", {}).items():
        v[tiles] |= 1 << int(p)
    return v


@pytest.mark.parametrize("c", by_kind("tensor"))
def test_golden_state_to_tensor(g, c):
    from test_oracle_golden import f32_expr
    eng = _hand_built(g, c, c["tiles"], fog=c["fog"], visible=_visible_plane(c))
    out = eng.observe(c["player"])[0][: 9 * c["w"] * c["h"]]
    for i, v in c["expect"]["values"]:
        assert out[i] == f32_expr(v), (i, out[i], v)


@pytest.mark.parametrize("c", by_kind("ser_mask"))
def test_golden_serializer_mask(g, c):
    eng = _hand_built(g, c, c["tiles"])
    m = g.unpack_legal_bits(eng.serializer_mask_bits()[0, c["player"]], c["w"], c["h"])
    assert len(m) == c["expect"]["size"]
    assert all(m[i] for i in c["expect"]["true"]) and not any(m[i] for i in c["expect"]["false"])


@pytest.mark.parametrize("c", by_kind("reward"))
def test_golden_rewards(g, c):
    """prev / cur are two hand-built states (rewards_test.go): snapshot prev, overwrite with cur."""
    from test_oracle_golden import expected_reward, experience_engine
    eng = _hand_built(g, c, c["prev"])
    eng.experience_begin()
    army, owner, typ = O.planes_from_tiles(c["w"], c["h"], c["cur"])
    eng.write_state({"army": army[None], "owner": owner[None], "type": typ[None], "turn": np.array([2], np.int32)})
    r, done = eng.experience_rewards()
    prev, cur = experience_engine(c, c["prev"]), experience_engine(c, c["cur"], alive=c.get("alive_cur"))
    exp = expected_reward(c, prev, cur)
    assert not done[0]
    assert r[0, c["player"]] == np.float32(O.lib().ora_calculate_reward(prev.e, cur.e, c["player"]))  # bit-exact vs oracle
    assert abs(float(r[0, c["player"]]) - float(exp)) <= max(c["expect"]["delta"], 0), (r, exp)


@pytest.mark.parametrize("c", by_kind("collector"))
def test_golden_collector(g, c):
    """collector_test.go's vectors through the HIP experience channel, both ways a consumer can take it: VecExperienceCollector
    (gvec_observe / gvec_serializer_mask / gvec_experience_begin / _rewards) and the compact records a rank ships
    (gvec_experience_records -> decode_records).  The reference test swaps in a hand-built currState; here the engine is
    poked into it.  The ABI treats an env whose Turn did not advance as "not stepped" (reward 0, record flagged void), so
    cur.Turn = prev.Turn + 1 where the Go test leaves it alone - Turn is not what those vectors assert."""
    import torch
    from generalsreinforcementlearning_amd.experience import VecExperienceCollector, decode_records
    w, h, e = c["w"], c["h"], c["expect"]
    eng = _hand_built(g, c, c["prev"])
    col = VecExperienceCollector(eng)
    col.before_step()
    acts = g.make_actions(1, 2, [(0, a["player"], a["from"][0], a["from"][1], a["to"][0], a["to"][1], a["move_all"]) for a in c["actions"]])
    army, owner, typ = O.planes_from_tiles(w, h, c["cur"])
    alive = np.ones((1, 2), np.uint8)
    for p, v in c.get("alive_cur", {}).items():
        alive[0, int(p)] = int(v)
    eng.write_state({"army": army[None], "owner": owner[None], "type": typ[None], "alive": alive,
                     "turn": np.array([c.get("cur_turn", 2)], np.int32)})
    batch = col.after_step(acts)
    slab = torch.zeros(eng.experience_record_bytes(), dtype=torch.uint8, device="cuda")
    eng.experience_records(slab.data_ptr(), actions=acts)
    eng.synchronize()
    dec = decode_records(slab.cpu().numpy(), eng.experience_record_layout())
    for b in (batch, dec):
        assert len(b["player_id"]) == e["count"] and sorted(int(p) for p in b["player_id"]) == e["player_ids"]
        if "turn" in e:
            assert int(b["turn"][0]) == e["turn"]
        if "done" in e:
            assert bool(b["done"][0]) == e["done"]
        if "tensor_shape" in e:
            assert list(np.asarray(b["state"][0]).shape) == e["tensor_shape"] == list(np.asarray(b["next_state"][0]).shape)
            assert len(b["action_mask"][0]) == e["mask_len"]
        if "reward" in e:
            assert b["reward"].dtype == np.float32 and b["reward"][0] == np.float32(e["reward"])
    assert all(dec["valid"])
    d = col.as_dicts(batch)[0]
    assert set(d) >= {"experience_id", "game_id", "player_id", "turn", "state", "action", "reward", "next_state", "done", "action_mask"}


@pytest.mark.parametrize("name,B,pattern,fog", [("20x20_p4", 128, [(20, 20, 4)], True), ("mixed", 96, [(10, 10, 2), (15, 15, 3), (20, 20, 4)], True),
                                               ("10x10_fog_off", 128, [(10, 10, 2)], False), ("tiny", 48, [(3, 3, 2), (5, 7, 3), (8, 8, 2)], True)],
                         ids=lambda v: v if isinstance(v, str) else None)
def test_experience_channel_matches_oracle(g, name, B, pattern, fog):
    sizes = [pattern[i % len(pattern)] for i in range(B)]
    mw, mh, mp = max(s[0] for s in sizes), max(s[1] for s in sizes), max(s[2] for s in sizes)
    army, owner, typ, w, h, p = H.gen_boards(99, sizes, mw, mh)
    eng = g.VecEngine(B, mw, mh, mp, fog_of_war=fog)
    ora = O.OracleBatch(B, mw, mh, mp, fog=fog)
    eng.reset(army, owner, typ, w, h, p)
    ora.reset(army, owner, typ, w, h, p)
    saw_terminal = False
    for k in range(150):
        acts = ora.agent_actions(5, 10)
        eng.experience_begin()
        ora.experience_begin()
        assert np.array_equal(eng.step(acts), ora.step(acts))
        hr, hd = eng.experience_rewards()
        orr, od = ora.rewards()
        assert np.array_equal(hr.view(np.uint32), orr.view(np.uint32)), f"{name} turn {k}: rewards differ {hr[hr != orr][:6]} vs {orr[hr != orr][:6]}"
        assert np.array_equal(hd, od.astype(bool))
        saw_terminal |= bool(od.any())
        if k % 10 == 0:
            assert np.array_equal(eng.serializer_mask_bits(), ora.serializer_mask())
            for player in range(mp):
                assert np.array_equal(eng.observe(player).view(np.uint32), ora.observe(player).view(np.uint32)), (name, k, player)
            allp = eng.observe(-1)
            assert np.array_equal(allp[:, 1].view(np.uint32), ora.observe(1).view(np.uint32))
    assert np.abs(orr).sum() > 0
    H.assert_states_equal(eng.game_state(), ora.read_state(), name)


def test_rewards_at_game_end_and_after_redeal(g):
    """Terminal +-1 (rewards.go:49-56) and the 'no predecessor' rule for re-dealt boards."""
    B = 128
    sizes = [(6, 6, 2)] * B
    army, owner, typ, w, h, p = H.gen_boards(3, sizes, 6, 6)
    eng = g.VecEngine(B, 6, 6, 2, auto_reset=True)
    ora = O.OracleBatch(B, 6, 6, 2)
    eng.reset(army, owner, typ, w, h, p)
    ora.reset(army, owner, typ, w, h, p)
    eng.build_board_pool(17, 4)
    ora.set_pool(17, 4)
    wins = 0
    for k in range(400):
        acts = ora.agent_actions(8)
        eng.experience_begin()
        ora.experience_begin()
        assert np.array_equal(eng.step(acts), ora.step(acts))
        hr, hd = eng.experience_rewards()
        orr, od = ora.rewards()
        assert np.array_equal(hr.view(np.uint32), orr.view(np.uint32)), k
        wins += int((orr == 1.0).sum())
    assert wins > 0, "no game finished: the terminal branch was not exercised"


def test_collector_layout_matches_stream_client(g):
    """VecExperienceCollector vs a per-experience restatement of SimpleCollector.OnStateTransition
    (collector.go:30-98) on the oracle; dict keys of experience_stream_client.py:146-157."""
    from generalsreinforcementlearning_amd.experience import VecExperienceCollector
    B, w, h, P = 32, 12, 12, 3
    sizes = [(w, h, P)] * B
    army, owner, typ, ws, hs, ps = H.gen_boards(8, sizes, w, h)
    eng = g.VecEngine(B, w, h, P)
    ora = O.OracleBatch(B, w, h, P)
    eng.reset(army, owner, typ, ws, hs, ps)
    ora.reset(army, owner, typ, ws, hs, ps)
    col = VecExperienceCollector(eng)
    n = 0
    for k in range(30):
        acts = ora.agent_actions(4)
        prev_obs = [ora.observe(p) for p in range(P)]
        prev_mask = ora.serializer_mask()
        col.before_step()
        ora.experience_begin()
        eng.step(acts)
        ora.step(acts)
        batch = col.after_step(acts)
        orr, od = ora.rewards()
        cur_obs = [ora.observe(p) for p in range(P)]
        for i, (e, p) in enumerate(zip(batch["env"], batch["player_id"])):
            assert acts[e, p]["flags"] & 1
            assert np.array_equal(batch["state"][i].ravel(), prev_obs[p][e]) and np.array_equal(batch["next_state"][i].ravel(), cur_obs[p][e])
            assert batch["reward"][i] == orr[e, p] and batch["done"][i] == bool(od[e])
            a = acts[e, p]
            dx, dy = int(a["to_x"]) - int(a["from_x"]), int(a["to_y"]) - int(a["from_y"])
            d = {(0, -1): 0, (0, 1): 1, (-1, 0): 2, (1, 0): 3}[(dx, dy)]
            assert batch["action"][i] == (int(a["from_y"]) * w + int(a["from_x"])) * 4 + d
            assert np.array_equal(batch["action_mask"][i], g.unpack_legal_bits(prev_mask[e, p], w, h))
            assert batch["action_mask"][i][batch["action"][i]]  # the agent's move was legal in the serializer's mask too
        n += len(batch["env"])
    d = col.as_dicts(batch)[0]
    assert set(d) == {"experience_id", "game_id", "player_id", "turn", "state", "action", "reward", "next_state", "done", "action_mask"}
    assert d["state"].shape == (9, h, w) and d["state"].dtype == np.float32 and d["action_mask"].dtype == np.bool_
    assert n > 1000


def test_collector_mixed_board_sizes(g):
    """A padded batch of 10x10 / 15x15 / 20x20 boards: every experience is shaped by its own env's board
    (TensorState.shape = [9, H, W], collector.go:64,71), action and mask indices use that env's width."""
    from generalsreinforcementlearning_amd.experience import VecExperienceCollector
    B = 48
    sizes = [[(10, 10, 2), (15, 15, 3), (20, 20, 4)][i % 3] for i in range(B)]
    army, owner, typ, ws, hs, ps = H.gen_boards(14, sizes, 20, 20)
    eng = g.VecEngine(B, 20, 20, 4)
    ora = O.OracleBatch(B, 20, 20, 4)
    eng.reset(army, owner, typ, ws, hs, ps)
    ora.reset(army, owner, typ, ws, hs, ps)
    col = VecExperienceCollector(eng)
    n = 0
    for k in range(25):
        acts = ora.agent_actions(6)
        prev = [ora.engine(e) for e in range(B)]
        prev_obs = {(e, p): prev[e].state_to_tensor(p) for e in range(B) for p in range(sizes[e][2])}
        prev_mask = {(e, p): prev[e].serializer_mask(p) for e in range(B) for p in range(sizes[e][2])}
        col.before_step()
        ora.experience_begin()
        eng.step(acts)
        ora.step(acts)
        batch = col.after_step(acts)
        orr, od = ora.rewards()
        assert isinstance(batch["state"], list)   # ragged
        for i, (e, p) in enumerate(zip(batch["env"], batch["player_id"])):
            w, h, _ = sizes[e]
            assert batch["state"][i].shape == (9, h, w) and batch["next_state"][i].shape == (9, h, w)
            assert np.array_equal(batch["state"][i].ravel(), prev_obs[(e, p)])
            assert np.array_equal(batch["next_state"][i].ravel(), ora.engine(e).state_to_tensor(p))
            a = acts[e, p]
            dx, dy = int(a["to_x"]) - int(a["from_x"]), int(a["to_y"]) - int(a["from_y"])
            d = {(0, -1): 0, (0, 1): 1, (-1, 0): 2, (1, 0): 3}[(dx, dy)]
            assert batch["action"][i] == (int(a["from_y"]) * w + int(a["from_x"])) * 4 + d
            assert batch["action_mask"][i].shape == (w * h * 4,)
            assert np.array_equal(batch["action_mask"][i], prev_mask[(e, p)].astype(bool))
            assert batch["action_mask"][i][batch["action"][i]]
            assert batch["reward"][i] == orr[e, p] and batch["done"][i] == bool(od[e])
        n += len(batch["env"])
    dd = col.as_dicts(batch)
    assert {d["state"].shape for d in dd} <= {(9, 10, 10), (9, 15, 15), (9, 20, 20)} and len({d["state"].shape for d in dd}) > 1
    assert n > 500


@pytest.mark.parametrize("sizes,auto_reset", [([(20, 20, 4)], False), ([(10, 10, 2), (15, 15, 3), (20, 20, 4)], False), ([(6, 6, 2)], True),
                                               ([(32, 32, 8), (25, 25, 5)], False)],
                         ids=["20x20_p4", "mixed", "6x6_autoreset", "32x32_p8"])
def test_experience_records_match_oracle_encoding_and_collector(g, sizes, auto_reset):
    """gvec_experience_records: (1) byte for byte the records a numpy restatement of the format builds from the
    oracle; (2) decoded by the product decoder they are the batch VecExperienceCollector builds from the device
    tensors (StateToTensor / GenerateActionMask / CalculateReward on the GPU) - so what crosses xGMI in compact
    form expands to exactly what SimpleCollector.OnStateTransition would have put on the wire."""
    import torch
    import _records as R
    from generalsreinforcementlearning_amd.experience import VecExperienceCollector, decode_records, expand_records_device
    B = 36
    per_env = [sizes[i % len(sizes)] for i in range(B)]
    mw, mh, mp = max(s[0] for s in per_env), max(s[1] for s in per_env), max(s[2] for s in per_env)
    army, owner, typ, ws, hs, ps = H.gen_boards(23, per_env, mw, mh)
    army[::5][typ[::5] == 1] = 70000                        # some wide envs: the record saturates at 65,535 (exact for the tensor)
    eng = g.VecEngine(B, mw, mh, mp, auto_reset=auto_reset)
    ora = O.OracleBatch(B, mw, mh, mp)
    eng.reset(army, owner, typ, ws, hs, ps)
    ora.reset(army, owner, typ, ws, hs, ps)
    if auto_reset:
        eng.build_board_pool(9, 3)
        ora.set_pool(9, 3)
    lay = eng.experience_record_layout()
    assert lay == R.layout_for(mw, mh, mp) and eng.experience_record_bytes() == 4 * lay["record_dw"]
    col = VecExperienceCollector(eng)
    slab = torch.zeros(B * lay["record_dw"], dtype=torch.int32, device="cuda")
    n = invalid = 0
    for k in range(60 if not auto_reset else 250):
        acts = ora.agent_actions(31, 5)
        snap = R.capture(ora)
        col.before_step()                                   # includes gvec_experience_begin
        ora.experience_begin()
        assert np.array_equal(eng.step(acts), ora.step(acts))
        eng.experience_records(slab.data_ptr(), actions=acts, env_id_base=1000)
        eng.synchronize()
        got = slab.cpu().numpy().view(np.uint32).reshape(B, lay["record_dw"])
        want = R.encode(ora, snap, acts, lay, env_id_base=1000)
        assert np.array_equal(got, want), f"turn {k}: records differ in envs {np.unique(np.argwhere(got != want)[:, 0])[:6]} at dwords {np.unique(np.argwhere(got != want)[:, 1])[:8]}"
        invalid += int((((got[:, 1] >> 24) & 4) == 0).sum())   # re-dealt (or frozen) envs: flagged, nobody acted in them
        if k % 6 == 0 or auto_reset:
            batch = col.after_step(acts)
            dec = decode_records(got, lay)
            assert np.array_equal(dec["env"], batch["env"] + 1000) and np.array_equal(dec["player_id"], batch["player_id"])
            for f in ("turn", "action", "done"):
                assert np.array_equal(dec[f], batch[f]), f
            assert np.array_equal(dec["reward"].view(np.uint32), batch["reward"].view(np.uint32))
            for i in range(len(dec["env"])):
                assert dec["valid"][i]                        # a re-dealt env has no actor: its record yields no experience
                for f in ("state", "next_state"):
                    assert np.array_equal(np.asarray(dec[f][i]).view(np.uint32), np.asarray(batch[f][i]).view(np.uint32)), (k, i, f)
                assert np.array_equal(dec["action_mask"][i], batch["action_mask"][i])
            # the GPU-side consumer (gvec_expand_experience_records) expands the same slab to the same experiences
            ex = expand_records_device(slab, lay)
            assert len(ex["env"]) == len(dec["env"])
            for f in ("env", "player_id", "turn", "action", "done", "width", "height"):
                assert np.array_equal(ex[f].cpu().numpy(), np.asarray(dec[f])), (k, f)
            assert np.array_equal(ex["reward"].cpu().numpy().view(np.uint32), dec["reward"].view(np.uint32))
            xs, xn, xm = ex["state"].cpu().numpy(), ex["next_state"].cpu().numpy(), ex["action_mask"].cpu().numpy()
            for i in range(len(dec["env"])):
                hw = int(dec["width"][i]) * int(dec["height"][i])
                for got, want in ((xs[i], dec["state"][i]), (xn[i], dec["next_state"][i])):
                    assert np.array_equal(got[: 9 * hw].view(np.uint32), np.asarray(want).reshape(-1).view(np.uint32)) and not got[9 * hw:].any(), (k, i)
                assert np.array_equal(xm[i][: 4 * hw], dec["action_mask"][i]) and not xm[i][4 * hw:].any()
            n += len(dec["env"])
    assert n > 200 and (invalid > 0) == auto_reset
    # a range of envs, the device agent's own actions
    eng.record_agent_actions(True)
    eng.experience_begin_range(8, 16)
    snap = R.capture(ora)
    ora.experience_begin()
    oacts = ora.agent_actions(77)
    eng.rollout(1, 77, 0, fused=False)
    ora.step(oacts)
    eng.experience_records(slab.data_ptr(), actions=None, env_begin=8, n=16, env_id_base=0)
    eng.synchronize()
    got = slab.cpu().numpy().view(np.uint32).reshape(B, lay["record_dw"])[:16]
    assert np.array_equal(got, R.encode(ora, snap, oacts, lay, envs=range(8, 24)))


@pytest.mark.gpu
def test_expand_records_rejects_what_is_not_a_layout_and_ignores_malformed_records(g):
    import ctypes as C
    import torch
    from generalsreinforcementlearning_amd.experience import expand_records_device
    eng = g.VecEngine(8, 10, 10, 2)
    lay = eng.experience_record_layout()
    bad = dict(lay, record_dw=lay["record_dw"] - 40)
    with pytest.raises(g.GvecError):
        expand_records_device(torch.zeros(8 * bad["record_dw"], dtype=torch.int32, device="cuda"), bad)
    # records whose header claims a board beyond the layout, or more players than slots: no experience, no out-of-bounds access
    junk = torch.zeros((8, lay["record_dw"]), dtype=torch.int32, device="cuda")
    junk[:, 1] = (200 | (200 << 8) | (7 << 16) | (4 << 24))
    junk[:, 2] = 0xFF
    ex = expand_records_device(junk, lay)
    assert len(ex["env"]) == 0


@pytest.mark.parametrize("maxp", [2, 4, 8])
@pytest.mark.parametrize("slots,parity", sorted(H.VARIANT_DIMS), ids=H.VARIANT_IDS)
def test_experience_channel_on_every_kernel_variant(g, maxp, slots, parity):
    """Rewards every turn, StateToTensor / GenerateActionMask for every player, the compact records byte for byte, and their
    expansion on the device - on the board limits that select each compiled <players, slots, parity> instantiation."""
    import torch
    import _records as R
    from generalsreinforcementlearning_amd.experience import decode_records, expand_records_device
    B = 12
    mw, mh, sizes = H.variant_batch(maxp, slots, parity, B)
    army, owner, typ, w, h, p = H.gen_boards(500 + slots, sizes, mw, mh)
    eng = g.VecEngine(B, mw, mh, maxp, fog_of_war=True)
    ora = O.OracleBatch(B, mw, mh, maxp, fog=True)
    eng.reset(army, owner, typ, w, h, p)
    ora.reset(army, owner, typ, w, h, p)
    t0 = (18 + np.arange(B) % 7).astype(np.int32)            # the growth turn (25) falls inside the run
    eng.write_state({"turn": t0})
    ora.write_state({"turn": t0})
    lay = eng.experience_record_layout()
    assert lay == R.layout_for(mw, mh, maxp)
    slab = torch.zeros(B * lay["record_dw"], dtype=torch.int32, device="cuda")
    ctx = f"<{maxp},{slots},{parity}>"
    for k in range(12):
        acts = ora.agent_actions(41, 10)
        snap = R.capture(ora)
        eng.experience_begin()
        ora.experience_begin()
        assert np.array_equal(eng.step(acts), ora.step(acts))
        hr, hd = eng.experience_rewards()
        orr, od = ora.rewards()
        assert np.array_equal(hr.view(np.uint32), orr.view(np.uint32)) and np.array_equal(hd, od.astype(bool)), (ctx, k)
        eng.experience_records(slab.data_ptr(), actions=acts, env_id_base=7)
        eng.synchronize()
        got = slab.cpu().numpy().view(np.uint32).reshape(B, lay["record_dw"])
        want = R.encode(ora, snap, acts, lay, env_id_base=7)
        assert np.array_equal(got, want), (ctx, k, np.unique(np.argwhere(got != want)[:, 0])[:6])
        if k % 4 == 3:
            assert np.array_equal(eng.serializer_mask_bits(), ora.serializer_mask()), (ctx, k)
            for player in range(maxp):
                assert np.array_equal(eng.observe(player).view(np.uint32), ora.observe(player).view(np.uint32)), (ctx, k, player)
            dec = decode_records(got, lay)
            ex = expand_records_device(slab, lay)
            assert len(ex["env"]) == len(dec["env"]) > 0
            for f in ("env", "player_id", "turn", "action", "done", "width", "height"):
                assert np.array_equal(ex[f].cpu().numpy(), np.asarray(dec[f])), (ctx, k, f)
            assert np.array_equal(ex["reward"].cpu().numpy().view(np.uint32), dec["reward"].view(np.uint32))
            xs, xn, xm = ex["state"].cpu().numpy(), ex["next_state"].cpu().numpy(), ex["action_mask"].cpu().numpy()
            for i in range(len(dec["env"])):
                hw = int(dec["width"][i]) * int(dec["height"][i])
                for a, b in ((xs[i], dec["state"][i]), (xn[i], dec["next_state"][i])):
                    assert np.array_equal(a[: 9 * hw].view(np.uint32), np.asarray(b).reshape(-1).view(np.uint32)) and not a[9 * hw:].any(), (ctx, k, i)
                assert np.array_equal(xm[i][: 4 * hw], dec["action_mask"][i]) and not xm[i][4 * hw:].any()
    H.assert_states_equal(eng.game_state(), ora.read_state(), ctx)


def test_record_replay_ring_holds_what_the_engine_wrote_and_expands_what_it_draws(g):
    """RecordReplayRing: the ring's rows are the records gvec_experience_records writes for each step (wrapping included),
    and a draw expands to exactly what decode_records makes of the drawn rows."""
    import torch
    from generalsreinforcementlearning_amd.experience import RecordReplayRing, decode_records
    B, n_rec, steps = 48, 20, 9
    sizes = [[(10, 10, 2), (12, 9, 3), (8, 8, 2)][i % 3] for i in range(B)]
    army, owner, typ, w, h, p = H.gen_boards(61, sizes, 12, 10)
    eng = g.VecEngine(B, 12, 10, 3, auto_reset=True)
    eng.reset(army, owner, typ, w, h, p)
    eng.build_board_pool(7, 2)
    eng.record_agent_actions(True)
    lay = eng.experience_record_layout()
    ring = RecordReplayRing(eng, capacity_records=70, seed=3)          # 9 steps x 20 records: wraps twice
    slab = torch.zeros(n_rec * lay["record_dw"], dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        ring.sample(1)
    want_rows = []
    for k in range(steps):
        eng.experience_begin_range(4, n_rec)
        eng.rollout(1, 900 + k, 5, fused=False)
        ring.append_step(None, 4, n_rec, env_id_base=100)
        eng.experience_records(slab.data_ptr(), None, 4, n_rec, 100)
        eng.synchronize()
        want_rows.append(slab.cpu().numpy().view(np.uint8).reshape(n_rec, -1).copy())
    allrows = np.concatenate(want_rows)
    assert ring.total_appended == steps * n_rec and len(ring) == 70 and ring.cursor == (steps * n_rec) % 70
    held = ring.ring.cpu().numpy()
    for j in range(steps * n_rec - 70, steps * n_rec):                  # the last 70 records, where the ring put them
        assert np.array_equal(held[j % 70], allrows[j]), j
    idx = ring.sample_indices(32)
    assert idx.unique().numel() == 32 and int(idx.max()) < 70
    got = ring.sample(32, indices=idx)
    dec = decode_records(held[idx.cpu().numpy()].view(np.uint32), lay)
    assert len(got["env"]) == len(dec["env"]) >= 32
    for f in ("env", "player_id", "turn", "action", "done", "width", "height"):
        assert np.array_equal(got[f].cpu().numpy(), np.asarray(dec[f])), f
    assert np.array_equal(got["reward"].cpu().numpy().view(np.uint32), dec["reward"].view(np.uint32))
    xs, xn, xm = got["state"].cpu().numpy(), got["next_state"].cpu().numpy(), got["action_mask"].cpu().numpy()
    for i in range(len(dec["env"])):
        hw = int(dec["width"][i]) * int(dec["height"][i])
        assert np.array_equal(xs[i][: 9 * hw].view(np.uint32), np.asarray(dec["state"][i]).reshape(-1).view(np.uint32))
        assert np.array_equal(xn[i][: 9 * hw].view(np.uint32), np.asarray(dec["next_state"][i]).reshape(-1).view(np.uint32))
        assert np.array_equal(xm[i][: 4 * hw], dec["action_mask"][i])
    with pytest.raises(ValueError):
        ring.append_step(None, 0, 71)
