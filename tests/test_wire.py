"""generalsreinforcementlearning_amd.wire: the reference's proto surface served from the batched engine.

In the build container the reference's committed stubs (python/generals_pb) are importable: the schema declared in
wire.py must describe the same wire format (field tables) and messages must cross-parse.  Everywhere (also on the
GPU box, where the reference does not exist) the adapters are checked against a scalar restatement of
convertGameStateToProto / createStreamUpdate (internal/grpc/gameserver/server.go:526-777) on oracle states."""
import os
import sys

import numpy as np
import pytest

import _harness as H
import _oracle as O
from generalsreinforcementlearning_amd import wire
from generalsreinforcementlearning_amd.experience import ExperienceBatcher
from generalsreinforcementlearning_amd.vec_engine import unpack_legal_bits

REF = os.environ.get("GRL_REFERENCE_DIR", "/root/reference")
HAVE_REF = os.path.exists(os.path.join(REF, "python", "generals_pb"))


def _ref_modules():
    sys.path.insert(0, os.path.join(REF, "python"))
    from generals_pb.common.v1 import common_pb2
    from generals_pb.experience.v1 import experience_pb2
    from generals_pb.game.v1 import game_pb2
    return common_pb2, game_pb2, experience_pb2


@pytest.mark.skipif(not HAVE_REF, reason="the reference's stubs exist in the build container only")
def test_schema_is_the_references_wire_format():
    common_pb2, game_pb2, experience_pb2 = _ref_modules()
    mods = {"generals.common.v1": common_pb2, "generals.game.v1": game_pb2, "generals.experience.v1": experience_pb2}
    checked = 0
    for spec in wire._SCHEMA.values():
        ref = mods[spec["package"]]
        for ename, values in spec["enums"].items():
            rd = ref.DESCRIPTOR.enum_types_by_name[ename]
            assert [(v.name, v.number) for v in rd.values] == [(v, i) for i, v in enumerate(values)], ename
        for mname in spec["messages"]:
            mine = wire.POOL.FindMessageTypeByName(f"{spec['package']}.{mname}")
            theirs = ref.DESCRIPTOR.message_types_by_name[mname]
            tab = lambda d: {f.number: (f.name, f.type, f.is_repeated if hasattr(f, 'is_repeated') else None, f.message_type.full_name if f.message_type else None,
                                        f.enum_type.full_name if f.enum_type else None,
                                        f.containing_oneof.name if f.containing_oneof else None) for f in d.fields}
            a, b = tab(mine), tab(theirs)
            if mname == "GameUpdate":
                b.pop(3)          # GameUpdate.event (game events): not produced by the turn engine, not declared
            assert a == b, (mname, a, b)
            checked += len(a)
    assert checked > 50


def _oracle_views(ora, viewer):
    st = ora.read_state()
    B = ora.B
    vis = np.zeros((B, ora.stride), bool)
    fog = np.zeros((B, ora.stride), bool)
    for e in range(B):
        v, f = ora.engine(e).player_visibility(viewer)
        vis[e, :len(v)], fog[e, :len(f)] = v, f
    return st, vis, fog


def _scalar_game_state(eng, viewer, legal):
    """convertGameStateToProto restated tile by tile on one oracle engine (server.go:526-610) -> plain dict."""
    v, f = eng.player_visibility(viewer)
    tiles = []
    for t in range(eng.w * eng.h):
        tl = eng.tile(t % eng.w, t // eng.w)
        typ, owner, army = {0: 1, 1: 2, 2: 3, 3: 4}[tl.type], tl.owner, tl.army
        if not v[t] and not f[t]:
            typ, owner, army = 1, -1, 0
        elif f[t] and not v[t]:
            owner, army = -1, 0
        tiles.append((typ, owner, army, bool(v[t]), bool(f[t])))
    players = []
    for p in range(eng.p):
        gi = eng.general_idx(p)
        pos = None
        if gi >= 0 and (not eng.alive(p) or eng.tile(gi % eng.w, gi // eng.w).owner == viewer):
            pos = (gi % eng.w, gi // eng.w)
        players.append((p, 1 if eng.alive(p) else 2, eng.army_count(p), len(eng.owned(p)), pos, "#%06X" % (p * 0x333333)))
    return {"turn": eng.turn, "tiles": tiles, "players": players, "winner": eng.winner if eng.game_over else -1, "mask": [bool(x) for x in legal]}


def _proto_as_dict(gs):
    return {"turn": gs.turn, "tiles": [(t.type, t.owner_id, t.army_count, t.visible, t.fog_of_war) for t in gs.board.tiles],
            "players": [(p.id, p.status, p.army_count, p.tile_count, (p.general_position.x, p.general_position.y) if p.HasField("general_position") else None,
                         p.color) for p in gs.players],
            "winner": gs.winner_id, "mask": list(gs.action_mask)}


def _rollout(fog=True):
    sizes = [(8, 8, 2), (10, 7, 3), (6, 6, 2)]
    per = [sizes[i % 3] for i in range(9)]
    army, owner, typ, ws, hs, ps = H.gen_boards(12, per, 10, 8)
    ora = O.OracleBatch(9, 10, 8, 3, fog=fog)
    ora.reset(army, owner, typ, ws, hs, ps)
    return ora, per, (army, owner, typ, ws, hs, ps)


@pytest.mark.parametrize("fog", [True, False], ids=["fog_on", "fog_off"])
def test_game_state_and_stream_update_vs_scalar_restatement(fog):
    ora, per, _ = _rollout(fog)
    deltas = fulls = 0
    for k in range(120):
        ora.step(ora.agent_actions(3, 10))
        if k % 7:
            continue
        bits = ora.legal_mask()
        for viewer in range(3):
            st, vis, fogm = _oracle_views(ora, viewer)
            for e in range(ora.B):
                w, h, P = per[e]
                if viewer >= P:
                    continue
                legal = unpack_legal_bits(bits[e, viewer], w, h)
                eng = ora.engine(e)
                gs = wire.game_state(st, vis, fogm, legal, e, viewer, game_id=f"g{e}")
                assert _proto_as_dict(gs) == _scalar_game_state(eng, viewer, eng.legal_mask(viewer))
                assert gs.board.width == w and gs.board.height == h and gs.game_id == f"g{e}"
                assert gs.current_phase == (7 if eng.game_over else 4) and gs.status == (3 if eng.game_over else 2)
                up = wire.stream_update(st, vis, fogm, legal, e, viewer)
                nc, nv = eng.L.ora_engine_changed_count(eng.e), eng.L.ora_engine_vis_changed_count(eng.e)
                want_delta = 0 < nc + nv < (w * h) // 5                              # server.go:640-644
                assert up.WhichOneof("update") == ("delta" if want_delta else "full_state")
                if want_delta:
                    deltas += 1
                    full = _scalar_game_state(eng, viewer, eng.legal_mask(viewer))
                    got = {(u.position.x, u.position.y): (u.tile.type, u.tile.owner_id, u.tile.army_count, u.tile.visible, u.tile.fog_of_war)
                           for u in up.delta.tile_updates}
                    touched = {t for t in range(w * h) if st["changed"][e, t] or st["vis_changed"][e, t]}
                    assert set(got) == {(t % w, t // w) for t in touched} and len(up.delta.tile_updates) == len(touched)
                    for (x, y), tile in got.items():
                        assert tile == full["tiles"][y * w + x]
                    assert up.delta.turn == eng.turn and [u.player_id for u in up.delta.player_updates] == list(range(P))
                    for u in up.delta.player_updates:                                # :741-752: position only when eliminated
                        assert u.state.HasField("general_position") == (not eng.alive(u.player_id) and eng.general_idx(u.player_id) >= 0)
                else:
                    fulls += 1
                    assert _proto_as_dict(up.full_state) == _scalar_game_state(eng, viewer, eng.legal_mask(viewer))
    assert deltas > 20 and fulls > 20


def test_delta_threshold_boundaries():
    """0 < |C| + |V| < N/5, integer division (server.go:636-644); a tile in both sets counts twice."""
    ora, per, _ = _rollout()
    st, vis, fog = _oracle_views(ora, 0)
    e, (w, h, P) = 0, per[0]                      # 8x8: N/5 = 12
    legal = np.zeros(w * h * 4, bool)
    for nc, nv, want in [(0, 0, "full_state"), (1, 0, "delta"), (6, 5, "delta"), (6, 6, "full_state"), (11, 0, "delta"), (12, 0, "full_state")]:
        st["changed"][e] = 0
        st["vis_changed"][e] = 0
        st["changed"][e, :nc] = 1
        st["vis_changed"][e, :nv] = 1               # overlapping tiles: counted in both sets
        up = wire.stream_update(st, vis, fog, legal, e, 0)
        assert up.WhichOneof("update") == want, (nc, nv)
        if want == "delta":
            assert len(up.delta.tile_updates) == max(nc, nv)


@pytest.mark.skipif(not HAVE_REF, reason="the reference's stubs exist in the build container only")
def test_messages_cross_parse_with_the_references_stubs():
    common_pb2, game_pb2, experience_pb2 = _ref_modules()
    from experience_stream_client import ExperienceConfig, ExperienceStreamClient
    from google.protobuf.json_format import MessageToDict
    ora, per, _ = _rollout()
    for _ in range(30):
        ora.step(ora.agent_actions(3))
    st, vis, fog = _oracle_views(ora, 1)
    bits = ora.legal_mask()
    for e in (1, 4):
        w, h, P = per[e]
        up = wire.stream_update(st, vis, fog, unpack_legal_bits(bits[e, 1], w, h), e, 1, game_id="x")
        theirs = game_pb2.GameUpdate.FromString(up.SerializeToString())
        assert MessageToDict(theirs) == MessageToDict(up)
        assert theirs.SerializeToString(deterministic=True) == up.SerializeToString(deterministic=True)
    rng = np.random.default_rng(0)
    d = {"experience_id": "id-1", "game_id": "g", "player_id": 1, "turn": 9, "state": rng.random((9, 5, 6), np.float32),
         "action": 17, "reward": float(np.float32(0.123)), "next_state": rng.random((9, 5, 6), np.float32), "done": False,
         "action_mask": rng.random(5 * 6 * 4) < 0.3}
    batch = wire.experience_batch([wire.experience(d), wire.experience(d)], batch_id=3, stream_id="s1")
    tb = experience_pb2.ExperienceBatch.FromString(batch.SerializeToString())
    assert tb.batch_id == 3 and tb.stream_id == "s1" and tb.metadata["batch_size"] == "2" and len(tb.experiences) == 2
    assert tb.experiences[0].metadata["collector_version"] == "1.0.0"
    back = ExperienceStreamClient(ExperienceConfig())._process_experience(tb.experiences[1])   # the reference's client code
    for k, v in d.items():
        if isinstance(v, np.ndarray):
            assert back[k].dtype == v.dtype and np.array_equal(back[k], v.reshape(back[k].shape)), k
        else:
            assert back[k] == v or (isinstance(v, float) and np.float32(back[k]) == np.float32(v)), k


def test_experience_batcher_32_or_100ms():
    """BatchProcessor as StreamAggregator configures it (stream_aggregator.go:64-69, batch_processor.go:33-118)."""
    now = [0.0]
    b = ExperienceBatcher(clock=lambda: now[0])
    assert b.batch_size == 32 and b.batch_timeout_s == 0.1
    out = b.add(range(70))
    assert [len(x) for x in out] == [32, 32] and out[0][0] == 0 and out[1][-1] == 63 and len(b.current) == 6
    assert b.poll() == []                       # 6 pending, no timeout yet
    now[0] = 0.099
    assert b.poll() == []
    now[0] = 0.1
    assert [list(x) for x in b.poll()] == [[64, 65, 66, 67, 68, 69]] and b.current == []
    now[0] = 5.0
    assert b.poll() == []                       # nothing pending: the ticker sends no empty batch (:109-118)
    b.add([1, 2])
    assert b.flush() == [1, 2]                  # Flush (:71-78)
    assert ExperienceBatcher(0, 0).batch_size == 32


@pytest.mark.gpu
def test_hip_states_to_protos_equal_oracle_states_to_protos():
    import generalsreinforcementlearning_amd as g
    ora, per, boards = _rollout()
    eng = g.VecEngine(9, 10, 8, 3)
    eng.reset(*boards)
    for k in range(90):
        acts = ora.agent_actions(3, 10)
        assert np.array_equal(eng.step(acts), ora.step(acts))
        if k % 9:
            continue
        hst = eng.game_state(fields=wire.STATE_FIELDS)
        hbits = eng.legal_action_mask_bits()
        for viewer in range(3):
            ost, ovis, ofog = _oracle_views(ora, viewer)
            hvis, hfog = eng.compute_player_visibility(viewer)
            for e in range(9):
                w, h, P = per[e]
                if viewer >= P:
                    continue
                legal = unpack_legal_bits(hbits[e, viewer], w, h)
                a = wire.stream_update(hst, hvis, hfog, legal, e, viewer, game_id="g")
                b = wire.stream_update(ost, ovis, ofog, unpack_legal_bits(ora.legal_mask()[e, viewer], w, h), e, viewer, game_id="g")
                a.ClearField("timestamp")
                b.ClearField("timestamp")
                # GeneralIdx with >= 2 generals is "some listed general" on both sides: compare modulo that field
                for m in (a, b):
                    ps = m.delta.player_updates if m.WhichOneof("update") == "delta" else m.full_state.players
                    for p in ps:
                        s = p.state if m.WhichOneof("update") == "delta" else p
                        if s.HasField("general_position"):
                            s.general_position.x, s.general_position.y = 0, 0
                assert a == b, (k, e, viewer)


@pytest.mark.gpu
@pytest.mark.parametrize("fog", [True, False], ids=["fog_on", "fog_off"])
def test_device_stream_deltas_equal_create_stream_update(fog):
    """gvec_stream_deltas (the delta-vs-full decision of server.go:636-644 and the delta's tile updates, built on the
    device) against wire.stream_update (the host restatement this file pins to the scalar createStreamUpdate): per env and
    viewer the same decision, and for deltas the same GameUpdate message - over a padded mixed batch, every turn, with
    aborted turns, eliminations and turns where nothing changed; and through a sharded handle."""
    import generalsreinforcementlearning_amd as g
    import _harness as H
    B = 60
    per = [[(10, 8, 3), (7, 7, 2), (12, 12, 4), (20, 20, 4)][i % 4] for i in range(B)]
    army, owner, typ, ws, hs, ps = H.gen_boards(17, per, 20, 20)
    eng = g.VecEngine(B, 20, 20, 4, fog_of_war=fog)
    many = g.VecEngine(B, 20, 20, 4, fog_of_war=fog, devices=[0, 0, 0])
    for e_ in (eng, many):
        e_.reset(army, owner, typ, ws, hs, ps)
    seen = {1: 0, 2: 0}
    for k in range(70):
        acts = eng.agent_actions(5, 15)
        eng.step(acts)
        many.step(acts)
        st = eng.game_state(fields=wire.STATE_FIELDS)
        for viewer in (0, 1, 3):
            kind, count, upd = eng.stream_deltas(viewer)
            pk, poff, pupd = eng.stream_deltas_packed(viewer)         # the same updates as one stream
            assert np.array_equal(pk, kind) and np.array_equal(np.diff(poff), count) and poff[0] == 0
            assert np.array_equal(pupd, np.concatenate([upd[e_, : count[e_]] for e_ in range(B)]))
            fk, foff, fupd = eng.stream_deltas_packed(viewer, full_tiles=True)   # kind-2 envs carry their whole board
            assert np.array_equal(fk, kind)
            k2, c2, u2 = many.stream_deltas(viewer)
            assert np.array_equal(kind, k2) and np.array_equal(count, c2)
            vis, fg = eng.compute_player_visibility(viewer)
            for e in range(B):
                w, h, P = per[e]
                if viewer >= P:
                    continue
                assert np.array_equal(upd[e, : count[e]], u2[e, : count[e]])
                want = wire.stream_update(st, vis, fg, np.zeros(w * h * 4, bool), e, viewer)
                got = wire.stream_update_from_delta(st, kind, count, upd, e, viewer)
                seen[int(kind[e])] += 1
                mine = fupd[foff[e]: foff[e + 1]]
                if want.WhichOneof("update") == "full_state":
                    assert kind[e] == 2 and got is None and count[e] == 0, (k, e, viewer)
                    assert len(mine) == w * h
                    fs = wire.full_state_from_tiles(st, mine, np.zeros(w * h * 4, bool), e, viewer)
                    assert fs == want.full_state, (k, e, viewer)
                else:
                    assert np.array_equal(mine, upd[e, : count[e]])
                    assert kind[e] == 1 and got is not None, (k, e, viewer)
                    want.ClearField("timestamp")
                    got.ClearField("timestamp")
                    if got != want:
                        a_ = [(u.position.x, u.position.y, u.tile.type, u.tile.owner_id, u.tile.army_count, u.tile.visible, u.tile.fog_of_war) for u in got.delta.tile_updates]
                        b_ = [(u.position.x, u.position.y, u.tile.type, u.tile.owner_id, u.tile.army_count, u.tile.visible, u.tile.fog_of_war) for u in want.delta.tile_updates]
                        raise AssertionError((k, e, viewer, [x for x in zip(a_, b_) if x[0] != x[1]][:4], len(a_), len(b_),
                                              str(got.delta.player_updates) == str(want.delta.player_updates)))
    assert seen[1] > 1000 and seen[2] > 50
    eng.close()
    many.close()


@pytest.mark.gpu
@pytest.mark.parametrize("maxp", [2, 4, 8])
@pytest.mark.parametrize("slots,parity", sorted(__import__("_harness").VARIANT_DIMS), ids=__import__("_harness").VARIANT_IDS)
def test_device_stream_deltas_on_every_kernel_variant(maxp, slots, parity):
    """The same comparison on the board limits that select each compiled <players, slots, parity> instantiation of
    stream_delta_kernel: a few turns around the growth turn (deltas, then full states), first and last viewer."""
    import generalsreinforcementlearning_amd as g
    import _harness as H
    B = 6
    mw, mh, per = H.variant_batch(maxp, slots, parity, B)
    army, owner, typ, ws, hs, ps = H.gen_boards(300 + slots, per, mw, mh)
    # player 0 holds every second plain tile: the growth turn then touches more than a fifth of the board (a full state)
    t = np.arange(mw * mh)[None, :]
    held = (typ == 0) & (owner < 0) & (t % 2 == 0) & (t < (ws * hs)[:, None])
    owner[held], army[held] = 0, 2
    eng = g.VecEngine(B, mw, mh, maxp, fog_of_war=True)
    eng.reset(army, owner, typ, ws, hs, ps)
    eng.write_state({"turn": np.full(B, 21, np.int32)})
    seen = {1: 0, 2: 0}
    for k in range(6):
        eng.step(eng.agent_actions(9, 10))
        st = eng.game_state(fields=wire.STATE_FIELDS)
        for viewer in (0, maxp - 1):
            kind, count, upd = eng.stream_deltas(viewer)
            fk, foff, fupd = eng.stream_deltas_packed(viewer, full_tiles=True)
            assert np.array_equal(fk, kind)
            vis, fg = eng.compute_player_visibility(viewer)
            for e in range(B):
                w, h, P = per[e]
                if viewer >= P:
                    continue
                want = wire.stream_update(st, vis, fg, np.zeros(w * h * 4, bool), e, viewer)
                got = wire.stream_update_from_delta(st, kind, count, upd, e, viewer)
                mine = fupd[foff[e]: foff[e + 1]]
                seen[int(kind[e])] += 1
                if want.WhichOneof("update") == "full_state":
                    assert kind[e] == 2 and got is None and count[e] == 0 and len(mine) == w * h, (k, e, viewer)
                    assert wire.full_state_from_tiles(st, mine, np.zeros(w * h * 4, bool), e, viewer) == want.full_state, (k, e, viewer)
                else:
                    assert kind[e] == 1 and got is not None and np.array_equal(mine, upd[e, : count[e]]), (k, e, viewer)
                    want.ClearField("timestamp")
                    got.ClearField("timestamp")
                    assert got == want, (k, e, viewer)
    assert seen[1] > 0 and seen[2] > 0
    eng.close()
