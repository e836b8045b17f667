"""GPU tests of the two army storage forms (gvec_device.hpp "army storage"): NARROW u16 pairs and the
exact int32 escape.  Tile.Army is a Go int (core/board.go:9): nothing may be clamped or wrapped at
the 16-bit boundary.  Expected values come from the oracle (int64 armies, core/movement.go:40-86
restated line by line); every board is kept far below 2^31 so int32 never wraps either."""
import numpy as np
import pytest

import _harness as H
import _oracle as O

pytestmark = pytest.mark.gpu

HF_WIDE = 4


@pytest.fixture(scope="module")
def g():
    import generalsreinforcementlearning_amd as g
    g.load()
    return g


class _RawDeviceArray:
    """Zero-copy view of a raw device pointer for torch.as_tensor (the __cuda_array_interface__ protocol)."""

    def __init__(self, ptr, n_u32):
        self.__cuda_array_interface__ = {"shape": (n_u32,), "typestr": "<u4", "data": (int(ptr), False), "version": 2}


def _header_flags(eng):
    """Header dword H_DIMS >> 24 of every env, read through the zero-copy header buffer (GVEC_BUF_HEADER)."""
    import torch
    eng.synchronize()
    t = torch.as_tensor(_RawDeviceArray(eng.device_buffer(0), eng.B * 24), device="cuda")
    h = t.cpu().numpy().view(np.uint32).reshape(eng.B, 24)
    return h[:, 1] >> 24


@pytest.mark.parametrize("w,h,P", [(6, 6, 2), (10, 10, 2), (20, 20, 4), (25, 25, 4), (32, 32, 8)],
                         ids=["6x6_p2", "10x10_p2", "20x20_p4", "25x25_p4", "32x32_p8"])
def test_lockstep_across_the_16_bit_boundary(g, w, h, P):
    """Armies planted at 32,767 / 65,534..65,537 / 131,071 / 2^24 +- 1 / 10^9 on owned tiles: moves, combat,
    production and the per-player sums must carry them exactly while envs cross between the two forms."""
    B = 96
    army, owner, typ, ws, hs, ps = H.gen_boards(5, [(w, h, P)] * B, w, h)
    vals = [32767, 32768, 65534, 65535, 65536, 65537, 131071, (1 << 24) - 1, (1 << 24) + 1, 10 ** 9]
    rng = np.random.default_rng(7)
    for e in range(B):
        gens = np.flatnonzero(typ[e] == 1)
        cities = np.flatnonzero(typ[e] == 2)
        if e % 4 != 3:                                   # every 4th env stays small: mixed forms in one launch
            for i, t in enumerate(gens):
                army[e, t] = vals[(e + i) % len(vals)]
            if len(cities) and e % 2 == 0:
                army[e, cities[0]] = vals[(e // 2) % len(vals)] + int(rng.integers(0, 3))
    eng = g.VecEngine(B, w, h, P)
    ora = O.OracleBatch(B, w, h, P)
    eng.reset(army, owner, typ, ws, hs, ps)
    ora.reset(army, owner, typ, ws, hs, ps)
    H.assert_states_equal(eng.game_state(), ora.read_state(), "reset")
    fl = _header_flags(eng)
    big = (army > 65535).any(1)
    assert np.array_equal((fl & HF_WIDE) != 0, big), "an env is WIDE exactly when one of its armies exceeds 65,535"
    saw = set()
    for k in range(6):
        H.run_lockstep(eng, ora, 25, seed=11 + k, invalid_permille=5, check_every=1, ctx=f"{w}x{h} block {k}")
        st = ora.read_state()
        fl = _header_flags(eng)
        big = (st["army"].astype(np.int64) > 65535).any(1)
        assert np.array_equal((fl & HF_WIDE) != 0, big)
        saw |= {bool(b) for b in big}
    assert saw == {True, False}
    # the fused rollout keeps boards in registers for many turns: the same exactness
    stats = eng.rollout(64, seed=3, fused=True)
    assert stats["env_steps"] == ora.rollout(64, 3, 0)
    H.assert_states_equal(eng.game_state(), ora.read_state(), "fused rollout over wide envs")


def test_env_returns_to_narrow_when_it_fits_again(g):
    w = h = 5
    tiles = [dict(x=0, y=0, owner=0, army=70000, type=1), dict(x=1, y=0, owner=1, army=69999, type=0),
             dict(x=4, y=4, owner=1, army=5, type=1)]
    army, owner, typ = O.planes_from_tiles(w, h, tiles)
    eng = g.VecEngine(1, w, h, 2, production=(0, 0, 0))
    ora = O.OracleBatch(1, w, h, 2, prod=(0, 0, 0))
    eng.reset(army[None], owner[None], typ[None])
    ora.reset(army[None], owner[None], typ[None], [w], [h], [2])
    assert _header_flags(eng)[0] & HF_WIDE
    acts = g.make_actions(1, 2, [(0, 0, 0, 0, 1, 0, True)])      # 69,999 attack 69,999: tie, defender keeps 0 (movement.go:85)
    assert eng.step(acts)[0] == ora.step(acts)[0] == 0
    st = eng.game_state()
    H.assert_states_equal(st, ora.read_state(), "after the attack")
    assert st["army"][0, 0] == 1 and st["army"][0, 1] == 0 and st["owner"][0, 1] == 1
    assert not (_header_flags(eng)[0] & HF_WIDE), "every army fits 16 bits again: back to the narrow form"


def test_negative_and_huge_armies_survive_state_round_trip(g):
    """gvec_write_state may poke any int32 (the reference's tests write e.gs.* directly): stored exactly."""
    B, w, h = 4, 7, 7
    eng = g.VecEngine(B, w, h, 2)
    eng.reset_generated(3)
    st = eng.game_state()
    a = st["army"].copy()
    a[0, 3] = -5
    a[1, 4] = 2 ** 31 - 1
    a[2, 5] = -2 ** 31
    eng.write_state({"army": a})
    assert np.array_equal(eng.game_state(fields=("army",))["army"], a)
    fl = _header_flags(eng)
    assert list((fl & HF_WIDE) != 0) == [True, True, True, False]


def test_record_slab_carries_wide_envs_and_rejects_foreign_headers(g):
    import torch
    B, w, h, P = 32, 12, 12, 3
    army, owner, typ, ws, hs, ps = H.gen_boards(9, [(w, h, P)] * B, w, h)
    army[::2][typ[::2] == 1] = 100000
    a = g.VecEngine(B, w, h, P)
    b = g.VecEngine(B, w, h, P)
    a.reset(army, owner, typ, ws, hs, ps)
    a.rollout(30, seed=2)
    rec = a.state_bytes_per_env()
    buf = torch.zeros(B * rec, dtype=torch.uint8, device="cuda")
    a.export_records(buf.data_ptr())
    a.synchronize()
    b.import_records(buf.data_ptr())
    sa, sb = a.game_state(), b.game_state()
    for f in sa:
        assert np.array_equal(sa[f], sb[f]), f
    # game over / fog / wide armies travel; the turn engine's bookkeeping flags (HF_SYNC, HF_VSMALL) restart on import
    assert np.array_equal(_header_flags(a) & 7, _header_flags(b) & 7)
    assert np.array_equal(a.legal_action_mask_bits(), b.legal_action_mask_bits())
    # a slab from an engine with other limits (or a corrupted one) is refused on the device, env by env
    hdr = buf[: B * 96].view(torch.int32).reshape(B, 24)
    bad = hdr.clone()
    bad[3, 1] = (40 | (12 << 8) | (3 << 16))        # W = 40 > max_width
    bad[5, 1] = (12 | (12 << 8) | (7 << 16))        # P = 7 > max_players
    bad[7, 20] = 1                                   # reciprocal of W does not match
    buf2 = buf.clone()
    buf2[: B * 96] = bad.reshape(-1).view(torch.uint8)
    c = g.VecEngine(B, w, h, P)
    c.reset_generated(1)
    before = c.game_state()
    with pytest.raises(g.GvecError) as ei:
        c.import_records(buf2.data_ptr())
    assert ei.value.code == -5  # GVEC_E_BOARD
    after = c.game_state()
    for e in (3, 5, 7):
        for f in ("army", "owner", "turn"):
            assert np.array_equal(before[f][e], after[f][e]), (e, f)
    ok = [e for e in range(B) if e not in (3, 5, 7)]
    assert np.array_equal(after["army"][ok], sa["army"][ok])
    d = g.VecEngine(B, 10, 10, P)                    # smaller limits: the 12x12 records do not fit
    small = torch.zeros(B * d.state_bytes_per_env(), dtype=torch.uint8, device="cuda")
    d.export_records(small.data_ptr())
    sh = small[: B * 96].view(torch.int32).reshape(B, 24)
    sh[:, 1] = (12 | (12 << 8) | (3 << 16))
    with pytest.raises(g.GvecError):
        d.import_records(small.data_ptr())


def test_reset_env_flag_redeals_from_the_pool(g):
    """GVEC_ACT_RESET_ENV: the caller ends an episode (truncation) - the env is re-dealt in that step."""
    B, w, h, P = 64, 8, 8, 2
    army, owner, typ, ws, hs, ps = H.gen_boards(2, [(w, h, P)] * B, w, h)
    eng = g.VecEngine(B, w, h, P, auto_reset=True)
    ora = O.OracleBatch(B, w, h, P)
    eng.reset(army, owner, typ, ws, hs, ps)
    ora.reset(army, owner, typ, ws, hs, ps)
    eng.build_board_pool(11, 5)
    ora.set_pool(11, 5)
    for k in range(40):
        acts = ora.agent_actions(3)
        if k % 7 == 3:
            acts["flags"][k % B::5, 0] |= 8
        assert np.array_equal(eng.step(acts), ora.step(acts))
        H.assert_states_equal(eng.game_state(), ora.read_state(), f"turn {k}")
    st = eng.game_state()
    assert (st["turn"] < 40).sum() >= 10
    # without a pool the flag is ignored: the turn is played
    e2, o2 = g.VecEngine(4, w, h, P), O.OracleBatch(4, w, h, P)
    e2.reset(army[:4], owner[:4], typ[:4], ws[:4], hs[:4], ps[:4])
    o2.reset(army[:4], owner[:4], typ[:4], ws[:4], hs[:4], ps[:4])
    acts = o2.agent_actions(1)
    acts["flags"][:, 0] |= 8
    assert np.array_equal(e2.step(acts), o2.step(acts))
    H.assert_states_equal(e2.game_state(), o2.read_state(), "no pool")


def test_config1_single_game_1000_turns(g):
    """BASELINE.json configs[0]: ONE 10x10 two-player game, random agents, 1,000 turns, through the ABI,
    every turn compared with the oracle (err, legal masks, full state)."""
    w = h = 10
    army, owner, typ, ws, hs, ps = H.gen_boards(1, [(w, h, 2)], w, h)
    eng = g.VecEngine(1, w, h, 2)
    ora = O.OracleBatch(1, w, h, 2)
    eng.reset(army, owner, typ, ws, hs, ps)
    ora.reset(army, owner, typ, ws, hs, ps)
    done_at = None
    for k in range(1000):
        acts = ora.agent_actions(1)
        oerr, obits = ora.step(acts, want_mask=True)
        herr, hbits = eng.step(acts, want_mask=True)
        assert np.array_equal(herr, oerr) and np.array_equal(hbits, obits), k
        H.assert_states_equal(eng.game_state(), ora.read_state(), f"turn {k}")
        if oerr[0] == 5 and done_at is None:
            done_at = k
    assert ora.read_state()["turn"][0] == (1000 if done_at is None else done_at)
