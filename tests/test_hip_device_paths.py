"""GPU tests of the zero-copy (GVEC_MEM_DEVICE) entry points, partial resets and degenerate boards."""
import ctypes as C

import numpy as np
import pytest

import _harness as H
import _oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import generalsreinforcementlearning_amd as g
    g.load()
    return g


def test_device_pointer_step_and_readback(g):
    import torch
    from generalsreinforcementlearning_amd._lib import StateView, check
    B, w, h, P = 512, 15, 15, 2
    sizes = [(w, h, P)] * B
    army, owner, typ, ws, hs, ps = H.gen_boards(12, sizes, w, h)
    host = g.VecEngine(B, w, h, P)
    dev = g.VecEngine(B, w, h, P, stream=torch.cuda.current_stream().cuda_stream)
    for e in (host, dev):
        e.reset(army, owner, typ, ws, hs, ps)
    ora = O.OracleBatch(B, w, h, P)
    ora.reset(army, owner, typ, ws, hs, ps)
    d_err = torch.zeros(B, dtype=torch.int32, device="cuda")
    d_bits = torch.zeros(B * P * dev.mask_bytes, dtype=torch.uint8, device="cuda")
    for k in range(60):
        acts = ora.agent_actions(2, 10)
        oerr, obits = ora.step(acts, want_mask=True)
        herr, hbits = host.step(acts, want_mask=True)
        d_acts = torch.from_numpy(acts.view(np.uint8).reshape(-1).copy()).cuda()
        dev.step_device(d_acts, d_err, d_bits)  # enqueued on torch's stream, no host copies inside the library
        assert np.array_equal(d_err.cpu().numpy(), oerr) and np.array_equal(herr, oerr)
        assert np.array_equal(d_bits.cpu().numpy().reshape(B, P, -1), obits) and np.array_equal(hbits, obits)
    # read_state straight into torch tensors
    t_army = torch.empty(B * w * h, dtype=torch.int32, device="cuda")
    t_owner = torch.empty(B * w * h, dtype=torch.int8, device="cuda")
    t_turn = torch.empty(B, dtype=torch.int32, device="cuda")
    v = StateView()
    v.army, v.owner, v.turn = t_army.data_ptr(), t_owner.data_ptr(), t_turn.data_ptr()
    check(dev.L.gvec_read_state(dev.h, 0, B, C.byref(v), 1), "gvec_read_state")
    torch.cuda.synchronize()
    st = ora.read_state()
    assert np.array_equal(t_army.cpu().numpy().reshape(B, -1), st["army"]) and np.array_equal(t_owner.cpu().numpy().reshape(B, -1), st["owner"])
    assert np.array_equal(t_turn.cpu().numpy(), st["turn"])
    # zero-copy view of the resident legal-mask buffer
    assert dev.device_buffer(3) not in (None, 0)


def test_mask_buffer_of_the_caller_may_sit_at_any_dword(g):
    """gvec_step with device pointers writes the masks into the caller's buffer; the kernel's wide (16-byte) stores
    must not assume more than the 4-byte alignment a uint32 array has."""
    import torch
    B, w, h, P = 64, 20, 20, 4
    army, owner, typ, ws, hs, ps = H.gen_boards(3, [(w, h, P)] * B, w, h)
    ora = O.OracleBatch(B, w, h, P)
    ora.reset(army, owner, typ, ws, hs, ps)
    engs = []
    for off in (0, 4, 8, 12):
        e = g.VecEngine(B, w, h, P, stream=torch.cuda.current_stream().cuda_stream)
        e.reset(army, owner, typ, ws, hs, ps)
        raw = torch.zeros(B * P * e.mask_bytes + 64, dtype=torch.uint8, device="cuda")
        engs.append((e, raw, off))
    for k in range(12):
        acts = ora.agent_actions(6, 20)
        oerr, obits = ora.step(acts, want_mask=True)
        d_acts = torch.from_numpy(acts.view(np.uint8).reshape(-1).copy()).cuda()
        for e, raw, off in engs:
            assert raw.data_ptr() % 16 == 0
            e.step_device(d_acts, None, raw.data_ptr() + off)
            got = raw[off:off + B * P * e.mask_bytes].cpu().numpy().reshape(obits.shape)
            assert np.array_equal(got, obits), (k, off)
            assert not raw[:off].any() and not raw[off + B * P * e.mask_bytes:].any(), "nothing outside the buffer is written"


def test_partial_reset_by_env_ids(g):
    B, w, h, P = 64, 10, 10, 2
    sizes = [(w, h, P)] * B
    army, owner, typ, ws, hs, ps = H.gen_boards(4, sizes, w, h)
    eng = g.VecEngine(B, w, h, P)
    ora = O.OracleBatch(B, w, h, P)
    eng.reset(army, owner, typ, ws, hs, ps)
    ora.reset(army, owner, typ, ws, hs, ps)
    H.run_lockstep(eng, ora, 30, seed=1, check_every=30, want_mask=False, ctx="before partial reset")
    ids = np.array([3, 17, 40, 63], np.int32)
    a2, o2, t2, w2, h2, p2 = H.gen_boards(99, [(w, h, P)] * 4, w, h)
    eng.reset(a2, o2, t2, w2, h2, p2, env_ids=ids)
    ora.reset(a2, o2, t2, w2, h2, p2, env_ids=ids)
    st = eng.game_state()
    assert (st["turn"][ids] == 0).all() and (np.delete(st["turn"], ids) == 30).all()
    H.assert_states_equal(st, ora.read_state(), "after partial reset")
    H.run_lockstep(eng, ora, 30, seed=2, check_every=10, ctx="after partial reset")


def _tiles(w, h, spec):
    army = np.zeros(w * h, np.int32); owner = np.full(w * h, -1, np.int8); typ = np.zeros(w * h, np.uint8)
    for (x, y, o, a, t) in spec:
        i = y * w + x
        owner[i], army[i], typ[i] = o, a, t
    return army, owner, typ


@pytest.mark.parametrize("w,h,P,spec", [
    (1, 1, 1, [(0, 0, 0, 3, 1)]),
    (1, 6, 2, [(0, 0, 0, 5, 1), (0, 5, 1, 5, 1), (0, 2, -1, 2, 2)]),
    (6, 1, 2, [(0, 0, 0, 5, 1), (5, 0, 1, 5, 1), (3, 0, -1, 0, 3)]),
    (2, 2, 2, [(0, 0, 0, 4, 1), (1, 1, 1, 4, 1)]),
    (32, 1, 2, [(0, 0, 0, 9, 1), (31, 0, 1, 9, 1)]),
    (1, 32, 3, [(0, 0, 0, 9, 1), (0, 31, 1, 9, 1), (0, 16, 2, 9, 1)]),
    (32, 32, 2, [(0, 0, 0, 50, 1), (31, 31, 1, 50, 1), (16, 16, -1, 40, 2)]),
], ids=["1x1", "1x6", "6x1", "2x2", "32x1", "1x32", "32x32"])
def test_degenerate_board_shapes(g, w, h, P, spec):
    B = 16
    a, o, t = _tiles(w, h, spec)
    army, owner, typ = np.tile(a, (B, 1)), np.tile(o, (B, 1)), np.tile(t, (B, 1))
    eng = g.VecEngine(B, w, h, P)
    ora = O.OracleBatch(B, w, h, P)
    eng.reset(army, owner, typ)
    ora.reset(army, owner, typ, [w] * B, [h] * B, [P] * B)
    H.assert_states_equal(eng.game_state(), ora.read_state(), "reset")
    assert np.array_equal(eng.legal_action_mask_bits(), ora.legal_mask())
    H.run_lockstep(eng, ora, 120, seed=6, invalid_permille=30, check_every=1, ctx=f"{w}x{h}")
    for p in range(P):
        hv, hf = eng.compute_player_visibility(p)
        for e in range(0, B, 5):
            ov, of = ora.engine(e).player_visibility(p)
            assert np.array_equal(hv[e, :w * h], ov.astype(bool)) and np.array_equal(hf[e, :w * h], of.astype(bool))


def test_empty_ranges_and_zero_turn_rollout(g):
    eng = g.VecEngine(8, 5, 5, 2)
    st = eng.game_state(3, 0)  # n = 0
    assert all(len(v) == 0 for v in st.values())
    assert eng.rollout(0, 1)["env_steps"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("mix", [(0, 0), (32768, 65536), (65536, 0), (45875, 19661)],
                         ids=["always_full_move", "half_always", "never_moves", "go_demo_helper_rates"])
def test_agent_mix_matches_oracle(g, mix):
    """gvec_set_agent_mix; (45875, 19661) are the rates of game.GenerateRandomActions (demo_helpers.go:20,44)."""
    B, w, h, p = 128, 12, 12, 3
    army, owner, typ, ww, hh, pp = H.gen_boards(77, [(w, h, p)] * B, w, h)
    eng = g.VecEngine(B, w, h, p)
    ora = O.OracleBatch(B, w, h, p)
    for e in (eng, ora):
        e.reset(army, owner, typ, ww, hh, pp)
        e.set_agent_mix(*mix)
    for k in range(40):
        ha, oa = eng.agent_actions(5), ora.agent_actions(5)
        assert np.array_equal(ha, oa), f"turn {k}"
        flags = ha["flags"]
        if mix == (0, 0):
            st = ora.read_state()
            has_move = ora.legal_mask().reshape(B, p, -1).any(-1)
            assert np.array_equal((flags & 1).astype(bool), has_move & st["alive"].astype(bool) & ~st["done"].astype(bool)[:, None])
            assert not (flags & 2).any()          # never a half move
        if mix == (65536, 0):
            assert not flags.any()
        if mix == (32768, 65536):
            assert ((flags & 1) == 0).any() and (flags[(flags & 1) == 1] & 2).all()
        assert np.array_equal(eng.step(ha), ora.step(oa))
    st = eng.rollout(60, 9, 0, fused=True)
    assert st["env_steps"] == ora.rollout(60, 9, 0)
    H.assert_states_equal(eng.game_state(), ora.read_state(), f"agent mix {mix}")
    with pytest.raises(g.GvecError):
        eng.set_agent_mix(-1, 0)


@pytest.mark.gpu
def test_failed_creation_reports_and_releases(g):
    """A handle that cannot be allocated returns a HIP error (no partial handle, nothing left allocated)."""
    import torch
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        with pytest.raises(g.GvecError) as ei:
            g.VecEngine(400_000_000, 32, 32, 8)   # ~ 5 TB of board state
        assert ei.value.code == -3  # GVEC_E_HIP
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (256 << 20)
    eng = g.VecEngine(4, 5, 5, 2)   # the library is still usable
    eng.reset_generated(1)
    assert eng.game_state()["turn"].tolist() == [0, 0, 0, 0]


@pytest.mark.gpu
def test_device_side_env_ids_are_range_checked(g):
    """env_ids handed over in DEVICE memory cannot be checked by the host: the import kernel rejects ids outside
    [0, num_envs) instead of writing out of bounds, and the valid envs of the call are still reset."""
    import torch
    from generalsreinforcementlearning_amd._lib import check
    B, w, h, P = 16, 8, 8, 2
    army, owner, typ, ws, hs, ps = H.gen_boards(21, [(w, h, P)] * B, w, h)
    eng = g.VecEngine(B, w, h, P, stream=torch.cuda.current_stream().cuda_stream)
    eng.reset(army, owner, typ, ws, hs, ps)
    eng.rollout(7, 3, 0)
    assert (eng.game_state()["turn"] == 7).all()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    n = 3
    ids = t(np.array([2, B + 5, 9], np.int32))          # the middle id is out of range
    d = [t(army[:n]), t(owner[:n]), t(typ[:n]), t(ws[:n]), t(hs[:n]), t(ps[:n])]
    rc = eng.L.gvec_reset(eng.h, ids.data_ptr(), n, *[x.data_ptr() for x in d], 1)
    assert rc == -4  # GVEC_E_RANGE
    assert b"env id out of range" in eng.L.gvec_last_error()
    turn = eng.game_state()["turn"]
    assert turn[2] == 0 and turn[9] == 0 and (np.delete(turn, [2, 9]) == 7).all()
    ids_ok = t(np.array([1, 4, 15], np.int32))
    check(eng.L.gvec_reset(eng.h, ids_ok.data_ptr(), n, *[x.data_ptr() for x in d], 1), "gvec_reset")
    assert (eng.game_state()["turn"][[1, 4, 15]] == 0).all()


def test_pinned_host_buffers_give_the_same_results(g):
    """gvec_host_alloc / VecEngine.pinned / step(pinned=True): page-locked buffers are only a faster road for the same bytes."""
    B = 512
    a, b = g.VecEngine(B, 12, 12, 3), g.VecEngine(B, 12, 12, 3)
    for e in (a, b):
        e.reset_generated(4)
    pacts = b.pinned((B, 3), g.ACTION_DTYPE)
    for k in range(25):
        acts = a.agent_actions(9, 8)
        pacts[...] = acts
        e1, m1 = a.step(acts, want_mask=True)
        e2, m2 = b.step(pacts, want_mask=True, pinned=True)
        assert np.array_equal(e1, e2) and np.array_equal(m1, m2), k
        assert np.array_equal(a.step(acts), b.step(pacts, pinned=True))
    sa, sb = a.game_state(), b.game_state()
    assert all(np.array_equal(sa[f], sb[f]) for f in sa)
    b.close()
    a.close()


def test_rollout_range_slices_equal_one_full_turn(g):
    """gvec_rollout_range: stepping every env exactly once, slice by slice (also on different streams, overlapping), is one
    gvec_rollout turn - agent keys, pool re-deals, masks, recorded moves and error codes included."""
    import torch
    B = 3000
    a = g.VecEngine(B, 12, 12, 3, auto_reset=True, stream=torch.cuda.current_stream().cuda_stream)
    b = g.VecEngine(B, 12, 12, 3, auto_reset=True, stream=torch.cuda.current_stream().cuda_stream)
    for e in (a, b):
        e.reset_generated(8)
        e.build_board_pool(16, 3)
        e.record_agent_actions(True)
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    cuts = [0, 1, 640, 641, 1999, B]
    for k in range(160):
        a.rollout(1, 5, 9, fused=False, want_stats=False)
        # b: the slices alternate between two streams; both wait for the previous turn, the next turn waits for both
        side.wait_stream(main)
        for i in range(len(cuts) - 1):
            b.set_stream((side if i % 2 else main).cuda_stream)
            b.rollout_range(cuts[i], cuts[i + 1] - cuts[i], 1, 5, 9)
        b.set_stream(main.cuda_stream)
        main.wait_stream(side)
        if k % 20 == 19:
            sa, sb = a.game_state(), b.game_state()
            assert all(np.array_equal(sa[f], sb[f]) for f in sa), k
            assert np.array_equal(a.legal_action_mask_bits(), b.legal_action_mask_bits())
            assert np.array_equal(a.recorded_actions(), b.recorded_actions()) and np.array_equal(a.last_errors(), b.last_errors())
    assert a.counters() == b.counters() and a.counters()["games_finished"] > 0
    with pytest.raises(g.GvecError):
        b.rollout_range(2990, 20, 1, 5)
    a.close(); b.close()
