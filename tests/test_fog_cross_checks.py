"""Second and third opinions on the fog of war (the reference holds NO asserting fog test, so
the optimized restatement in oracle/ is otherwise pinned by nothing but itself).

1. The reference has a legacy twin of the fog update (internal/game/visibility.go:19-144, selected by
   features.use_optimized_visibility = false).  It is restated independently in oracle/ (Tile.SetVisible per
   player and tile instead of bit masks) and must leave the SAME VisibleBitfield as the optimized restatement
   after every turn of long lock-step rollouts - including aborted turns, list desync and eliminations.
   Where the two paths differ in the reference: only DiscoveredBitfield (the legacy path sets it through
   Tile.SetVisible, core/board.go:57-60; the optimized path never does - H9).  Checked too: what one legacy update
   discovers is a subset of what it leaves visible (it clears first, then sets).
2. A property that needs no restatement at all: while an env's OwnedTiles lists match the board (no aborted turn
   so far, H5/H6), full and incremental updates coincide and every ALIVE player's visibility after Step(t+1) is the
   3x3 dilation (visibility_optimized.go:9-13) of its list at the end of Step(t) (H1: fog lags one turn)."""
import numpy as np
import pytest

import _harness as H
import _oracle as O


def _dilate3(m, w, h):
    g = m.reshape(h, w)
    out = np.zeros((h + 2, w + 2), bool)
    for dy in range(3):
        for dx in range(3):
            out[dy:dy + h, dx:dx + w] |= g
    return out[1:-1, 1:-1].reshape(-1)


@pytest.mark.parametrize("sizes,invalid", [([(10, 10, 2)], 0), ([(15, 15, 3), (12, 9, 4)], 20), ([(20, 20, 4), (7, 7, 2), (6, 6, 3)], 8)],
                         ids=["10x10_p2", "15x15_p3_mixed_invalid", "20x20_p4_mixed"])
def test_legacy_fog_twin_and_dilation_property(sizes, invalid):
    B = 24
    per_env = [sizes[i % len(sizes)] for i in range(B)]
    mw, mh, mp = max(s[0] for s in per_env), max(s[1] for s in per_env), max(s[2] for s in per_env)
    army, owner, typ, ws, hs, ps = H.gen_boards(31, per_env, mw, mh)
    ora = O.OracleBatch(B, mw, mh, mp)
    ora.reset(army, owner, typ, ws, hs, ps)
    clean = np.ones(B, bool)            # no aborted turn so far: lists match the board
    checked = desynced = 0
    discovered_some = False
    for k in range(220):
        st = ora.read_state()
        legacy_vis, legacy_disc = ora.next_fog_legacy()
        acts = ora.agent_actions(17, invalid)
        err = ora.step(acts)
        now = ora.read_state()
        live = st["done"] == 0
        # 1. legacy twin == optimized restatement, on every live env, desynced or not
        assert np.array_equal(now["visible"][live], legacy_vis[live]), f"turn {k}: legacy and optimized fog differ"
        # H9: what this one legacy update discovered = the bits it set through Tile.SetVisible(true) (board.go:57-60):
        # never a bit that ends up invisible; the optimized path discovers nothing at all
        assert (legacy_disc[live] & ~legacy_vis[live] == 0).all()
        discovered_some = discovered_some or bool(legacy_disc[live].any())
        # 2. dilation property on envs whose lists match the board
        for e in np.flatnonzero(live):
            w, h, P = per_env[e]
            n = w * h
            if not (clean[e] and np.array_equal(st["listed"][e, :n], st["owner"][e, :n])):
                desynced += 1
                continue
            for p in range(P):
                if not st["alive"][e, p]:
                    continue
                want = _dilate3(st["listed"][e, :n] == p, w, h)
                got = (now["visible"][e, :n] >> p) & 1
                assert np.array_equal(got.astype(bool), want), (k, e, p)
                checked += 1
        clean &= (err == 0) | ~live
    assert checked > 1000 and discovered_some
    assert desynced > 0 or invalid == 0   # the desync paths were exercised by the twin comparison
