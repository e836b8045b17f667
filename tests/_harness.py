"""Shared helpers for parity tests: board batches, state comparison."""
import numpy as np

import _oracle as O

TILE_FIELDS = ("army", "owner", "type", "visible", "listed", "changed", "vis_changed")
ENV_FIELDS = ("turn", "done", "winner", "width", "height", "players")
PLAYER_FIELDS = ("alive", "army_count", "tile_count")


def gen_boards(seed, sizes, max_w, max_h):
    """sizes: list of (w, h, p) per env -> planes [n][max_w*max_h] from the oracle's map generator."""
    n, stride = len(sizes), max_w * max_h
    army = np.zeros((n, stride), np.int32)
    owner = np.full((n, stride), -1, np.int8)
    typ = np.zeros((n, stride), np.uint8)
    for i, (w, h, p) in enumerate(sizes):
        rc, a, o, t = O.mapgen(seed, i, w, h, p, stride)
        assert rc == 0
        army[i], owner[i], typ[i] = a, o, t
    w = np.array([s[0] for s in sizes], np.int32)
    h = np.array([s[1] for s in sizes], np.int32)
    p = np.array([s[2] for s in sizes], np.int32)
    return army, owner, typ, w, h, p


def assert_states_equal(hip, ora, ctx=""):
    """hip / ora: dicts from VecEngine.game_state / OracleBatch.read_state (full field sets)."""
    for f in TILE_FIELDS + ENV_FIELDS + PLAYER_FIELDS:
        if not np.array_equal(hip[f], ora[f]):
            bad = np.argwhere(hip[f] != ora[f])
            e = int(bad[0][0])
            raise AssertionError(f"{ctx}: field '{f}' differs in {len(set(b[0] for b in bad))} env(s); first env {e} "
                                 f"idx {bad[0][1:]}: hip={hip[f][tuple(bad[0])]} oracle={ora[f][tuple(bad[0])]} "
                                 f"(turn {ora['turn'][e]}, size {ora['width'][e]}x{ora['height'][e]} P{ora['players'][e]})")
    # GeneralIdx: the reference's value depends on Go map iteration order when a player holds >= 2
    # generals.  Contract: -1 iff no listed general, else SOME listed general tile.
    hg, og = hip["general_idx"], ora["general_idx"]
    assert np.array_equal(hg >= 0, og >= 0), f"{ctx}: general_idx sign differs"
    n, P = hg.shape
    for e, p in np.argwhere(hg >= 0):
        t = hg[e, p]
        assert ora["listed"][e, t] == p and ora["type"][e, t] == 1, f"{ctx}: env {e} player {p} general_idx {t} is not a listed general"


def run_lockstep(eng, ora, turns, seed, invalid_permille=0, check_every=1, want_mask=True, ctx=""):
    """Drives both engines with the oracle's random agent; compares err/masks/state."""
    for k in range(turns):
        acts = ora.agent_actions(seed, invalid_permille)
        if want_mask:
            oerr, obits = ora.step(acts, want_mask=True)
            herr, hbits = eng.step(acts, want_mask=True)
            assert np.array_equal(hbits, obits), f"{ctx} turn {k}: legal mask differs in envs {np.unique(np.argwhere(hbits != obits)[:, 0])[:8]}"
        else:
            oerr = ora.step(acts)
            herr = eng.step(acts)
        assert np.array_equal(herr, oerr), f"{ctx} turn {k}: err differs: envs {np.argwhere(herr != oerr)[:8].ravel()} hip={herr[herr != oerr][:8]} ora={oerr[herr != oerr][:8]}"
        if (k + 1) % check_every == 0 or k == turns - 1:
            assert_states_equal(eng.game_state(), ora.read_state(), f"{ctx} after turn {k + 1}")
