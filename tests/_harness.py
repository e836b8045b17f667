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


# Every compiled instantiation of the kernels: <max players 2 / 4 / 8> x <1, 2, 4, 7, 10, 16 slots of 64 tiles> x <odd / even
# number of plane dwords> (gvec_kernels.hip dispatch, launch_step).  The board limits below select each <slots, parity>.
VARIANT_DIMS = {(1, "odd"): (5, 5), (1, "even"): (8, 8), (2, "odd"): (9, 10), (2, "even"): (11, 11), (4, "odd"): (14, 15),
                (4, "even"): (16, 16), (7, "odd"): (20, 20), (7, "even"): (21, 21), (10, "odd"): (24, 25), (10, "even"): (25, 25),
                (16, "odd"): (30, 32), (16, "even"): (32, 32)}
VARIANT_IDS = [f"slots{s}_{o}" for s, o in sorted(VARIANT_DIMS)]


def variant_batch(maxp, slots, parity, B):
    """(max_w, max_h, per-env (w, h, players)) of a ragged batch on the limits that select the variant: the variant follows
    the handle's player limit; a small board takes as many generals as the generator can space out."""
    mw, mh = VARIANT_DIMS[(slots, parity)]
    assert (mw * mh + 63) // 64 <= slots and (slots == 1 or mw * mh > {2: 64, 4: 128, 7: 256, 10: 448, 16: 640}[slots])
    players = {2: [2], 4: [3, 4], 8: [5, 8, 6]}[maxp]
    small = (max(3, mw - 3), max(3, mh - 2))
    dims = [(mw, mh) if i % 3 else small for i in range(B)]
    return mw, mh, [d + (min(players[i % len(players)], max(2, d[0] * d[1] // 30)),) for i, d in enumerate(dims)]
