"""Scenario tests for the hazards the reference's own tests do NOT pin (SURVEY.md section 8c,
S1-S7): expected values are derived below from the cited Go lines (paths relative to
/root/reference/internal/game/).  Each scenario runs on the CPU oracle (always) and on the HIP
path through the C ABI (-m gpu); both expose the same plane API.

Board: 10x10 (stats full-update threshold N/5 = 20, fog threshold N/10 = 10), turns kept off
multiples of 25 unless stated.
"""
import numpy as np
import pytest

import _oracle as O

N_, G_, C_, M_ = 0, 1, 2, 3
W = H = 10


def T(x, y, owner=-1, army=0, type=N_):
    return dict(x=x, y=y, owner=owner, army=army, type=type)


def idx(x, y):
    return y * W + x


class OracleBackend:
    def __init__(self, players, tiles):
        self.P = players
        self.e = O.OracleBatch(1, W, H, players)
        army, owner, typ = O.planes_from_tiles(W, H, tiles)
        self.e.reset(army[None], owner[None], typ[None], [W], [H], [players])

    def step(self, moves):
        acts = np.zeros((1, self.P), O.ACTION_DTYPE)
        for (p, fx, fy, tx, ty, move_all) in moves:
            acts[0, p] = (fx, fy, tx, ty, 1 | (0 if move_all else 2), (0, 0, 0))
        return int(self.e.step(acts)[0])

    def state(self):
        return {k: v[0] for k, v in self.e.read_state().items()}

    def mask(self, p):
        from generalsreinforcementlearning_amd.vec_engine import unpack_legal_bits
        return unpack_legal_bits(self.e.legal_mask()[0, p], W, H)


class HipBackend(OracleBackend):
    def __init__(self, players, tiles):
        import generalsreinforcementlearning_amd as g
        self.P = players
        self.e = g.VecEngine(1, W, H, players)
        army, owner, typ = O.planes_from_tiles(W, H, tiles)
        self.e.reset(army[None], owner[None], typ[None], [W], [H], [players])

    def state(self):
        return {k: v[0] for k, v in self.e.game_state().items()}

    def mask(self, p):
        return self.e.get_legal_action_mask(0, p)


BACKENDS = [pytest.param(OracleBackend, id="oracle"), pytest.param(HipBackend, id="hip", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("B", BACKENDS)
def test_s1_neutral_capture_gets_no_production_this_turn(B):
    """H7: production walks the lists as of the LAST stats pass (production_manager.go:39-62); a
    neutral city captured this turn is not in the capturer's list yet."""
    b = B(2, [T(0, 0, 0, 5, G_), T(1, 0, -1, 2, C_), T(9, 9, 1, 2, G_)])
    assert b.step([(0, 0, 0, 1, 0, True)]) == 0
    s = b.state()
    # 4 move, 4 > 2 -> capture with 4-2 = 2 (movement.go:69-72); G0 keeps 1, +1 production = 2;
    # the city is unlisted this turn -> still 2
    assert (s["owner"][idx(1, 0)], s["army"][idx(1, 0)], s["army"][idx(0, 0)], s["army"][idx(9, 9)]) == (0, 2, 2, 3)
    assert s["listed"][idx(1, 0)] == 0  # joined L0 in the end-of-turn pass (it is in C, stats.go:108-126)
    assert b.step([]) == 0
    s = b.state()
    assert (s["army"][idx(1, 0)], s["army"][idx(0, 0)]) == (3, 3)


@pytest.mark.parametrize("B", BACKENDS)
def test_s2_captured_tile_still_produces_via_losers_stale_list(B):
    """H7: the loser is alive and still lists the tile, the owner is not re-checked
    (production_manager.go:39-62): 3 vs 1 -> 2, +1 through P1's list = 3."""
    b = B(2, [T(5, 5, 0, 2, G_), T(0, 0, 0, 4, N_), T(1, 0, 1, 1, C_), T(9, 9, 1, 2, G_)])
    assert b.step([(0, 0, 0, 1, 0, True)]) == 0
    s = b.state()
    assert (s["owner"][idx(1, 0)], s["army"][idx(1, 0)]) == (0, 3)
    assert s["listed"][idx(1, 0)] == 0


def _s3_board():
    # P0: general (5,5), A=(0,0) army 10, A2=(1,1) army 5.  P1: general (9,9), B=(1,0) army 3.
    return [T(5, 5, 0, 2, G_), T(0, 0, 0, 10, N_), T(1, 1, 0, 5, N_), T(9, 9, 1, 2, G_), T(1, 0, 1, 3, N_)]


@pytest.mark.parametrize("B", BACKENDS)
def test_s3_s4_aborted_turn_and_list_desync(B):
    """H5/H6: P0 (lower id) captures the tile P1 moves from; P1's move fails ErrNotOwned at
    application time (action.go:82-84); the first error is returned after the captures stand
    (engine.go:111-113) and ProcessTurn skips production, stats and game-over
    (turn_processor.go:55-57)."""
    b = B(2, _s3_board())
    assert b.step([(0, 0, 0, 1, 0, True), (1, 1, 0, 2, 0, True)]) == 3  # ErrNotOwned
    s = b.state()
    assert s["turn"] == 1  # Turn stays incremented
    assert (s["owner"][idx(1, 0)], s["army"][idx(1, 0)], s["army"][idx(0, 0)]) == (0, 6, 1)  # 9 vs 3 -> 6
    assert (s["army"][idx(5, 5)], s["army"][idx(9, 9)]) == (2, 2)  # NO production this turn
    assert s["vis_changed"][idx(1, 0)] == 1  # V kept for the next fog update
    assert s["listed"][idx(1, 0)] == 1  # no stats pass: still in P1's list, not in P0's
    assert s["owner"][idx(2, 0)] == -1  # P1's move did not happen
    # turn 2, nobody moves: C = {generals (production)}; B is not in C -> dropped from L1
    # (owner check stats.go:97) and NOT added to L0 (stats.go:108-126 only looks at C)
    assert b.step([]) == 0
    s = b.state()
    assert s["listed"][idx(1, 0)] == -1 and s["owner"][idx(1, 0)] == 0
    # S4 consequences of the desync: not counted, not in the legal mask
    assert s["tile_count"][0] == 3 and s["army_count"][0] == 3 + 1 + 5  # general 3, A 1, A2 5; B's 6 excluded
    m = b.mask(0)
    assert not m[idx(1, 0) * 4: idx(1, 0) * 4 + 4].any()  # legal_moves.go:37 walks the list
    assert m[idx(1, 1) * 4 + 0]  # A2 can move up into B
    # P0 moves A2 -> B (own tile): B joins C (movement.go:57-60) -> re-listed by the incremental pass
    assert b.step([(0, 1, 1, 1, 0, True)]) == 0
    s = b.state()
    assert s["listed"][idx(1, 0)] == 0 and s["army"][idx(1, 0)] == 10 and s["tile_count"][0] == 4
    assert b.mask(0)[idx(1, 0) * 4 + 1]  # and can move right again


@pytest.mark.parametrize("B", BACKENDS)
def test_s3b_stale_listed_city_produces_for_the_new_owner_and_heals(B):
    """After an aborted turn a captured CITY stays in the loser's list; next turn it produces through
    that stale list (H7), which puts it in C, which re-lists it for its real owner (H6)."""
    tiles = [T(5, 5, 0, 2, G_), T(0, 0, 0, 10, N_), T(9, 9, 1, 2, G_), T(1, 0, 1, 3, C_)]
    b = B(2, tiles)
    assert b.step([(0, 0, 0, 1, 0, True), (1, 1, 0, 2, 0, True)]) == 3
    assert b.state()["army"][idx(1, 0)] == 6
    assert b.step([]) == 0
    s = b.state()
    assert (s["army"][idx(1, 0)], s["owner"][idx(1, 0)], s["listed"][idx(1, 0)]) == (7, 0, 0)


@pytest.mark.parametrize("B", BACKENDS)
def test_s5_victim_capture_in_elimination_turn_stays_with_the_dead_player(B):
    """H2/H4: a player whose general falls earlier in the turn still moves (Alive as of turn start,
    action_processor.go:56-60); what it captures is not in its pre-turn list, so the turnover
    (engine.go:130-137) leaves it with the dead player, and dead players never produce
    (production_manager.go:40-42)."""
    tiles = [T(0, 0, 0, 2, G_), T(4, 4, 0, 10, N_), T(5, 4, 1, 1, G_), T(7, 7, 1, 6, N_), T(8, 7, -1, 2, C_),
             T(9, 0, 2, 2, G_)]
    b = B(3, tiles)
    assert b.step([(0, 4, 4, 5, 4, True), (1, 7, 7, 8, 7, True)]) == 0
    s = b.state()
    assert list(s["alive"]) == [1, 0, 1] and s["done"] == 0
    assert s["owner"][idx(5, 4)] == 0 and s["army"][idx(5, 4)] == 8 + 1  # captured general produces for P0
    assert s["owner"][idx(7, 7)] == 0  # P1's pre-turn land turned over
    assert (s["owner"][idx(8, 7)], s["army"][idx(8, 7)]) == (1, 3)  # 5 vs 2 -> 3, stays with dead P1, no production
    for _ in range(3):
        assert b.step([]) == 0
    assert b.state()["army"][idx(8, 7)] == 3  # a dead player's city never produces


@pytest.mark.parametrize("B", BACKENDS)
def test_s6_chain_elimination_resurrects_the_middle_player(B):
    """H4: orders are applied in capture order [1->0, 2->1] (movement.go:100-118, engine.go:120-151);
    P2's listed land goes to 'dead' P1, whose new general tile is in C, so the stats pass marks P1
    alive again (stats.go:108-126,133-135)."""
    tiles = [T(0, 0, 0, 2, G_), T(4, 4, 0, 10, N_), T(5, 4, 1, 1, G_), T(5, 5, 1, 3, N_), T(7, 7, 1, 10, N_),
             T(8, 7, 2, 1, G_), T(9, 9, 2, 4, N_)]
    b = B(3, tiles)
    assert b.step([(0, 4, 4, 5, 4, True), (1, 7, 7, 8, 7, True)]) == 0
    s = b.state()
    assert list(s["alive"]) == [1, 1, 0] and s["done"] == 0 and s["winner"] == -1
    assert (s["owner"][idx(7, 7)], s["army"][idx(7, 7)]) == (0, 1)   # P1's attacker tile -> P0
    assert (s["owner"][idx(5, 5)], s["army"][idx(5, 5)]) == (0, 3)   # P1's land -> P0
    assert (s["owner"][idx(9, 9)], s["army"][idx(9, 9)]) == (1, 4)   # P2's land -> P1
    assert (s["owner"][idx(5, 4)], s["army"][idx(5, 4)]) == (0, 9)   # 9 vs 1 -> 8, +1 production
    assert (s["owner"][idx(8, 7)], s["army"][idx(8, 7)]) == (1, 9)   # 9 vs 1 -> 8, +1 (P1 alive again)
    assert s["army"][idx(0, 0)] == 3
    assert s["general_idx"][1] == idx(8, 7) and s["general_idx"][2] == -1
    assert s["general_idx"][0] in (idx(0, 0), idx(5, 4))  # two generals: order-dependent in Go (H6)


@pytest.mark.parametrize("B", BACKENDS)
def test_s7_fog_lags_one_turn_behind_ownership(B):
    """H1: Turn++ -> updateFogOfWar -> clear C,V (turn_processor.go:124-135): visibility after
    Step(t) reflects the lists at the end of t-1."""
    b = B(2, [T(0, 9, 0, 2, G_), T(2, 2, 0, 5, N_), T(9, 9, 1, 2, G_), T(8, 8, 1, 3, N_)])
    s0 = b.state()
    assert s0["visible"][idx(3, 2)] & 1 and not (s0["visible"][idx(4, 2)] & 1)
    assert b.step([(0, 2, 2, 3, 2, True)]) == 0
    s1 = b.state()
    assert s1["owner"][idx(3, 2)] == 0
    for y in (1, 2, 3):
        assert not (s1["visible"][idx(4, y)] & 1)  # new 3x3 not lit yet
    assert b.step([]) == 0
    s2 = b.state()
    for y in (1, 2, 3):
        assert s2["visible"][idx(4, y)] & 1  # lit by turn 2's incremental update (visibility_optimized.go:56-97)
    # P1 never had anything near: untouched bits
    assert not (s2["visible"][idx(3, 2)] & 2)


@pytest.mark.parametrize("B", BACKENDS)
def test_loser_keeps_seeing_until_next_turn_and_loses_it_then(B):
    """Companion to S7: the loser's bits around a captured tile are cleared at the NEXT fog update
    unless another listed tile of theirs is in range (visibility_optimized.go:76-94)."""
    b = B(2, [T(0, 9, 0, 2, G_), T(2, 2, 0, 9, N_), T(3, 2, 1, 1, N_), T(9, 9, 1, 2, G_)])
    assert b.state()["visible"][idx(4, 2)] & 2
    assert b.step([(0, 2, 2, 3, 2, True)]) == 0
    assert b.state()["visible"][idx(4, 2)] & 2  # still lit: fog ran before the move
    assert b.step([]) == 0
    s = b.state()
    assert not (s["visible"][idx(4, 2)] & 2) and (s["visible"][idx(4, 2)] & 1)


@pytest.mark.parametrize("B", BACKENDS)
def test_full_update_thresholds(B):
    """|C| > N/5 forces a full stats pass (stats.go:20-21): on a growth turn every owned normal tile
    produces and joins C, so a desynced tile is re-listed without being touched."""
    b = B(2, _s3_board() + [T(x, 7, 0, 1, N_) for x in range(10)] + [T(x, 6, 0, 1, N_) for x in range(10)])
    assert b.step([(0, 0, 0, 1, 0, True), (1, 1, 0, 2, 0, True)]) == 3
    assert b.step([]) == 0
    assert b.state()["listed"][idx(1, 0)] == -1
    # fast-forward to the turn before a growth turn
    if isinstance(b, HipBackend):
        b.e.write_state({"turn": np.array([24], np.int32)})
    else:
        b.e.write_state({"turn": np.array([24], np.int32)})
    assert b.step([]) == 0  # turn 25: 23 listed P0 tiles + P1 general produce -> |C| = 24 > 20 -> full pass
    s = b.state()
    assert s["turn"] == 25 and s["listed"][idx(1, 0)] == 0
    assert s["army"][idx(1, 0)] == 6  # it was unlisted when production ran, so it did not grow
