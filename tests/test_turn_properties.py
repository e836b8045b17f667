"""Restatement-free properties of the turn (tests/_properties.py) over long random rollouts with deliberately invalid
moves: aborted turns (H5), list desync (H6), stale-list production (H7), the incremental / full stats rule, combat
arithmetic, ChangedTiles / VisibilityChangedTiles - for the CPU oracle here, for the HIP engine under -m gpu.  Neither
is compared with the other: each is compared with the Go rules as written."""
import numpy as np
import pytest

import _harness as H
import _oracle as O
import _properties as PR

CASES = [([(10, 10, 2)], 25), ([(12, 9, 3), (8, 8, 2), (15, 15, 4)], 15), ([(20, 20, 4), (6, 6, 2)], 40)]
IDS = ["10x10_p2", "mixed_p2_p4", "20x20_p4_many_invalid"]


def _run(engine_step, engine_state, agent, B, turns):
    counters = PR.new_counters()
    before = engine_state()
    for _ in range(turns):
        acts = agent()
        err = engine_step(acts)
        after = engine_state()
        for e in range(B):
            PR.check_turn(before, acts, err, after, e, counters)
        before = after
    return counters


def _coverage(c):
    assert c["aborted_checked"] > 20 and c["incremental"] > 500 and c["moves"] > 500 and c["desynced_start"] > 5, c


@pytest.mark.parametrize("sizes,invalid", CASES, ids=IDS)
def test_oracle_turns_obey_the_rules(sizes, invalid):
    B = 16
    per_env = [sizes[i % len(sizes)] for i in range(B)]
    mw, mh, mp = max(s[0] for s in per_env), max(s[1] for s in per_env), max(s[2] for s in per_env)
    army, owner, typ, ws, hs, ps = H.gen_boards(71, per_env, mw, mh)
    ora = O.OracleBatch(B, mw, mh, mp)
    ora.reset(army, owner, typ, ws, hs, ps)
    c = _run(ora.step, ora.read_state, lambda: ora.agent_actions(5, invalid), B, 260)
    _coverage(c)
    if sizes[0][0] == 10:
        assert c["full"] > 0        # growth turns change more than N/5 tiles on a well-populated board


@pytest.mark.gpu
@pytest.mark.parametrize("sizes,invalid", CASES, ids=IDS)
def test_hip_turns_obey_the_rules(sizes, invalid):
    import generalsreinforcementlearning_amd as g
    B = 16
    per_env = [sizes[i % len(sizes)] for i in range(B)]
    mw, mh, mp = max(s[0] for s in per_env), max(s[1] for s in per_env), max(s[2] for s in per_env)
    army, owner, typ, ws, hs, ps = H.gen_boards(71, per_env, mw, mh)
    eng = g.VecEngine(B, mw, mh, mp)
    eng.reset(army, owner, typ, ws, hs, ps)
    c = _run(eng.step, eng.game_state, lambda: eng.agent_actions(5, invalid), B, 260)
    _coverage(c)
