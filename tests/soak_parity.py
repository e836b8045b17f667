#!/usr/bin/env python3
"""Long lock-step soak of the HIP engine against the CPU oracle (not collected by pytest: run by hand on a GPU box).

    python tests/soak_parity.py [seeds=3] [turns=1500]

Every configuration of test_hip_parity.CONFIGS with more seeds and more turns, every turn compared
(err, legal masks, every state field), plus auto-reset rollouts (per-turn and fused launches) compared
against the oracle's batch rollout.  Prints one line per case so a long run shows progress.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _harness as H  # noqa: E402
import _oracle as O  # noqa: E402
import generalsreinforcementlearning_amd as g  # noqa: E402

CONFIGS = [
    ("10x10_p2_fog_off", 256, [(10, 10, 2)], False),
    ("15x15_p2_fog_on", 192, [(15, 15, 2)], True),
    ("20x20_p4_fog_on", 192, [(20, 20, 4)], True),
    ("mixed_padded", 192, [(10, 10, 2), (15, 15, 3), (20, 20, 4)], True),
    ("tiny_boards", 64, [(3, 3, 2), (5, 5, 2), (8, 8, 3), (7, 5, 2), (2, 9, 2)], True),
    ("wide_p8", 48, [(25, 25, 8), (32, 32, 5), (32, 17, 6), (19, 31, 7)], True),
    ("odd_strides", 64, [(14, 16, 4), (21, 21, 3), (12, 8, 4)], True),
    ("25x25_p4", 48, [(25, 25, 4), (22, 24, 3)], True),
    ("20x20_p8", 48, [(20, 20, 8), (16, 21, 6), (21, 21, 5)], True),
    ("25x25_p8", 32, [(25, 25, 8), (24, 26, 7)], True),
]


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    turns = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    t_start = time.time()
    for name, B, pattern, fog in CONFIGS:
        for seed in range(seeds):
            sizes = [pattern[(i + seed) % len(pattern)] for i in range(B)]
            mw, mh, mp = max(s[0] for s in sizes), max(s[1] for s in sizes), max(s[2] for s in sizes)
            army, owner, typ, w, h, p = H.gen_boards(9000 + seed, sizes, mw, mh)
            eng = g.VecEngine(B, mw, mh, mp, fog_of_war=fog)
            ora = O.OracleBatch(B, mw, mh, mp, fog=fog)
            eng.reset(army, owner, typ, w, h, p)
            ora.reset(army, owner, typ, w, h, p)
            t0 = ((np.arange(B) * 7 + seed) % 25).astype(np.int32)
            eng.write_state({"turn": t0})
            ora.write_state({"turn": t0})
            H.run_lockstep(eng, ora, turns, seed=100 + seed, invalid_permille=(0, 5, 25)[seed % 3], check_every=1, ctx=f"{name}/s{seed}")
            done = int(ora.read_state()["done"].sum())
            print(f"[soak] {name} seed {seed}: {turns} turns x {B} boards identical ({done} games finished)  t={time.time() - t_start:.0f}s", flush=True)
    # auto-reset rollouts: per-turn launches and fused launches against the oracle's rollout
    for fused in (False, True):
        for (w, h, p) in ((20, 20, 4), (15, 15, 2), (7, 7, 3)):
            B = 384
            army, owner, typ, ww, hh, pp = H.gen_boards(5150, [(w, h, p)] * B, w, h)
            eng = g.VecEngine(B, w, h, p, auto_reset=True)
            ora = O.OracleBatch(B, w, h, p)
            eng.reset(army, owner, typ, ww, hh, pp)
            ora.reset(army, owner, typ, ww, hh, pp)
            eng.build_board_pool(61, 777)
            ora.set_pool(61, 777)
            fin = 0
            for chunk in range(6):
                st = eng.rollout(250, 31 + chunk, 10, fused=fused)
                steps = ora.rollout(250, 31 + chunk, 10)
                assert st["env_steps"] == steps
                fin += st["games_finished"]
                H.assert_states_equal(eng.game_state(), ora.read_state(), f"rollout fused={fused} {w}x{h} chunk {chunk}")
                assert np.array_equal(eng.legal_action_mask_bits(), ora.legal_mask())
            print(f"[soak] auto-reset rollout fused={fused} {w}x{h} P{p}: 1500 turns x {B} boards identical ({fin} games finished)  t={time.time() - t_start:.0f}s", flush=True)
    print("[soak] all identical")


if __name__ == "__main__":
    main()
