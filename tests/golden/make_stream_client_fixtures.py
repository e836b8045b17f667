#!/usr/bin/env python3
"""Generates tests/golden/stream_client_fixtures.json — runs in the BUILD container only.

What it pins: the record layout a trainer sees.  For a fixed set of oracle transitions it builds the
`experiencepb.Experience` messages SimpleCollector.OnStateTransition would emit
(/root/reference/internal/experience/collector.go:30-98: state / next_state as TensorState{shape =
[9, H, W], data}, action = Serializer.ActionToIndex, reward, done, action_mask), serialises them,
and runs the REFERENCE'S OWN client code on the parsed message:
`ExperienceStreamClient._process_experience` (/root/reference/python/experience_stream_client.py:134-158,
imported here, never copied).  The resulting dicts (arrays as nested lists, float32 values as their
exact Python floats) plus the inputs that reproduce them (boards, per-turn actions) are the fixture.

Nothing of the reference travels: only this JSON does.  Consumers:
  * tests/test_stream_client_fixtures.py (CPU): the oracle replays the inputs and must reproduce the dicts;
  * tests/test_hip_experience.py (GPU): VecExperienceCollector.as_dicts on the HIP engine must equal them.

    python tests/golden/make_stream_client_fixtures.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("GRL_REFERENCE_DIR", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(REF, "python"))

import _harness as H  # noqa: E402
import _oracle as O  # noqa: E402

from experience_stream_client import ExperienceConfig, ExperienceStreamClient  # noqa: E402  (the reference's client)
from generals_pb.experience.v1 import experience_pb2  # noqa: E402  (the reference's committed stubs)

SIZES = [(5, 5, 2), (6, 5, 3), (7, 7, 4)]   # (W, H, players): a padded mixed batch
MAX_W, MAX_H, MAX_P = 7, 7, 4
WARM_TURNS, RECORD_TURNS, SEED = 14, 5, 2026


def action_index(a, w):
    """Serializer.ActionToIndex (internal/experience/serializer.go:179-198)."""
    dx, dy = int(a["to_x"]) - int(a["from_x"]), int(a["to_y"]) - int(a["from_y"])
    d = {(0, -1): 0, (0, 1): 1, (-1, 0): 2, (1, 0): 3}.get((dx, dy), 0)
    return (int(a["from_y"]) * w + int(a["from_x"])) * 4 + d


def flow_experience(i):
    """A small deterministic Experience (3x2 board) for the queue / statistics / dataset flow."""
    st = [float((i * 7 + k) % 5) / 4.0 for k in range(9 * 2 * 3)]
    nx = [float((i * 3 + k) % 7) / 8.0 for k in range(9 * 2 * 3)]
    return experience_pb2.Experience(experience_id=f"flow-{i}", game_id=f"game-{i % 3}", player_id=i % 2, turn=10 + i,
                                     state=experience_pb2.TensorState(shape=[9, 2, 3], data=st), action=i % 24, reward=i * 0.25 - 1.0,
                                     next_state=experience_pb2.TensorState(shape=[9, 2, 3], data=nx), done=(i % 5 == 4),
                                     action_mask=[(i + k) % 3 == 0 for k in range(24)] if i % 4 else [])


def client_flow():
    """The reference client's queue, drop accounting, statistics, get_batch and ExperienceDataset on hand-built batches
    (no server: _process_batch is what the stream worker calls per batch, experience_stream_client.py:105-110)."""
    from experience_stream_client import ExperienceDataset
    client = ExperienceStreamClient(ExperienceConfig(buffer_size=7))
    sizes = [3, 4, 2, 5]                                   # 14 experiences into a queue of 7: the rest is dropped
    i = 0
    for bi, n in enumerate(sizes):
        batch = experience_pb2.ExperienceBatch(batch_id=bi, stream_id="flow")
        for _ in range(n):
            batch.experiences.append(flow_experience(i))
            i += 1
        client._process_batch(experience_pb2.ExperienceBatch.FromString(batch.SerializeToString()))
    stats = client.get_stats()
    first = client.get_batch(4, timeout=1.0)
    stats_after = client.get_stats()
    ds = ExperienceDataset(client, buffer_size=5)
    ds.fill_buffer(min_size=3)                              # drains the remaining 3
    buf_ids = [e["experience_id"] for e in ds.buffer]
    np.random.seed(3)
    draw = [e["experience_id"] for e in ds.sample(2)]
    short = [e["experience_id"] for e in ds.sample(9)]     # more than there is: the whole buffer
    d0 = first[0]
    return {"buffer_size": 7, "batch_sizes": sizes,
            "stats": {k: stats[k] for k in ("total_experiences", "total_batches", "dropped_experiences", "queue_size", "streaming")},
            "last_batch_time_is_float": isinstance(stats["last_batch_time"], float),
            "get_batch_4": [e["experience_id"] for e in first], "queue_size_after": stats_after["queue_size"],
            "dataset_buffer": buf_ids, "dataset_seed": 3, "dataset_sample_2": draw, "dataset_sample_9": short,
            "first_types": {k: type(v).__name__ for k, v in d0.items()},
            "first": {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in d0.items()},
            "mask_none_ids": [e["experience_id"] for e in first if e["action_mask"] is None]}


def main():
    B = len(SIZES)
    army, owner, typ, ws, hs, ps = H.gen_boards(SEED, SIZES, MAX_W, MAX_H)
    ora = O.OracleBatch(B, MAX_W, MAX_H, MAX_P)
    ora.reset(army, owner, typ, ws, hs, ps)
    client = ExperienceStreamClient(ExperienceConfig())          # no connection is made
    turns, expected = [], []
    for k in range(WARM_TURNS + RECORD_TURNS):
        acts = ora.agent_actions(SEED + 1)
        record = k >= WARM_TURNS
        if record:
            prev = {(e, p): (ora.engine(e).state_to_tensor(p), ora.engine(e).serializer_mask(p))
                    for e in range(B) for p in range(SIZES[e][2])}
            ora.experience_begin()
        turns.append(acts.view(np.uint8).reshape(B, MAX_P, 8)[:, :, :5].tolist())   # from_x, from_y, to_x, to_y, flags
        ora.step(acts)
        if not record:
            continue
        rewards, done = ora.rewards()
        turn_now = ora.read_state(fields=("turn",))["turn"]
        for e in range(B):
            w, h, P = SIZES[e]
            for p in range(P):
                if not (acts[e, p]["flags"] & 1):
                    continue                                      # collector.go:33-37: players that took an action
                st, mask = prev[(e, p)]
                exp = experience_pb2.Experience(
                    experience_id=f"fixture-{len(expected)}", game_id=f"fixture-env{e}", player_id=p, turn=int(turn_now[e]),
                    state=experience_pb2.TensorState(shape=[9, h, w], data=st.tolist()),
                    action=action_index(acts[e, p], w), reward=float(rewards[e, p]),
                    next_state=experience_pb2.TensorState(shape=[9, h, w], data=ora.engine(e).state_to_tensor(p).tolist()),
                    done=bool(done[e]), action_mask=[bool(v) for v in mask])
                wire = exp.SerializeToString()                     # what the stream carries
                d = client._process_experience(experience_pb2.Experience.FromString(wire))
                assert d["state"].dtype == np.float32 and d["action_mask"].dtype == np.bool_
                expected.append({"env": e, "record_turn": k - WARM_TURNS,
                                 "dtypes": {"state": str(d["state"].dtype), "next_state": str(d["next_state"].dtype),
                                            "action_mask": str(d["action_mask"].dtype)},
                                 "types": {key: type(v).__name__ for key, v in d.items()},
                                 "dict": {key: (v.tolist() if isinstance(v, np.ndarray) else v) for key, v in d.items()}})
    out = {"comment": "generated by tests/golden/make_stream_client_fixtures.py with the reference's "
                      "ExperienceStreamClient._process_experience (experience_stream_client.py:134-158); do not edit",
           "sizes": SIZES, "max": [MAX_W, MAX_H, MAX_P], "seed": SEED, "warm_turns": WARM_TURNS, "record_turns": RECORD_TURNS,
           "boards": {"army": army.tolist(), "owner": owner.tolist(), "type": typ.tolist()},
           "turn_actions": turns, "expected": expected}
    out["client_flow"] = client_flow()
    path = os.path.join(HERE, "stream_client_fixtures.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"wrote {path}: {len(expected)} experiences, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
