"""VecExperienceStreamClient / ExperienceDataset (the drop-in for python/experience_stream_client.py) against what the
REFERENCE'S OWN client did with the same batches (tests/golden/stream_client_fixtures.json "client_flow", recorded by
tests/golden/make_stream_client_fixtures.py), then - under -m gpu - fed by a real engine."""
import json
import os
import time

import numpy as np
import pytest

from generalsreinforcementlearning_amd.experience_stream import (ExperienceConfig, ExperienceDataset, VecExperienceStreamClient,
                                                                 engine_experience_source)

FLOW = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stream_client_fixtures.json")))["client_flow"]


def flow_experience(i):
    """The experience make_stream_client_fixtures.flow_experience(i) puts on the wire, as the dict a local source yields.
    (float32 on the wire: TensorState.data is `repeated float`.)"""
    st = np.array([float((i * 7 + k) % 5) / 4.0 for k in range(54)], np.float32).reshape(9, 2, 3)
    nx = np.array([float((i * 3 + k) % 7) / 8.0 for k in range(54)], np.float32).reshape(9, 2, 3)
    return {"experience_id": f"flow-{i}", "game_id": f"game-{i % 3}", "player_id": i % 2, "turn": 10 + i, "state": st, "action": i % 24,
            "reward": float(np.float32(i * 0.25 - 1.0)), "next_state": nx, "done": i % 5 == 4,
            "action_mask": np.array([(i + k) % 3 == 0 for k in range(24)]) if i % 4 else []}


def _batches():
    i, out = 0, []
    for n in FLOW["batch_sizes"]:
        out.append([flow_experience(i + k) for k in range(n)])
        i += n
    return out


def test_client_flow_matches_the_reference_client():
    client = VecExperienceStreamClient(ExperienceConfig(buffer_size=FLOW["buffer_size"]), lambda cfg: iter(()))
    for b in _batches():
        client.ingest_batch(b)
    stats = client.get_stats()
    assert {k: stats[k] for k in FLOW["stats"]} == FLOW["stats"] and isinstance(stats["last_batch_time"], float)
    assert set(stats) == {"total_experiences", "total_batches", "dropped_experiences", "last_batch_time", "queue_size", "streaming"}
    first = client.get_batch(4, timeout=1.0)
    assert [e["experience_id"] for e in first] == FLOW["get_batch_4"] and client.get_stats()["queue_size"] == FLOW["queue_size_after"]
    assert [e["experience_id"] for e in first if e["action_mask"] is None] == FLOW["mask_none_ids"]
    d0 = first[0]
    assert {k: type(v).__name__ for k, v in d0.items()} == FLOW["first_types"]
    for k, want in FLOW["first"].items():
        got = d0[k]
        if isinstance(got, np.ndarray):
            assert got.dtype == np.float32 and np.array_equal(got, np.array(want, np.float32)), k
        else:
            assert got == want, k
    e1 = first[1]
    assert e1["action_mask"].dtype == np.bool_ and e1["state"].shape == (9, 2, 3)
    ds = ExperienceDataset(client, buffer_size=5)
    ds.fill_buffer(min_size=3)
    assert [e["experience_id"] for e in ds.buffer] == FLOW["dataset_buffer"]
    np.random.seed(FLOW["dataset_seed"])
    assert [e["experience_id"] for e in ds.sample(2)] == FLOW["dataset_sample_2"]
    assert [e["experience_id"] for e in ds.sample(9)] == FLOW["dataset_sample_9"]
    assert client.get_experience(timeout=0.05) is None


def test_streaming_thread_lifecycle():
    """connect / start_streaming / stop_streaming / disconnect around a finite source (experience_stream_client.py:61-114)."""
    closed = []

    class Source:
        def __init__(self):
            self.it = iter(_batches())

        def __iter__(self):
            return self

        def __next__(self):
            return next(self.it)

        def close(self):
            closed.append(True)

    client = VecExperienceStreamClient(ExperienceConfig(buffer_size=100), lambda cfg: Source())
    client.connect()
    client.start_streaming()
    client.start_streaming()            # "Streaming already started" or a finished thread restarted: harmless either way
    t0 = time.time()
    while client.get_stats()["total_experiences"] < 14 and time.time() - t0 < 5:
        time.sleep(0.01)
    st = client.get_stats()
    assert st["total_experiences"] == 14 and st["total_batches"] == 4 and st["dropped_experiences"] == 0 and st["queue_size"] == 14
    got = client.get_batch(14, timeout=2.0)
    assert [e["experience_id"] for e in got] == [f"flow-{i}" for i in range(14)]
    client.stop_streaming()
    assert not client.get_stats()["streaming"]
    client.disconnect()
    assert closed == [True]


@pytest.mark.gpu
def test_stream_client_over_an_engine_source():
    """The whole chain on a GPU: engine -> compact records -> GPU expansion -> StreamAggregator-style batches -> the
    client's queue -> the dicts a trainer consumes.  Checked against VecExperienceCollector on a twin engine stepped with
    the same (device-agent) actions, and the request filters."""
    import torch
    import generalsreinforcementlearning_amd as g
    from generalsreinforcementlearning_amd.experience import VecExperienceCollector
    B, w, h, p, n_rec, steps = 64, 8, 8, 3, 24, 12

    def make():
        e = g.VecEngine(B, w, h, p, auto_reset=True, stream=torch.cuda.current_stream().cuda_stream)
        e.reset_generated(11)
        e.build_board_pool(8, 2)
        return e

    eng, twin = make(), make()
    cfg = ExperienceConfig(batch_size=16, follow=False, buffer_size=10000, max_batch_wait_ms=10_000)
    client = VecExperienceStreamClient(cfg, engine_experience_source(eng, n_rec, seed=5, max_steps=steps))
    client.connect()
    client.start_streaming()
    # the twin: the same turns through the collector (host tensors), envs [0, n_rec) only
    col = VecExperienceCollector(twin)
    want = []
    for _ in range(steps):
        acts = twin.agent_actions(5)
        col.before_step()
        twin.step(acts)
        b = col.after_step(acts)
        for i in range(len(b["env"])):
            if b["env"][i] < n_rec and b["turn"][i] > 0:
                want.append((int(b["env"][i]), int(b["player_id"][i]), int(b["turn"][i]), int(b["action"][i]), float(b["reward"][i]),
                             bool(b["done"][i]), b["state"][i], b["next_state"][i], b["action_mask"][i]))
    t0 = time.time()
    while client.get_stats()["streaming"] and time.time() - t0 < 30:
        time.sleep(0.02)
    got = client.get_batch(10 ** 6, timeout=1.0)
    st = client.get_stats()
    assert st["dropped_experiences"] == 0 and st["total_experiences"] == len(got) and st["total_batches"] == -(-len(got) // 16)
    # an env re-dealt in a step yields no experience on either side; everything else matches one to one, in order
    assert len(got) == len(want) > 100
    for d, (env, pl, turn, action, reward, done, s0, s1, mask) in zip(got, want):
        assert d["game_id"] == f"vec-env{env}" and d["player_id"] == pl and d["turn"] == turn and d["action"] == action and d["done"] == done
        assert np.float32(d["reward"]).tobytes() == np.float32(reward).tobytes()
        assert d["state"].shape == (9, h, w) and d["state"].dtype == np.float32
        assert np.array_equal(d["state"].view(np.uint32), np.asarray(s0).reshape(9, h, w).view(np.uint32))
        assert np.array_equal(d["next_state"].view(np.uint32), np.asarray(s1).reshape(9, h, w).view(np.uint32))
        assert d["action_mask"].dtype == np.bool_ and np.array_equal(d["action_mask"], mask)
    client.stop_streaming()
    client.disconnect()
    # request filters (StreamExperiencesRequest.game_ids / player_ids)
    eng2 = make()
    cfg2 = ExperienceConfig(batch_size=8, follow=False, buffer_size=10000, game_ids=["vec-env3", "vec-env5"], player_ids=[1])
    c2 = VecExperienceStreamClient(cfg2, engine_experience_source(eng2, n_rec, seed=5, max_steps=steps))
    c2.connect()
    c2.start_streaming()
    t0 = time.time()
    while c2.get_stats()["streaming"] and time.time() - t0 < 30:
        time.sleep(0.02)
    sel = c2.get_batch(10 ** 6, timeout=0.5)
    assert sel and all(d["game_id"] in ("vec-env3", "vec-env5") and d["player_id"] == 1 for d in sel)
    assert [(d["game_id"], d["turn"]) for d in sel] == [(d["game_id"], d["turn"]) for d in got if d["game_id"] in ("vec-env3", "vec-env5") and d["player_id"] == 1]


@pytest.mark.gpu
def test_reference_shaped_training_loop_on_the_reference_names():
    """rl_training_example.py:164-240 in outline, with nothing changed but the import: ExperienceStreamClient(config) ->
    connect -> start_streaming -> ExperienceDataset(client, buffer_size) -> fill_buffer(min_size) -> sample(batch) ->
    get_stats -> stop_streaming -> disconnect."""
    from generalsreinforcementlearning_amd.experience_stream_client import (ExperienceConfig, ExperienceDataset, ExperienceStreamClient,
                                                                            configure_local_engine)
    configure_local_engine(num_envs=128, width=10, height=10, players=2, records_per_step=32, seed=3, board_pool=32)
    with pytest.raises(TypeError):
        configure_local_engine(boards=3)
    config = ExperienceConfig(server_address="localhost:50051", batch_size=64, follow=True, buffer_size=2000)
    client = ExperienceStreamClient(config)
    client.connect()
    client.start_streaming()
    try:
        import time
        t0 = time.time()
        while client.get_stats()["total_experiences"] < 500 and time.time() - t0 < 120:   # the engine opens on the stream thread
            time.sleep(0.05)
        dataset = ExperienceDataset(client, buffer_size=1000)
        dataset.fill_buffer(min_size=400)
        assert len(dataset.buffer) >= 400
        batch = dataset.sample(32)
        assert len(batch) == 32
        e = batch[0]
        assert set(e) >= {"experience_id", "game_id", "player_id", "turn", "state", "action", "reward", "next_state", "done", "action_mask"}
        assert e["state"].shape == (9, 10, 10) and e["state"].dtype == np.float32 and e["next_state"].shape == (9, 10, 10)
        assert e["action_mask"].shape == (400,) and isinstance(e["action"], int) and isinstance(e["reward"], float) and isinstance(e["done"], bool)
        stats = client.get_stats()
        assert stats["total_experiences"] >= 400 and stats["total_batches"] >= 400 // 64 and "queue_size" in stats and "dropped_experiences" in stats
    finally:
        client.stop_streaming()
        client.disconnect()
    assert client._engine is None
