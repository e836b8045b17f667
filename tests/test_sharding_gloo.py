"""N>1 path on CPU: world_size-2 gloo run of the sharding + record-gather plumbing bench.py uses."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from generalsreinforcementlearning_amd.sharding import RecordGather, shard_range


def test_shard_range_partitions():
    for total in (1, 7, 8, 262144, 262145):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(n for _, n in spans) == total
            for (b0, n0), (b1, _) in zip(spans, spans[1:]):
                assert b0 + n0 == b1
            assert max(n for _, n in spans) - min(n for _, n in spans) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total, rec = 1000, 64
        begin, n = shard_range(total, world, rank)
        rg = RecordGather(16 * rec, torch.device("cpu"))
        out = []
        for step in range(3):
            # each rank fills its slab with (global env id, step) markers for 16 of its envs
            ids = begin + (np.arange(16) + 16 * step) % n
            payload = np.repeat(((ids * 7 + step) % 251).astype(np.uint8), rec)
            rg.send.copy_(torch.from_numpy(payload))
            got = rg.gather()
            if rank == 0:
                out.append([g.numpy().copy() for g in got])
        dist.barrier()
        if rank == 0:
            q.put(out)
    finally:
        dist.destroy_process_group()


def test_record_gather_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for step, slabs in enumerate(out):
        assert len(slabs) == 2
        for r, slab in enumerate(slabs):
            begin, n = shard_range(1000, 2, r)
            ids = begin + (np.arange(16) + 16 * step) % n
            assert np.array_equal(slab, np.repeat(((ids * 7 + step) % 251).astype(np.uint8), 64))


# ---- experience records over the N > 1 gather path --------------------------------------------------
def _rank_rollout(rank, steps):
    """Rank `rank`'s boards and turns (deterministic): yields (oracle, snapshot, actions) after every step."""
    import _harness as H
    import _oracle as O
    import _records as R
    sizes = [[(8, 8, 2), (10, 7, 3)][(i + rank) % 2] for i in range(6)]
    army, owner, typ, ws, hs, ps = H.gen_boards(40 + rank, sizes, 10, 8)
    ora = O.OracleBatch(6, 10, 8, 3)
    ora.reset(army, owner, typ, ws, hs, ps)
    for k in range(steps):
        acts = ora.agent_actions(9 + rank)
        prev = {(e, p): (ora.engine(e).state_to_tensor(p), ora.engine(e).serializer_mask(p)) for e in range(6) for p in range(sizes[e][2])}
        snap = R.capture(ora)
        ora.experience_begin()
        ora.step(acts)
        yield ora, snap, acts, prev, sizes


def _exp_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _records as R
    from generalsreinforcementlearning_amd.experience import decode_records
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lay = R.layout_for(10, 8, 3)
        rg = RecordGather(6 * lay["record_dw"] * 4, torch.device("cpu"))
        decoded = []
        for ora, snap, acts, prev, sizes in _rank_rollout(rank, 12):
            rec = R.encode(ora, snap, acts, lay, env_id_base=rank * 1000)       # what gvec_experience_records writes
            rg.send.copy_(torch.from_numpy(rec.view(np.uint8).reshape(-1)))
            got = rg.gather()
            if rank == 0:
                decoded.append([decode_records(g.numpy(), lay) for g in got])   # the StreamAggregator side expands
        dist.barrier()
        if rank == 0:
            q.put(decoded)
    finally:
        dist.destroy_process_group()


def test_experience_records_gathered_and_decoded_world2_gloo():
    """The one exchange step of the path: every rank ships compact experience records to rank 0, which
    expands them into the Experience fields (collector.go:30-98).  Here on gloo / CPU with records built from
    the oracle by the format's numpy restatement; the GPU test checks the HIP kernel emits those same bytes."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    decoded = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = 0
    gens = [_rank_rollout(r, 12) for r in range(2)]
    for step in range(12):
        for r in range(2):
            ora, snap, acts, prev, sizes = next(gens[r])
            rewards, done = ora.rewards()
            turn = ora.read_state(fields=("turn",))["turn"]
            d = decoded[step][r]
            want = [(e, p) for e in range(6) for p in range(sizes[e][2]) if acts[e, p]["flags"] & 1]
            assert [(int(e) - 1000 * r, int(p)) for e, p in zip(d["env"], d["player_id"])] == want
            for i, (e, p) in enumerate(want):
                w, h, _ = sizes[e]
                assert np.asarray(d["state"][i]).shape == (9, h, w)
                assert np.array_equal(np.asarray(d["state"][i]).ravel().view(np.uint32), prev[(e, p)][0].view(np.uint32))
                assert np.array_equal(np.asarray(d["next_state"][i]).ravel().view(np.uint32), ora.engine(e).state_to_tensor(p).view(np.uint32))
                assert np.array_equal(d["action_mask"][i], prev[(e, p)][1].astype(bool))
                assert d["reward"][i] == rewards[e, p] and d["done"][i] == bool(done[e]) and d["turn"][i] == turn[e] and d["valid"][i]
                assert d["action_mask"][i][d["action"][i]]
            total += len(want)
    assert total > 100
