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
