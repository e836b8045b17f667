"""Multi-GPU plumbing for the turn engine: boards are independent, so ranks own disjoint
contiguous env ranges and never exchange data on the step path.  The only exchange is the gather
of compact experience/state record slabs to the rank that feeds the host-side aggregator
(reference: internal/grpc/gameserver/stream_aggregator.go:75-155).  torch.distributed is plumbing
here: backend "nccl" is RCCL over xGMI on MI355X; the same code runs on "gloo" for CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(total_envs, world_size, rank):
    """Contiguous env range [begin, begin+n) of `rank`; sizes differ by at most one."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(total_envs, world_size)
    n = base + (1 if rank < rem else 0)
    begin = rank * base + min(rank, rem)
    return begin, n


class RecordGather:
    """Gathers one fixed-size uint8 slab per rank to `dst` (a per-link-bound gather-to-root over
    xGMI: each peer uses one ~153 GB/s link, so slabs are the compact resident records, never
    expanded observation tensors)."""

    def __init__(self, slab_bytes, device, dst=0, group=None):
        self.dst, self.group = dst, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.send = torch.empty(slab_bytes, dtype=torch.uint8, device=device)
        self.recv = [torch.empty_like(self.send) for _ in range(self.world)] if self.rank == dst else None

    def gather(self):
        """Collective; returns the list of per-rank slabs on dst, None elsewhere."""
        dist.gather(self.send, self.recv, dst=self.dst, group=self.group)
        return self.recv
