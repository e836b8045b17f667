"""Uniform draws without replacement on the device, shared by the replay memories (env_pool.DeviceReplayBuffer,
experience.RecordReplayRing): `random.sample(range(size), k)`'s contract - k distinct values, ValueError when k > size."""


def distinct_indices(torch, size, k, device, generator):
    if k > size or k < 0:
        raise ValueError("Sample larger than population or is negative")          # random.sample's own words
    if size <= (1 << 16) or 4 * k >= size:
        return torch.randperm(size, device=device, generator=generator)[:k]
    got = torch.empty(0, dtype=torch.int64, device=device)                        # big ring, small draw: draw, drop repeats, top up
    while got.numel() < k:
        both = torch.cat([got, torch.randint(size, (2 * (k - got.numel()) + 16,), device=device, generator=generator)])
        uniq, inverse = torch.unique(both, return_inverse=True)
        first = torch.full((uniq.numel(),), both.numel(), dtype=torch.int64, device=device).scatter_reduce_(
            0, inverse, torch.arange(both.numel(), device=device), reduce="amin")
        got = both[first.sort().values]                                           # first occurrences, in drawing order
    return got[:k]
