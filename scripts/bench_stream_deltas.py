#!/usr/bin/env python3
"""The per-turn broadcast of the gRPC surface (createStreamUpdate, server.go:632-777) for B boards: gvec_stream_deltas on the
device against reading the boards back (gvec_read_state + gvec_player_visibility), which is what building the updates on
the host needs.   usage: scripts/bench_stream_deltas.py [B]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import generalsreinforcementlearning_amd as g
from generalsreinforcementlearning_amd._lib import check

B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
eng = g.VecEngine(B, 20, 20, 4, auto_reset=True, stream=torch.cuda.current_stream().cuda_stream)
eng.reset_generated(1)
eng.build_board_pool(1024, 2)
eng.rollout(120, 3, 0, fused=False, want_stats=False)
cap = eng.L.gvec_stream_delta_cap(eng.h)
kind = torch.zeros(B, dtype=torch.uint8, device="cuda")
count = torch.zeros(B, dtype=torch.int32, device="cuda")
upd = torch.zeros((B, cap), dtype=torch.int64, device="cuda")
def dev():
    check(eng.L.gvec_stream_deltas(eng.h, 0, kind.data_ptr(), count.data_ptr(), upd.data_ptr(), 1), "gvec_stream_deltas")
dev(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): dev()
e1.record(); torch.cuda.synchronize()
k, c = kind.cpu().numpy(), count.cpu().numpy()
out = {"boards": B, "kernel_ms": e0.elapsed_time(e1) / 20, "delta_envs": int((k == 1).sum()), "full_state_envs": int((k == 2).sum()),
       "mean_updates_per_delta": float(c[k == 1].mean()) if (k == 1).any() else 0.0, "cap": cap,
       "bytes_a_consumer_needs_per_env": float(8 * c[k == 1].mean() + 5) if (k == 1).any() else None}
t0 = time.perf_counter()
for _ in range(3): eng.stream_deltas(0)
out["host_call_ms_pageable_full_buffers"] = (time.perf_counter() - t0) / 3 * 1e3
eng.stream_deltas_packed(0)
t0 = time.perf_counter()
for _ in range(5): eng.stream_deltas_packed(0)
out["host_call_ms_packed_pinned"] = (time.perf_counter() - t0) / 5 * 1e3
t0 = time.perf_counter()
for _ in range(2):
    eng.game_state(fields=("army", "owner", "type", "changed", "vis_changed"))
    eng.compute_player_visibility(0)
out["read_boards_back_ms"] = (time.perf_counter() - t0) / 2 * 1e3
print(json.dumps(out))
