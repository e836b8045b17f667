#!/usr/bin/env python3
"""The consumer side of the exchange step: gvec_expand_experience_records on k records (20x20 4P, or GVEC_BOARD=W,H,P), bytes
written per second.   usage: scripts/bench_expand.py [k ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import generalsreinforcementlearning_amd as g
from generalsreinforcementlearning_amd.experience import RecordExpander

BW, BH, BP = (int(v) for v in os.environ.get("GVEC_BOARD", "20,20,4").split(","))
for k in [int(v) for v in sys.argv[1:]] or [4096, 32768]:
    eng = g.VecEngine(k, BW, BH, BP, auto_reset=True, stream=torch.cuda.current_stream().cuda_stream)
    eng.reset_generated(1)
    eng.build_board_pool(64, 2)
    eng.rollout(50, 3, 0, fused=True, want_stats=False)
    eng.experience_begin()
    eng.record_agent_actions(True)
    eng.rollout(1, 4, 0, fused=False, want_stats=False)
    lay = eng.experience_record_layout()
    slab = torch.empty(k * eng.experience_record_bytes(), dtype=torch.uint8, device="cuda")
    eng.experience_records(slab.data_ptr())
    ex = RecordExpander(lay, k, "cuda:0")
    ex.expand(slab)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        slots = ex.expand(slab)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    present = int((slots["meta"][:, 0] != 0).sum().item())
    written = k * lay["mp"] * ((2 * 9 * 4 + 4) * lay["stride"] + 32)
    print(json.dumps({"board": [BW, BH, BP], "records": k, "experiences": present, "ms": ms, "written_gb": written / 1e9, "write_gbs": written / ms / 1e6,
                      "experiences_per_s": present / ms * 1e3, "record_bytes_in": k * lay["record_dw"] * 4}))
    eng.close()
