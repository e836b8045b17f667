import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import generalsreinforcementlearning_amd as g
B, W, H, P = 262144, 20, 20, 4
eng = g.VecEngine(B, W, H, P)
eng.reset_generated(1)
acts = eng.agent_actions(1)
for want_mask in (False, True):
    eng.step(acts, want_mask=want_mask)
    t0 = time.perf_counter(); n = 5
    for _ in range(n):
        eng.step(acts, want_mask=want_mask)
    dt = (time.perf_counter() - t0) / n
    print(f"host-buffer gvec_step want_mask={want_mask}: {dt*1e3:.2f} ms/step -> {B/dt/1e6:.1f} M env-steps/s (PCIe-inclusive, pageable numpy buffers)")
pacts = eng.pinned(acts.shape, acts.dtype)
pacts[...] = acts
for want_mask in (False, True):
    eng.step(pacts, want_mask=want_mask, pinned=True)
    t0 = time.perf_counter(); n = 5
    for _ in range(n):
        eng.step(pacts, want_mask=want_mask, pinned=True)
    dt = (time.perf_counter() - t0) / n
    print(f"host-buffer gvec_step want_mask={want_mask}: {dt*1e3:.2f} ms/step -> {B/dt/1e6:.1f} M env-steps/s (PCIe-inclusive, PINNED buffers from gvec_host_alloc)")

