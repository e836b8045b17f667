import sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else ".")
import torch
from generalsreinforcementlearning_amd import _lib
from generalsreinforcementlearning_amd.vec_engine import VecEngine
from generalsreinforcementlearning_amd._lib import check
B, W, H, P = 262144, 20, 20, 4
obs4 = torch.empty(B * 4 * 9 * 400, dtype=torch.float32, device="cuda")
engs = []
for p in sys.argv[1:]:
    L = _lib.load_from(os.path.abspath(p))
    e = VecEngine(B, W, H, P, lib=L, stream=torch.cuda.current_stream().cuda_stream)
    e.reset_generated(1); e.rollout(50, 1, 0, fused=True, want_stats=False)
    engs.append((p, e, L))
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for r in range(3):
    for p, e, L in engs:
        t1 = timeit(lambda: check(L.gvec_observe(e.h, 0, obs4.data_ptr(), 1)), 10)
        t4 = timeit(lambda: check(L.gvec_observe(e.h, -1, obs4.data_ptr(), 1)))
        print(f"{p}: observe(1 player) {t1:.3f} ms = {B/t1/1e3:.0f} M/s {B*14400/t1/1e6:.0f} GB/s | all 4: {t4:.3f} ms = {4*B/t4/1e3:.0f} M tensors/s {4*B*14400/t4/1e6:.0f} GB/s")
