import sys
sys.path.insert(0, ".")
import numpy as np, torch
import generalsreinforcementlearning_amd as g
from generalsreinforcementlearning_amd._lib import check
B, W, H, P = 262144, 20, 20, 4
eng = g.VecEngine(B, W, H, P, stream=torch.cuda.current_stream().cuda_stream)
eng.reset_generated(1)
eng.rollout(50, 1, 0, fused=True, want_stats=False)
L = eng.L
obs1 = torch.empty(B * 9 * 400, dtype=torch.float32, device="cuda")
obs4 = torch.empty(B * 4 * 9 * 400, dtype=torch.float32, device="cuda")
rew = torch.empty(B * 4, dtype=torch.float32, device="cuda")
done = torch.empty(B, dtype=torch.uint8, device="cuda")
bits = torch.empty(B * 4 * eng.mask_bytes, dtype=torch.uint8, device="cuda")
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
t = timeit(lambda: check(L.gvec_observe(eng.h, 0, obs1.data_ptr(), 1)))
print(f"observe(player 0): {t:.3f} ms  -> {B/t/1e3:.1f} M obs/s, write {B*9*400*4/t/1e6:.0f} GB/s")
t = timeit(lambda: check(L.gvec_observe(eng.h, -1, obs4.data_ptr(), 1)), 5)
print(f"observe(all 4 players): {t:.3f} ms -> {4*B/t/1e3:.1f} M obs/s, write {4*B*9*400*4/t/1e6:.0f} GB/s")
t = timeit(lambda: check(L.gvec_experience_begin(eng.h)))
print(f"experience_begin (snapshot): {t:.3f} ms")
t = timeit(lambda: check(L.gvec_experience_rewards(eng.h, rew.data_ptr(), done.data_ptr(), 1)))
print(f"experience_rewards: {t:.3f} ms -> {B/t/1e3:.1f} M env/s")
t = timeit(lambda: check(L.gvec_serializer_mask(eng.h, bits.data_ptr(), 1)))
print(f"serializer_mask: {t:.3f} ms")
