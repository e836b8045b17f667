import time, os, sys, torch
sys.path.insert(0, os.getcwd())
import generalsreinforcementlearning_amd as g
B=262144
stream = torch.cuda.current_stream()
eng = g.VecEngine(B, 20, 20, 4, auto_reset=True, stream=stream.cuda_stream)
eng.reset_generated(1); eng.build_board_pool(4096, 2); eng.record_agent_actions(True)
buf = torch.empty(4096*eng.experience_record_bytes(), dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()
def t(name, f, n=300):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    dt=(time.perf_counter()-t0)/n*1e6; torch.cuda.synchronize(); print(f"{name:32s}{dt:8.1f} us host time per call", flush=True)
t("rollout(1)", lambda: eng.rollout(1, 1, 0, fused=False, want_stats=False), 100)
t("experience_begin_range", lambda: eng.experience_begin_range(0, 4096))
t("experience_records", lambda: eng.experience_records(buf.data_ptr(), None, 0, 4096, 0))
ev=[None]
def sidework():
    side.wait_stream(stream)
    with torch.cuda.stream(side):
        ev[0]=side.record_event()
t("side.wait_stream+record_event", sidework)
t("stream.wait_event", lambda: stream.wait_event(ev[0]))
