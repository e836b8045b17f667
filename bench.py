#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched Generals.io turn engine on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` every process is one
     rank; as a PLAIN command the parent - which never touches the GPU - starts the N ranks itself as child processes
     with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and relays rank 0's JSON line)

One *step* = one engine turn for every board of the batch: one launch of the HIP step
kernel (on-device random agent sampling from the legal mask -> movement/combat ->
eliminations -> production -> stats -> fog of war -> legal-mask emission), with the whole
board state read from and written back to HBM.  Workload = BASELINE.json's metric config:
20x20 boards, 4 players, fog of war on, legal mask on, 262,144 boards per GPU (weak scaling:
every rank owns its own boards; no collective on the data path).  Finished games are re-dealt
from a pre-generated board pool (auto-reset).

Rank 0 prints ONE JSON line.  `roofline` prices the step kernel against HBM bandwidth with
SURVEY 8(d)'s algorithmic bytes per env-step; `cpu_baseline` times the CPU oracle (a C
restatement of the Go engine: kind "port") on the host cores on a bounded sample.

Scaling modes (N > 1): the default is weak scaling (--envs-per-gpu boards on every rank);
`--total-envs T` is strong scaling - BASELINE.json configs[3] as stated, 262,144 boards sharded
across the ranks with sharding.shard_range - and `config.workload` says which one ran.
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0  # what a float4 copy reaches on this part (same guide): the practical ceiling of any streaming kernel


def algorithmic_bytes(w, h, p, mask=True):
    """state read 8 B/tile + write 8 B/tile, actions, player stats, packed masks, scalars."""
    return 16 * w * h + 4 * p + 12 * p + (p * ((4 * w * h + 7) // 8) if mask else 0) + 16


def step_kernel_name(w, h, p):
    """The step_kernel instantiation gvec_create picks for this board size (gvec_kernels.hip pick_variant)."""
    maxp = next(m for m in (2, 4, 8) if m >= p)
    nslot = next(n for n in (1, 2, 4, 7, 10, 16) if n * 64 >= w * h)
    odd = w * h <= 32 * (2 * nslot - 1)
    return f"gvec::step_kernel<{maxp}, {nslot}, true, {'true' if odd else 'false'}>"


def kernel_source_hash():
    """Identifies the kernel build a PMC traffic figure belongs to (profiles/pmc_traffic.json is stamped with it)."""
    d = os.path.join(ROOT, "generalsreinforcementlearning_amd", "csrc")
    hh = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            hh.update(open(os.path.join(d, f), "rb").read())
    return hh.hexdigest()[:16]


def step_kernel_isa_hash(w=20, h=20, p=4):
    """sha256 of the step kernel's generated gfx950 instructions (csrc/build/*.s, kept by the build): unlike the source
    hash it survives edits that do not touch this kernel's code (the C ABI file, other kernels, comments).  None when the
    build's ISA listing is not there."""
    import re
    path = os.path.join(ROOT, "generalsreinforcementlearning_amd", "csrc", "build", "gvec_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not os.path.exists(path):
        return None
    maxp = next(m for m in (2, 4, 8) if m >= p)
    nslot = next(n for n in (1, 2, 4, 7, 10, 16) if n * 64 >= w * h)
    odd = 1 if w * h <= 32 * (2 * nslot - 1) else 0
    sym = f"_ZN4gvec11step_kernelILi{maxp}ELi{nslot}ELb1ELb{odd}EEEvNS_8StepArgsE"
    hh, inside = hashlib.sha256(), False
    for line in open(path):
        if line.startswith(sym + ":"):
            inside = True
            continue
        if inside:
            if line.startswith(".Lfunc_end"):
                break
            t = line.split(";")[0].strip()
            if t and not t.startswith((".", "//")) and not re.match(r"^\.?L[A-Za-z_0-9]*:$", t):
                hh.update(re.sub(r"\.LBB\d+_", ".LBB_", t).encode() + b"\n")   # block labels carry the function's index in the file
    return hh.hexdigest()[:16] if inside else None


def go_probe():
    """The Go engine itself can only be timed where a Go toolchain AND the reference module exist
    (bench/go/step_bench_test.go; GRL_REFERENCE_DIR = a checkout of the reference).  Probed, never assumed."""
    go = shutil.which("go")
    if go is None:
        return {"go": None, "status": "Go unavailable on this host (shutil.which('go') is None)"}
    try:
        ver = subprocess.run([go, "version"], capture_output=True, text=True, timeout=20).stdout.strip()
    except Exception as e:  # a broken toolchain is an absent one
        return {"go": go, "status": f"Go unavailable on this host (`go version` failed: {e})"}
    ref = os.environ.get("GRL_REFERENCE_DIR")
    if not ref or not os.path.exists(os.path.join(ref, "go.mod")):
        return {"go": go, "version": ver, "status": "Go present; set GRL_REFERENCE_DIR to a checkout of the reference to time Engine.Step "
                                                    "with bench/go/step_bench_test.go (the reference does not travel with this repository)"}
    try:
        dst = os.path.join(ref, "internal", "game", "zz_gvec_step_bench_test.go")
        shutil.copy(os.path.join(ROOT, "bench", "go", "step_bench_test.go"), dst)
        try:
            r = subprocess.run([go, "test", "./internal/game/", "-run", "^$", "-bench", "BenchmarkEngineStep20x20P4", "-benchtime", "5s"],
                               cwd=ref, capture_output=True, text=True, timeout=300)
        finally:
            os.unlink(dst)
        line = next((l for l in r.stdout.splitlines() if l.startswith("BenchmarkEngineStep20x20P4")), None)
        if line is None:
            return {"go": go, "version": ver, "status": "go test produced no benchmark line", "stderr": r.stderr[-400:]}
        ns = float(line.split()[2])
        return {"go": go, "version": ver, "status": "ok", "engine_step_ns_per_op": ns, "env_steps_per_s_one_goroutine": 1e9 / ns, "line": line}
    except Exception as e:
        return {"go": go, "version": ver, "status": f"go benchmark failed: {e}"}


def cpu_baseline(w, h, p, fog, seed, budget_s=15.0):
    """Times the CPU oracle (test infrastructure) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import _harness as H
    import _oracle as O
    # the GPU box gives one GPU a 16-core CPU share whatever the affinity mask says
    cores = max(1, min(len(os.sched_getaffinity(0)), os.cpu_count() or 1, int(os.environ.get("GVEC_CPU_THREADS", "16"))))
    envs = 256 * cores
    sizes = [(w, h, p)] * envs
    army, owner, typ, ws, hs, ps = H.gen_boards(seed, sizes, w, h)
    ora = O.OracleBatch(envs, w, h, p, fog=fog)
    ora.reset(army, owner, typ, ws, hs, ps)
    ora.set_pool(1024, seed + 17)
    # the same work per env-step as the GPU leg: agent sampling from the legal mask, Engine.Step, then the players' legal
    # masks packed (what the step kernel emits every turn)
    bits = np.zeros((envs, p, ora.mask_bytes), np.uint8)
    # calibrate, then run ~budget_s of work
    t0 = time.perf_counter()
    s0 = ora.rollout(4, seed, 0, threads=cores, legal_bits=bits)
    dt = max(time.perf_counter() - t0, 1e-6)
    turns = int(max(8, min(20000, budget_s / (dt / 4))))
    t0 = time.perf_counter()
    steps = ora.rollout(turns, seed, 0, threads=cores, legal_bits=bits)
    dt = time.perf_counter() - t0
    one = O.OracleBatch(256, w, h, p, fog=fog)
    one.reset(army[:256], owner[:256], typ[:256], ws[:256], hs[:256], ps[:256])
    t1 = time.perf_counter()
    s1 = one.rollout(max(8, turns // 8), seed, 0, threads=1, legal_bits=bits[:256])
    d1 = time.perf_counter() - t1
    del s0
    return {"value": steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "single_thread_value": s1 / d1,
            "sample": f"{envs} boards {w}x{h} P{p} fog={'on' if fog else 'off'} x {turns} turns, oracle random agent + "
                      f"Engine.Step restatement + packed legal masks every turn (the GPU leg's work), OpenMP over boards; {dt:.1f} s of CPU work",
            "note": "C restatement of the Go engine (oracle/generals_oracle.c), never the Go engine itself",
            "go_reference": go_probe()}


def shard_plan(args, world, rank):
    """-> (boards of this rank, boards of the whole job, "weak" | "strong", first global env id of this rank)."""
    if args.total_envs > 0:
        from generalsreinforcementlearning_amd.sharding import shard_range
        begin, n = shard_range(args.total_envs, world, rank)
        return n, args.total_envs, "strong", begin
    return args.envs_per_gpu, args.envs_per_gpu * world, "weak", rank * args.envs_per_gpu


def gather_slab_envs(args, world):
    """Records per rank per gather: the same on every rank (dist.gather needs equal slabs), so it is bounded by the
    SMALLEST shard - total // world under strong scaling with a total that does not divide."""
    smallest = (args.total_envs // world) if args.total_envs > 0 else args.envs_per_gpu
    return max(0, min(args.gather_envs, smallest))


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` as a plain command: this process has made no GPU call (torch is not even imported
    yet) and starts the N ranks as fresh child processes - never an exec of a process that touched the GPU.  Rank 0's
    stdout (the JSON line) is relayed; every other stream is inherited.  A rank that dies takes the others with it
    (they would wait in the next barrier forever): exact PIDs only."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            bad = [p for p in procs if p.poll() not in (None, 0)]
            if bad:
                rc = bad[0].returncode
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                break
            time.sleep(0.05)
        out = procs[0].stdout.read() if procs[0].stdout else ""
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
            rc = rc or p.returncode
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.stdout.write(out)
    sys.stdout.flush()
    return rc


class _NullEngine:
    """Stands in for VecEngine in --rehearse-cpu: same calls, no work (there is no CPU engine)."""

    def __init__(self, rec_bytes):
        self.rec = rec_bytes

    def rollout(self, *a, **k):
        return None

    def state_bytes_per_env(self):
        return self.rec

    def export_records(self, ptr, lo, n):
        return None

    def experience_begin_range(self, lo, n):
        return None

    def experience_records(self, ptr, actions=None, env_begin=0, n=None, env_id_base=0):
        return None

    def experience_record_bytes(self):
        return self.rec


def rehearse_cpu(args):
    """The N > 1 choreography of main() on CPU tensors with the gloo backend: rendezvous from the
    torchrun environment, per-step record gather to rank 0, barrier-bracketed timing, MAX over ranks,
    one JSON line from rank 0."""
    import torch
    import torch.distributed as dist
    from generalsreinforcementlearning_amd.sharding import RecordGather
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if os.environ.get("GVEC_BENCH_FAIL_RANK") == str(rank):   # tests: a rank that dies before the rendezvous
        raise SystemExit(3)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    B, total, mode, env_base = shard_plan(args, world, rank)
    eng = _NullEngine(64)
    ge = gather_slab_envs(args, world)
    rg = RecordGather(ge * eng.experience_record_bytes(), torch.device("cpu"), dst=0) if ge > 0 else None
    got = 0
    for k in range(args.warmup):
        eng.rollout(1, args.seed, 0, fused=False, want_stats=False)
    dist.barrier()
    t0 = time.perf_counter()
    K = max(1, args.gather_every)
    for k in range(args.steps):
        gathering = rg is not None and k % K == K - 1
        if gathering:
            eng.experience_begin_range(0, ge)
        eng.rollout(1, args.seed, 0, fused=False, want_stats=False)
        if gathering:
            eng.experience_records(rg.send.data_ptr(), None, 0, ge, env_base)
            rg.send.fill_((rank * 31 + k) % 251)
            out = rg.gather()
            if rank == 0:
                assert all(int(out[r][0]) == (r * 31 + k) % 251 for r in range(world))
                got += 1
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"rehearsal": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "gathers": got,
                          "value": total * args.steps / float(t.item()), "scaling": mode, "total_envs": total,
                          "envs_rank0": B, "gather_envs_per_rank": ge}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=262144, help="weak scaling: boards on every rank")
    ap.add_argument("--total-envs", type=int, default=0,
                    help="strong scaling: this many boards in total, sharded across the ranks (BASELINE configs[3]: 262144)")
    ap.add_argument("--prewarm-s", type=float, default=0.75,
                    help="untimed clock-settle phase before the counted warmup: step launches for at least this many seconds")
    ap.add_argument("--width", type=int, default=20)
    ap.add_argument("--height", type=int, default=20)
    ap.add_argument("--players", type=int, default=4)
    ap.add_argument("--fog", type=int, default=1)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--pool", type=int, default=4096)
    ap.add_argument("--gather-envs", type=int, default=4096,
                    help="N>1 only: compact records per rank gathered to rank 0 each step (experience slab; 0 = off)")
    ap.add_argument("--gather-every", type=int, default=4,
                    help="N>1 only: the record gather runs on every K-th step (BASELINE configs[3]: 'RCCL gather each K turns')")
    ap.add_argument("--record-overlap", type=int, default=0,
                    help="N>1 only: 1 = on a gathering step the sampled slice (snapshot -> its turn -> record kernels) runs on a second "
                         "compute stream beside the turn of the rest of the batch (gvec_rollout_range); 2 = also the boards after the slice "
                         "on a stream of their own; 0 (default) = everything in line on one stream.  Measured neutral: the three are within "
                         "run-to-run noise of each other (DESIGN.md section 8)")
    ap.add_argument("--fingerprint", action="store_true", help="diagnostics: add a sha256 of the final headers and legal masks to the line")
    ap.add_argument("--gather-mode", type=int, default=0,
                    help="diagnostics: 1 = record kernels only (no collective), 2 = collective on the compute stream (no side stream)")
    ap.add_argument("--mixed", action="store_true",
                    help="BASELINE configs[4]: env i gets a 10x10 / 15x15 / 20x20 board with 2 + i%%3 players in one padded batch "
                         "(a parity-test configuration, not the bench line)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the record-gather choreography even at world size 1: "
                         "exercises the N>1 code path (side stream, events, experience records, decode) on a single GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="control-flow rehearsal of the multi-rank path on CPU (gloo, no GPU, no engine work): "
                         "used by tests/test_bench_distributed.py; prints a line marked \"rehearsal\": true, never a result")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a plain `python bench.py --gpus N`: no launcher made the ranks, so this (GPU-less) process does
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    if args.rehearse_cpu:
        return rehearse_cpu(args)
    import generalsreinforcementlearning_amd as g

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # diagnostics for a ONE-GPU box (never a result: the line says "rehearsal_same_device"): GVEC_BENCH_SAME_DEVICE=1 lets the N
    # ranks share device 0 - RCCL refuses two ranks on one GPU, so gloo carries the control traffic and the record slabs
    # go through host memory - to run the whole N > 1 choreography (own ranks, real engines, record kernels, decode)
    same_dev = os.environ.get("GVEC_BENCH_SAME_DEVICE") == "1"
    if same_dev:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    use_gloo = same_dev or os.environ.get("GVEC_BENCH_DIST_INIT") == "gloo"
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher made WORLD_SIZE={world} ranks")
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        # librccl writes a five-line banner (versions, hostname, library path) to STDOUT when its first communicator comes
        # up: stdout carries rank 0's ONE JSON line and nothing else, so file descriptor 1 points at stderr until RCCL is up
        sys.stdout.flush()
        saved_fd1 = os.dup(1)
        os.dup2(2, 1)
        if use_gloo:
            dist.init_process_group("gloo")   # diagnostics only (scripts/rccl_tax.sh, GVEC_BENCH_SAME_DEVICE)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL
        # RCCL builds a communicator's channels and loads its kernels at the FIRST collective: ~16 ms during which the GPU
        # sits idle.  Rounds 1-2 paid that inside the barrier that brackets the timed region and then measured the clock
        # ramp that follows an idle gap (first 50 launches 270 us, next 50 235 us, then the plain 220 us:
        # profiles/r03_rccl_tax.json) - the "5-7 % RCCL tax" of DESIGN.md section 8.  Pay it here, before the
        # clock-settle phase; the bracketing barriers are then ~20 us collectives.
        dist.barrier()
        torch.cuda.synchronize()
        sys.stdout.flush()
        os.dup2(saved_fd1, 1)
        os.close(saved_fd1)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream()

    W, H, P = args.width, args.height, args.players
    B, total_envs, scaling, env_base = shard_plan(args, world, rank)
    eng = g.VecEngine(B, W, H, P, fog_of_war=bool(args.fog), device=local_rank, auto_reset=True, stream=stream.cuda_stream)
    if args.mixed:
        import numpy as np
        side = np.array([10, 15, 20], np.int32)
        ws, ps = side[np.arange(B) % 3], (2 + np.arange(B) % 3).astype(np.int32)
        ws, ps = np.minimum(ws, min(W, H)), np.minimum(ps, P)
        eng.reset_generated(args.seed * 1000003 + rank, ws, ws, ps)
        pw, pp = side[np.arange(args.pool) % 3], (2 + np.arange(args.pool) % 3).astype(np.int32)
        eng.build_board_pool(args.pool, args.seed * 7919 + rank, np.minimum(pw, min(W, H)), np.minimum(pw, min(W, H)), np.minimum(pp, P))
    else:
        # "reset cost reported separately" - map generation (one thread per board), import into the resident
        # layout and performInitialSetup for all B boards, and the auto-reset pool; outside the timed region
        torch.cuda.synchronize()
        tr0 = time.perf_counter()
        eng.reset_generated(args.seed * 1000003 + rank)
        eng.synchronize()
        tr1 = time.perf_counter()
        eng.build_board_pool(args.pool, args.seed * 7919 + rank)
        eng.synchronize()
        tr2 = time.perf_counter()
        reset_cost = {"reset_generated_ms": (tr1 - tr0) * 1e3, "boards": B, "boards_per_s": B / (tr1 - tr0),
                      "pool_build_ms": (tr2 - tr1) * 1e3, "pool_boards": args.pool,
                      "note": "gvec_reset_generated = on-device map generator + import + performInitialSetup (full stats, full fog, game-over "
                              "check) for every board; a finished board re-dealt from the pool costs its step nothing extra (it replaces the turn)"}
    seed = args.seed

    if args.mixed:
        reset_cost = None
    rgs, side, slab_free = None, None, None
    if dist is not None and args.gather_envs > 0:
        # the one real exchange step of the path: compact EXPERIENCE RECORDS -> rank 0 (the StreamAggregator side,
        # internal/grpc/gameserver/stream_aggregator.go:75-155), expanded there (experience.decode_records).
        # Two slabs alternate so step k+1's records never wait for step k's gather on the side stream.
        from generalsreinforcementlearning_amd.sharding import RecordGather
        ge = gather_slab_envs(args, world)
        if use_gloo:
            class _HostStagedGather(RecordGather):   # gloo has no CUDA gather: device slab -> pinned host -> gloo -> device slabs on rank 0
                def __init__(self, nbytes):
                    super().__init__(nbytes, torch.device("cpu"), dst=0)
                    self.h_send, self.h_recv = self.send.pin_memory(), self.recv
                    self.send = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                    self.recv = [torch.empty_like(self.send) for _ in range(self.world)] if self.rank == 0 else None

                def gather(self):
                    self.h_send.copy_(self.send)
                    dist.gather(self.h_send, self.h_recv, dst=0)
                    if self.rank == 0:
                        for d, hsrc in zip(self.recv, self.h_recv):
                            d.copy_(hsrc)
                    return self.recv
            rgs = [_HostStagedGather(ge * eng.experience_record_bytes()) for _ in range(2)] if ge > 0 else None
        else:
            rgs = [RecordGather(ge * eng.experience_record_bytes(), dev, dst=0) for _ in range(2)] if ge > 0 else None
        slab_free = [None, None]
        side = torch.cuda.Stream()
        rec_stream = torch.cuda.Stream()           # the sampled slice's own compute stream (--record-overlap)
        rest_stream = torch.cuda.Stream()          # --record-overlap 2: the boards after the slice
        if rgs is not None:
            # the gather's send/recv channels connect lazily too: one untimed gather per slab, on the side stream it will use
            with torch.cuda.stream(side):
                for rg_ in rgs:
                    rg_.send.zero_()
                    rg_.gather()
            torch.cuda.synchronize()

    K = max(1, args.gather_every)

    def one_step_overlapped(k):
        """A gathering step with the sampled slice on its own stream: snapshot -> the slice's turn (its moves recorded) ->
        record kernels run BESIDE the turn of the other boards (disjoint envs: gvec_rollout_range), so the three small,
        latency-bound kernels (~30 us in line) hide behind the 0.2-ms launch; the next step waits for both."""
        lo = ((k // K) * ge) % max(1, B - ge + 1)
        i = (k // K) & 1
        rec_stream.wait_stream(stream)                                 # the previous turn is done
        if slab_free[i] is not None:
            rec_stream.wait_event(slab_free[i])                        # slab i was last read by the gather before the previous one
        eng.set_stream(rec_stream.cuda_stream)
        eng.experience_begin_range(lo, ge)
        eng.record_agent_actions(True)
        eng.rollout_range(lo, ge, 1, seed, 0)
        eng.record_agent_actions(False)
        eng.experience_records(rgs[i].send.data_ptr(), None, lo, ge, env_base)
        eng.set_stream(stream.cuda_stream)
        done = rec_stream.record_event()
        # the boards before and after the slice: two launches that would drain and refill the GPU one after the other on
        # one stream - the second one goes to its own stream and fills the machine together with the first
        tail_done = None
        if lo + ge < B:
            if lo > 0 and args.record_overlap >= 2:
                rest_stream.wait_stream(stream)
                eng.set_stream(rest_stream.cuda_stream)
                eng.rollout_range(lo + ge, B - lo - ge, 1, seed, 0)
                eng.set_stream(stream.cuda_stream)
                tail_done = rest_stream.record_event()
            else:
                eng.rollout_range(lo + ge, B - lo - ge, 1, seed, 0)
        if lo > 0:
            eng.rollout_range(0, lo, 1, seed, 0)
        if tail_done is not None:
            stream.wait_event(tail_done)
        stream.wait_event(done)                                        # the next turn touches the slice again
        side.wait_event(done)
        with torch.cuda.stream(side):
            rgs[i].gather()                                            # RCCL gather over xGMI, overlapped with the next steps
            slab_free[i] = side.record_event()

    def one_step(k):
        gathering = rgs is not None and k % K == K - 1
        if gathering and args.record_overlap and args.gather_mode == 0:
            return one_step_overlapped(k)
        if gathering:
            lo = ((k // K) * ge) % max(1, B - ge + 1)
            eng.experience_begin_range(lo, ge)                    # captureStateForExperience for the sampled slice
            eng.record_agent_actions(True)                        # the record needs the moves the device agent plays in this step
        eng.rollout(1, seed, 0, fused=False, want_stats=False)
        if gathering:
            eng.record_agent_actions(False)
            i = (k // K) & 1
            if slab_free[i] is not None:
                stream.wait_event(slab_free[i])                   # slab i was last read by the gather before the previous one
            eng.experience_records(rgs[i].send.data_ptr(), None, lo, ge, env_base)   # compute stream, after this step's kernel
            if args.gather_mode == 1:
                return
            if args.gather_mode == 2:
                rgs[i].gather()
                return
            side.wait_stream(stream)
            with torch.cuda.stream(side):
                rgs[i].gather()                                    # RCCL gather over xGMI, overlapped with the next steps
                slab_free[i] = side.record_event()

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # clock-settle phase (untimed, not counted as warmup): a cold MI355X needs a few hundred ms of load before its
    # clocks and HBM settle - a 20-step run measured straight after the reset reads ~9 % slower than a 200-step one
    prewarm_steps, tp = 0, time.perf_counter()
    while time.perf_counter() - tp < args.prewarm_s:
        for _ in range(16):
            eng.rollout(1, seed + 7, 0, fused=False, want_stats=False)
        prewarm_steps += 16
        torch.cuda.synchronize()
    prewarm_s = time.perf_counter() - tp
    for k in range(args.warmup):
        one_step(k)
    sync_all()
    played0 = eng.counters()          # H_CNT_* summed over the boards: one reduction launch + read-back, outside the events
    sync_all()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for k in range(args.steps):
        one_step(args.warmup + k)
    ev1.record(stream)
    sync_all()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / max(1, args.steps)  # HIP events on the launch stream
    played1 = eng.counters()
    played = {k: played1[k] - played0[k] for k in played1}   # turns the engines actually played inside the timed region
    if dist is not None:
        rdev = torch.device("cpu") if use_gloo else dev
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([played["env_steps"], played["aborted_turns"], played["games_finished"]], dtype=torch.int64, device=rdev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        played = dict(zip(("env_steps", "aborted_turns", "games_finished"), (int(v) for v in c.tolist())))

    fused = None
    if not args.no_fused:
        kf = max(1, min(args.steps, 64))
        eng.rollout(kf, seed + 1, 0, fused=True, want_stats=False)
        torch.cuda.synchronize()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record(stream)
        eng.rollout(kf, seed + 2, 0, fused=True, want_stats=False)
        f1.record(stream)
        torch.cuda.synchronize()
        fused = {"turns_per_launch": kf, "env_steps_per_s_per_gpu": B * kf / (f0.elapsed_time(f1) / 1e3),
                 "note": "same turns with board state kept in registers/LDS across one launch (gvec_rollout fused=1)"}

    gathered = None
    if rgs is not None and rank == 0:
        # the consumer side, outside the timed region: expand the last gathered slabs into Experience fields
        import bisect
        from generalsreinforcementlearning_amd.experience import RecordExpander
        torch.cuda.synchronize()
        lay = eng.experience_record_layout()
        n_g = (args.warmup + args.steps) // K                      # gathers so far; the last one filled slab (n_g - 1) & 1
        last = rgs[(n_g - 1) & 1].recv if n_g > 0 else []
        # the GPU-side consumer (gvec_expand_experience_records): records -> StateToTensor x2 + action mask + scalars, on this
        # GPU, into buffers a consumer allocates once; timed on its second run (the first loads the kernel)
        n_exp, envs_seen, expand_ms = 0, [], None
        if last:
            ex = RecordExpander(lay, ge, dev)
            ex.expand(last[0])
            x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            x0.record()
            slots = [ex.expand(t_)["meta"].clone() for t_ in last]
            x1.record()
            torch.cuda.synchronize()
            expand_ms = x0.elapsed_time(x1)
            for m_ in slots:
                m_ = m_[m_[:, 0] != 0]
                n_exp += int(m_.shape[0])
                envs_seen += m_[:, 1].cpu().tolist()
        begins = [shard_plan(args, world, r)[3] for r in range(world)]   # first global env id of every rank
        gathered = {"record_bytes": eng.experience_record_bytes(), "records_per_rank_per_gather": ge,
                    "gather_every_steps": K,
                    "experiences_decoded_last_step": n_exp,
                    "expand_ms": expand_ms, "expand_written_bytes": len(last) * ge * lay["mp"] * ((2 * 9 * 4 + 4) * lay["stride"] + 32),
                    "ranks_seen": sorted({bisect.bisect_right(begins, int(e)) - 1 for e in envs_seen})}
    if rank == 0:
        n = world
        kernel_s = kernel_ms / 1e3
        # roofline.frac: the bytes one env-step MUST move by construction of the resident layout (header, mutable +
        # constant planes in, header + mutable planes out, narrow armies both ways, the legal masks out) - from the
        # library's own layout constants (gvec_step_traffic_bytes), never hard-coded - over the kernel's time over peak.
        tb = eng.step_traffic_bytes()
        comp = tb["read"] + tb["write"] + tb["mask"]
        achieved = comp * B / kernel_s / 1e9
        # contract_*: SURVEY 8(d)'s ALGORITHMIC bytes (8 B per tile each way with int32 armies and every list stored):
        # the kernel moves about 62 % of them, so this figure can pass 1 and is kept for comparison only.
        abytes = algorithmic_bytes(W, H, P, True)
        contract = abytes * B / kernel_s / 1e9
        # traffic: HBM bytes per launch from the rocprofv3 PMC passes of THIS kernel build (profiles/pmc_traffic.json carries
        # the hash of the kernel sources it was measured on; a figure from another build is dropped, not reported)
        traffic, traffic_note = None, "no PMC record for this kernel build"
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                isa = step_kernel_isa_hash(W, H, P)
                same_build = rec.get("kernel_source_hash") == kernel_source_hash() or (isa is not None and rec.get("step_kernel_isa_hash") == isa)
                if rec.get("envs") == B and rec.get("board") == [W, H, P] and same_build:
                    traffic, traffic_note = rec.get("hbm_bytes_per_launch"), rec.get("source")
            except Exception:
                traffic = None
        mode_txt = (f"{total_envs} boards sharded over {n} GPU(s) ({B} on rank 0; strong scaling)" if scaling == "strong"
                    else f"{B} boards/GPU (weak scaling)")
        if args.mixed:
            mode_txt += " MIXED 10x10/15x15/20x20 boards with 2-4 players padded to"
        launches = total_envs * args.steps
        out = {
            "metric": "env steps/sec (whole node), 20x20 4P fog-on; state bit-exact vs the C restatement of the Go engine "
                      "(the reference's own test vectors pass on both)",
            # value counts the turns the engines actually PLAYED in the timed region (the device's own H_CNT_STEPS counters,
            # summed over every board and rank): a step that re-deals a finished board plays no turn and is not counted
            "value": played["env_steps"] / elapsed,
            "unit": "env-steps/s",
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "int32", "data": "synthetic", "prewarm_s": round(prewarm_s, 3), "prewarm_steps": prewarm_steps,
            "env_steps_played": played["env_steps"], "board_launches": launches,
            "played_fraction": played["env_steps"] / max(1, launches),
            "aborted_turns": played["aborted_turns"], "games_finished": played["games_finished"],
            "board_launches_per_s": launches / elapsed,
            "config": {"workload": f"{mode_txt} x {W}x{H} {P}P fog-{'on' if args.fog else 'off'} + legal mask, on-device random "
                                   f"agent, auto-reset pool {args.pool}, 1 turn per launch",
                       "envs_per_gpu": B, "total_envs": total_envs, "board": [W, H, P], "parallelism": f"env-sharded x{n}",
                       "gather_envs_per_step": (ge if rgs is not None else 0),
                       "gather_every_steps": (K if rgs is not None else 0)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_note,
                         "traffic_gbs": (traffic / kernel_s / 1e9 if traffic else None),
                         "traffic_frac": (traffic / kernel_s / 1e9 / HBM_PEAK_GBS if traffic else None),
                         "copy_peak": HBM_COPY_GBS, "copy_frac": achieved / HBM_COPY_GBS,
                         "contract_bytes_per_env_step": abytes, "contract_gbs": contract, "contract_frac": contract / HBM_PEAK_GBS,
                         "kernel": step_kernel_name(W, H, P), "bytes_per_env_step": comp, "bytes_breakdown": tb,
                         "units_per_launch": B, "kernel_ms": kernel_ms,
                         "note": "achieved / frac: bytes the step kernel must move by construction of its resident layout "
                                 "(bytes_breakdown: read + write + mask per env-step, from gvec_step_traffic_bytes) x boards per launch / "
                                 "kernel time (HIP events on the launch stream) / 8 TB/s spec peak; copy_frac: the same over the 6.29 TB/s a "
                                 "float4 copy reaches on this part; traffic*: PMC-measured HBM bytes of this build; contract_*: SURVEY 8(d)'s "
                                 "7,280-B algorithmic figure (int32 armies, lists always stored), which this kernel undercuts - it can pass 1"},
        }
        if reset_cost:
            out["reset_cost"] = reset_cost
        if args.fingerprint:
            import ctypes as C
            hh = hashlib.sha256()
            for which, nbytes in ((0, B * 24 * 4), (3, B * P * eng.mask_bytes)):
                buf = (C.c_uint8 * nbytes)()
                g._lib.check(eng.L.gvec_read_buffer(eng.h, which, 0, nbytes, buf), "gvec_read_buffer")
                hh.update(bytes(buf))
            out["state_fingerprint"] = hh.hexdigest()[:24]
        if same_dev:
            out["rehearsal_same_device"] = True   # N ranks on ONE GPU over gloo: a choreography check, not a measurement
        if gathered:
            out["experience_gather"] = gathered
            out["roofline"]["kernel_ms_note"] = ("HIP-event span / steps on the compute stream: with the experience gather on it also holds the "
                                                 "snapshot and record kernels of the sampled slice (about 26 us per step at 4,096 records)")
        if fused:
            out["fused_rollout"] = fused
        if n == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(W, H, P, bool(args.fog), args.seed)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
