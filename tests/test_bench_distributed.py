"""bench.py's multi-rank choreography under the driver's launch line, on CPU with gloo (world 2):
`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 ... --rehearse-cpu`."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_world2_gloo_rehearsal():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--envs-per-gpu", "1024", "--gather-envs", "64", "--gather-every", "1", "--rehearse-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["rehearsal"] is True and d["n_gpus"] == 2 and d["steps"] == 5 and d["gathers"] == 5 and d["scaling"] == "weak"


def test_bench_world2_strong_scaling_rehearsal():
    """--total-envs: BASELINE configs[3] as stated - the boards are SHARDED across the ranks (sharding.shard_range)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--total-envs", "2049", "--gather-envs", "64", "--gather-every", "1", "--rehearse-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["scaling"] == "strong" and d["total_envs"] == 2049 and d["envs_rank0"] == 1025 and d["gathers"] == 3


def test_bench_gathers_on_every_kth_step_by_default():
    """BASELINE configs[3]: 'RCCL gather each K turns' - the default K is 4: 9 timed steps carry 2 gathers."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "9", "--warmup", "0",
           "--envs-per-gpu", "512", "--gather-envs", "32", "--rehearse-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["steps"] == 9 and d["gathers"] == 2


def test_bench_refuses_mismatched_world():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-cpu", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def _plain_env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def test_bench_plain_command_line_spawns_its_own_ranks():
    """`python bench.py --gpus 2 ...` with NO launcher around it (the shape of the driver's single-GPU line): the parent,
    which never touches the GPU, starts the ranks itself and relays rank 0's ONE JSON line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--envs-per-gpu", "256", "--gather-envs", "16", "--gather-every", "2", "--rehearse-cpu"],
                       capture_output=True, text=True, timeout=240, cwd=ROOT, env=_plain_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["rehearsal"] is True and d["n_gpus"] == 2 and d["steps"] == 4 and d["gathers"] == 2 and d["total_envs"] == 512


def test_bench_plain_command_line_strong_scaling_with_a_total_that_does_not_divide():
    """2,049 boards over 2 ranks: shards of 1,025 and 1,024; every rank gathers the same slab (bounded by the smallest
    shard) and tags its records with its shard's first global env id."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0",
                        "--total-envs", "2049", "--gather-envs", "4096", "--gather-every", "1", "--rehearse-cpu"],
                       capture_output=True, text=True, timeout=240, cwd=ROOT, env=_plain_env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["envs_rank0"] == 1025 and d["gather_envs_per_rank"] == 1024 and d["gathers"] == 3


def test_bench_plain_command_line_propagates_a_rank_failure():
    """A rank that dies must not leave the others waiting in a barrier: the parent ends them and returns non-zero."""
    env = dict(_plain_env(), GVEC_BENCH_FAIL_RANK="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0",
                        "--envs-per-gpu", "64", "--rehearse-cpu"], capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode != 0


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_two_ranks_with_real_engines_on_one_gpu():
    """The N > 1 choreography with REAL engines (own ranks, env-sharded boards, record kernels on every K-th step, gather,
    GPU-side expansion on rank 0, played-turn counters summed over ranks): two ranks share this box's one GPU
    (GVEC_BENCH_SAME_DEVICE=1: gloo carries what RCCL would - RCCL refuses two ranks on one device).  A choreography
    check, not a measurement: the line says so."""
    env = dict(_plain_env(), GVEC_BENCH_SAME_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "4",
                        "--envs-per-gpu", "8192", "--gather-envs", "512", "--gather-every", "3", "--pool", "256", "--prewarm-s", "0.05",
                        "--no-cpu-baseline", "--no-fused"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # rank 0 prints ONE line, nothing else reaches stdout
    d = json.loads(lines[0])
    assert d["rehearsal_same_device"] is True and d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["total_envs"] == 16384
    assert d["board_launches"] == 16384 * 12 and 0.99 * d["board_launches"] <= d["env_steps_played"] <= d["board_launches"]
    g_ = d["experience_gather"]
    assert g_["ranks_seen"] == [0, 1] and g_["records_per_rank_per_gather"] == 512 and g_["experiences_decoded_last_step"] > 512


@pytest.mark.gpu
def test_bench_three_ranks_strong_scaling_with_real_engines_on_one_gpu():
    """BASELINE configs[3]'s form - a fixed total sharded across the ranks - with a total that does not divide: shards of
    5,462 / 5,461 / 5,461 boards, equal gather slabs, records tagged with global env ids of all three shards."""
    env = dict(_plain_env(), GVEC_BENCH_SAME_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "8", "--warmup", "2",
                        "--total-envs", "16384", "--gather-envs", "8192", "--gather-every", "2", "--pool", "256", "--prewarm-s", "0.05",
                        "--no-cpu-baseline", "--no-fused"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    assert d["scaling"] == "strong" and d["config"]["total_envs"] == 16384 and d["config"]["envs_per_gpu"] == 5462
    assert d["board_launches"] == 16384 * 8 and d["env_steps_played"] <= d["board_launches"]
    g_ = d["experience_gather"]
    assert g_["records_per_rank_per_gather"] == 5461 and g_["ranks_seen"] == [0, 1, 2]


@pytest.mark.gpu
def test_bench_overlapped_record_chain_changes_nothing_but_the_time():
    """--record-overlap 1 (the sampled slice's snapshot -> turn -> records on a second compute stream beside the rest of the
    batch) against --record-overlap 0 (everything in line): the same games - final headers and masks, played / aborted /
    finished counts - and the same gathered experiences.  World size 1 over real RCCL (--force-dist)."""
    outs = []
    for overlap in ("1", "0"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "30", "--warmup", "6",
                            "--envs-per-gpu", "20000", "--gather-envs", "1500", "--gather-every", "3", "--pool", "64", "--prewarm-s", "0",
                            "--width", "12", "--height", "12", "--players", "3", "--record-overlap", overlap, "--fingerprint",
                            "--no-cpu-baseline", "--no-fused"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=_plain_env())
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, r.stdout
        outs.append(json.loads(lines[0]))
    a, b = outs
    for k in ("state_fingerprint", "env_steps_played", "aborted_turns", "games_finished", "board_launches"):
        assert a[k] == b[k], (k, a[k], b[k])
    assert a["experience_gather"]["experiences_decoded_last_step"] == b["experience_gather"]["experiences_decoded_last_step"] > 1500
    assert a["games_finished"] >= 0 and a["env_steps_played"] > 0.99 * a["board_launches"]
