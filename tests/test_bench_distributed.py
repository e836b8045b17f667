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
