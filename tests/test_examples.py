"""The example learner runs end to end on the resident collection loop (a smoke test: a few updates, finite loss)."""
import importlib.util
import math
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_resident_dqn_example_runs():
    spec = importlib.util.spec_from_file_location("train_dqn_resident", os.path.join(ROOT, "examples", "train_dqn_resident.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    out = m.main(["--num-envs", "256", "--board", "8", "--buffer-size", "20000", "--batch-size", "128", "--updates", "12",
                  "--collect-per-update", "3", "--target-every", "5", "--max-steps-per-episode", "20"])
    assert out["env_steps"] > 256 * 30 and out["episodes"] >= 256 and out["ring_fill"] == 20000 or out["ring_fill"] == out["env_steps"]
    assert all(math.isfinite(x) for x in out["loss"]) and len(out["loss"]) >= 1
