"""The driver's own bench command on the GPU box: one JSON line with the contract's fields, measured through the branch the library
reports, and not far from the committed figures (a regression guard, not a performance claim: the gates are loose)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_driver_command_prints_the_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extras",
                          "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                     # ONE JSON line
    r = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert r["metric"] == base["metric"] and r["unit"] == "drone-steps/s"
    assert r["n_gpus"] == 1 and r["steps"] == 20 and r["warmup"] == 5 and r["higher_is_better"] is True and r["scaling"] == "weak"
    assert r["dtype"] == "f32" and r["data"] == "synthetic" and r["vs_baseline"] is None
    assert "65536 envs x 8 drones" in r["config"]["workload"]
    assert abs(r["value"] - 65536 * 8 / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["streams"] == 2                                  # a 20-step call of this shard takes the two-chain branch (policy: >= 16 steps)
    assert abs(rf["achieved"] - 212 * 65536 * 8 / (rf["us_per_step"] * 1e-6) / 1e9) < 1e-6 * rf["achieved"]     # 212 B per drone-step
    assert rf["traffic"] is not None and abs(rf["traffic"] / rf["bytes_per_launch"] - 1) < 0.01                      # PMC bytes = algorithmic bytes
    assert rf["frac"] > 0.6 and r["value"] > 2.2e10            # committed: 0.80-0.82, 2.96-3.02e10
    assert r["state_sane"] is True and r["ranks_seen"] == 1


def test_strong_scaling_shard_line():
    """`--scaling strong --shard-of 8`: rank 0's eighth of config 3's env set alone on this GPU -- the line says strong, names the slice,
    and the library picked the whole-rollout launch form for this launch-bound shard size by itself (mds_set_rollout_form 0)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--scaling", "strong", "--shard-of", "8", "--steps", "200", "--warmup", "20",
                          "--no-extras", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert r["scaling"] == "strong" and r["n_gpus"] == 1 and r["config"]["shard_of"]["G"] == 8
    assert r["config"]["envs_per_gpu"] == 8192 and r["config"]["env_slice_rank0"] == [0, 8192] and "STRONG scaling" in r["config"]["workload"]
    assert r["config"]["launch_form"] == 2 and "k_rollout_geometric" in r["roofline"]["kernel"]
    assert abs(r["value"] - 8192 * 8 / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
    assert r["state_sane"] is True and r["value"] > 1.5e10     # committed: 3.3e10 (1.97 us per step); one launch per step would be 1.4e10
