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
    # the library's own launch form for this shard and call length: the whole-rollout kernel, the 20 timed steps in ONE launch (state in
    # registers): the observation row per drone-step + state / parameters once per launch
    assert r["config"]["launch_form"] == 2 and rf["streams"] == 1 and rf["steps_per_launch"] == 20 and "k_rollout_geometric" in rf["kernel"]
    bpd = 80 + 132 / 20
    assert abs(rf["achieved"] - bpd * 65536 * 8 / (rf["us_per_step"] * 1e-6) / 1e9) < 1e-6 * rf["achieved"]
    assert abs(rf["bytes_per_launch"] - bpd * 65536 * 8 * 20) < 1 and abs(rf["us_per_launch"] - 20 * rf["us_per_step"]) < 1e-9
    # PMC bytes of the launch (committed): the observation rows are rewritten in place step after step with default-policy stores, the L2
    # absorbs most rewrites -- well under the algorithmic bytes, and never above them
    assert rf["traffic"] is not None and 0.15 < rf["traffic"] / rf["bytes_per_launch"] < 1.02 and 0.3 < rf["valu"]["frac_of_issue_rate"] < 1.0
    assert rf["frac"] > 0.4 and r["value"] > 3.5e10            # committed: 0.57-0.61, 4.8-5.7e10
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
