"""N>1 path of bench.py on CPU: 2 ranks over gloo exercise the rendezvous, the barriers and the
max-over-ranks timing reduction (the data path itself has no collective: env shards are
independent)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_gloo_dry_run():
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--gather-obs"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # only rank 0 prints
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2
    assert rec["elapsed"] >= 0.02               # the slower rank (sleeps 20 ms) defines the time
    assert rec["gathered"] == [6, [0.0, 1.0]]   # optional swarm all-gather: rank-major env order, every rank's block present


def test_self_launch_without_torchrun():
    """`python bench.py --gpus 2` as the driver calls it (no torchrun, no rendezvous in the environment): the parent starts the two
    ranks itself, before it imports torch, relays rank 0's line and returns the children's exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and len(rec["elapsed_per_rank"]) == 2
    assert rec["elapsed_per_rank"][1] > rec["elapsed_per_rank"][0] and rec["elapsed"] >= max(rec["elapsed_per_rank"]) - 1e-9


def test_self_launch_propagates_a_failing_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--dry-run-fail-rank", "1"], cwd=ROOT,
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode != 0


def test_parent_of_self_launch_never_imports_torch():
    """The parent must not touch the GPU (a GPU-initialised process may not start ranks that re-initialise it): it returns from
    main() before `import torch`."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main("):]
    assert main.index("return self_launch(") < main.index("import torch")
    head = src[:src.index("def main(")]
    assert "import torch\n" not in "\n".join(l for l in head.splitlines() if not l.startswith(" "))      # no module-level torch import


def test_each_rank_gets_different_envs_and_same_shape():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.make_inputs(16, 8, "c3", 1000)
    b = bench.make_inputs(16, 8, "c3", 1001)
    assert a[0].shape == b[0].shape == (16, 8, 3) and a[2].shape == (16, 8, 7)
    assert not np.allclose(a[0], b[0])
    # per-env centre offsets in U(-5,5)^2, phase rule of CBFTestOrd3.py:450
    assert np.abs(a[2][..., 2:4]).max() <= 5.0
    np.testing.assert_allclose(a[2][0, :, 6], 2 * np.pi * np.arange(8) / 8.25)
    c = bench.make_inputs(4, 4, "c2", 0)
    np.testing.assert_allclose(c[2][0, :, 6], -(np.pi / 4) * (np.arange(4) - 1))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT
    env.update(extra)
    return env


def test_world_size_mismatch_exits_with_a_message():
    """`--gpus 3` under a 2-rank launch: every rank exits non-zero with a message before any barrier (nobody waits for a third rank)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29534", os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-run-cpu"]
    out = subprocess.run(cmd, cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=240)
    assert out.returncode != 0
    assert "--gpus 3 but WORLD_SIZE=2" in out.stderr + out.stdout


def test_missing_local_rank_exits_with_a_message():
    """A rendezvous environment without LOCAL_RANK is refused (the rank would silently share GPU 0 with rank 0), not defaulted."""
    env = _clean_env(RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29535")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert "LOCAL_RANK not set" in out.stderr + out.stdout


def test_a_rank_that_never_arrives_ends_in_an_error_not_a_hang():
    """Rank 0 started by hand, rank 1 never starts: the rendezvous gives up after MDS_BENCH_DIST_TIMEOUT seconds with a non-zero exit
    (the backend's default would keep the process at the rendezvous for 10-30 minutes)."""
    import time
    env = _clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29536", MDS_BENCH_DIST_TIMEOUT="8")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=180)
    assert out.returncode != 0
    assert time.time() - t0 < 120


def test_rank_zero_builds_while_the_others_wait(tmp_path):
    """The library-is-missing path: the decision to build is collective (MAX of "missing" over the ranks), rank 0 builds, everybody meets
    at the barrier -- also when rank 1 only looks after rank 0 has finished (a local decision would leave rank 0 alone at the
    barrier), and when the file is already there."""
    lib = tmp_path / "libfake.so"
    for pre_existing in (False, True):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--dry-run-build-file", str(lib)]
        out = subprocess.run(cmd, cwd=ROOT, env=_clean_env(MDS_BENCH_DIST_TIMEOUT="60"), capture_output=True, text=True, timeout=240)
        assert out.returncode == 0, out.stderr[-2000:]
        rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        assert rec["built"] is (not pre_existing) and lib.exists()
        assert rec["ranks_seen"] == 2 and len(rec["value_per_rank"]) == 2 and len(rec["elapsed_per_rank"]) == 2
        assert rec["node_wall"] >= max(rec["elapsed_per_rank"]) - 1e-3      # first start .. last finish covers every rank's interval


def test_strong_scaling_partition_is_the_one_gpu_env_set():
    """--scaling strong (GPU g owns envs [g E / G, (g + 1) E / G) of config 3's env set): two gloo ranks build their shards
    through the same function as the GPU path; their slices tile the env axis and their union is, bit for bit, the 1-GPU run's env set
    (same seed).  Weak scaling (the default the driver times): the ranks own distinct envs."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29538", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--scaling", "strong", "--dry-run-envs", "48"]
    out = subprocess.run(cmd, cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert rec["shard"] == {"scaling": "strong", "envs_total": 48, "slices": [[0, 24], [24, 48]], "slices_tile_the_env_axis": True,
                            "union_equals_one_gpu_set": True}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu"], cwd=ROOT, env=_clean_env(),
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert rec["shard"]["scaling"] == "weak" and rec["shard"]["ranks_own_distinct_envs"] and rec["shard"]["slices"] == [[0, 64], [0, 64]]


def test_shard_inputs_slices_and_refuses_a_ragged_partition():
    sys.path.insert(0, ROOT)
    import bench
    import pytest
    full = bench.make_inputs(64, 8, "c3", 1000)
    for G in (1, 2, 4, 8):
        parts = [bench.shard_inputs(64, 8, "c3", g, G, "strong") for g in range(G)]
        assert [p[3] for p in parts] == [(g * 64 // G, (g + 1) * 64 // G) for g in range(G)]
        for j in range(3):
            np.testing.assert_array_equal(np.concatenate([p[j] for p in parts]), full[j])
    with pytest.raises(SystemExit):
        bench.shard_inputs(64, 8, "c3", 0, 3, "strong")
    w = bench.shard_inputs(16, 8, "c3", 1, 2, "weak")
    np.testing.assert_array_equal(w[2], bench.make_inputs(16, 8, "c3", 1001)[2])
