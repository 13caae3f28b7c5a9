"""`python bench.py --gpus N` must start N ranks by itself when no launcher did (missing #1): the launch plan (command
line, environment) on CPU; the real thing -- two ranks sharing the box's one GPU, gloo carrying the gather -- under `-m gpu`.
Reference: a run starts its own workers, src/read_alignment_scanner.rs:606-660 (rayon pool), src/worker_thread_data.rs:21-30."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("plo_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_launch_plan_for_eight_gpus():
    b = _bench()
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "3"]
    stale = {"WORLD_SIZE": "4", "RANK": "3", "LOCAL_RANK": "3", "MASTER_PORT": "1", "MASTER_ADDR": "elsewhere", "PATH": "/usr/bin"}
    cmd, env = b.launch_plan(8, argv, port=29555, base_env=stale)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == argv  # the same arguments reach every rank
    # the child's environment: nothing of an outer launcher leaks in, dmabuf IPC stays on, the ranks know who started them
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        assert k not in env
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PLO_BENCH_LAUNCHED"] == "1" and env["PATH"] == "/usr/bin"
    # a free port is picked when none is given
    cmd2, _ = b.launch_plan(2, [], base_env={})
    assert int(cmd2[cmd2.index("--master-port") + 1]) > 0


def test_world_size_disagreeing_with_gpus_is_an_error():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "must agree" in p.stderr and p.stdout.strip() == ""


@pytest.mark.gpu
def test_bench_starts_two_ranks_by_itself():
    """no WORLD_SIZE in the environment: bench.py --gpus 2 launches two ranks (sharing the one GPU; gloo), shards, lifts, gathers,
    and rank 0 verifies the gathered record set against its own whole-set HIP result"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PLO_BENCH_SHARE_GPU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--workload", "chr20", "--reads", "30000",
                        "--steps", "3", "--warmup", "1", "--e2e-reads", "4000", "--e2e-window", "700"], env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout  # ONE JSON line
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["value"] > 0
    assert r["verify"]["gathered_equals_single_gpu_result"] is True
    assert r["config"]["dist_backend"] == "gloo" and r["config"]["launched_by"].startswith("bench.py")
    assert len(r["shard"]["windows_per_rank"]) == 2 and all(n > 0 for n in r["shard"]["windows_per_rank"])
    # ... and BASELINE configs[3] as a BAM run: one input BAM, every rank lifts its part (plo_bam_open_range) into its own shard; the shards'
    # union holds the expected records
    sh = r["end_to_end_sharded"]
    assert sh["ranks"] == 2 and sh["reads"] == 4000 and all(n > 0 for n in sh["reads_per_rank"]) and sh["value"] > 0
    assert sh["records_verified"] >= 300 and sh["verification"]["ok"] is True


@pytest.mark.gpu
def test_bench_two_ranks_lift_wgs30x_as_a_window_pipeline():
    """BASELINE configs[3] at its own size on the one GPU of the box: bench.py --gpus 2 --workload wgs30x -- 2 M reads,
    the reference's 20 Mb windows dealt to two ranks, each rank's windows also lifted as four batches per step on two contexts with the
    record gather of batch i behind the compute of batch i+1; rank 0 checks both gathered record sets against its single-GPU result"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PLO_BENCH_SHARE_GPU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--workload", "wgs30x", "--steps", "2", "--warmup", "1",
                        "--e2e-reads", "0", "--pipeline-batches", "4", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=2400)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert r["n_gpus"] == 2 and r["config"]["workload"] == "wgs30x" and r["config"]["reads_total"] == 2_000_000 and r["value"] > 0
    v = r["verify"]
    assert v["gathered_equals_single_gpu_result"] is True and v["window_pipeline_equals_single_gpu_result"] is True and v["items"] > 2_000_000
    wp = r["gather_modes"]["window_pipeline"]
    assert wp["batches_per_rank_and_step"] == 4 and len(wp["reads_per_batch_this_rank"]) == 4 and wp["value"] > 0
    assert v["this_rank"]["single_gpu_pro_rata_ms"] > 0 and v["this_rank"]["ms_per_step_no_gather"] > 0
