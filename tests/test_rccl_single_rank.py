"""The `nccl` (= RCCL) branches of the multi-GPU code on the one GPU a test box has: a world of ONE rank makes every RCCL
call an N-GPU run makes (process-group setup with a device id, all_gather_into_tensor on uint8 / int64 device tensors,
all_reduce MAX / SUM, barrier, the gradient reducer's bucketed all_reduce) with the collectives degenerating to copies.
N > 1 is covered on CPU over gloo (tests/test_parallel_gloo.py, test_ddp_gloo.py); scaling itself cannot be measured here.
Each check runs in its own process (a process group is process-global); a hang of the communicator setup on a box is
reported as a skip, not a failure."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env_extra, timeout=240):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", **env_extra)
    try:
        return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL communicator setup did not finish on this box")


def test_parallel_helpers_over_rccl_world_of_one():
    r = _run([sys.executable, "tools/rccl_single_rank_check.py"], {"MASTER_PORT": "29533"})
    assert r.returncode == 0 and "rccl single-rank check ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("partition", ["frames", "blocks"])
def test_bench_multi_gpu_branches_over_rccl_world_of_one(partition):
    r = _run([sys.executable, "bench.py", "--workload", "config1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
              "--no-x3-record", "--no-streamed-record", "--partition", partition, "--block", "16"],
             {"PCC_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29541" if partition == "frames" else "29542"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["steps"] == 2
    if partition == "frames":
        assert line["blocks"]["decoded_points"] == line["blocks"]["points_per_frame"]


def test_training_step_reducer_over_rccl_world_of_one():
    """tools/train_bench.py with the process group up: GradBucketReducer's post-accumulate hooks launch their bucketed
    all-reduces on RCCL during backward, red.finish() waits for them, Adam steps; the loss stays finite"""
    r = _run([sys.executable, "tools/train_bench.py", "--batch", "2", "--block", "128", "--steps", "2", "--warmup", "1"],
             {"PCC_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29551"}, timeout=400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["last_loss"] == line["last_loss"]      # not NaN
