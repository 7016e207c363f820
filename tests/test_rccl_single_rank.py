"""The `nccl` (= RCCL) branches of the multi-GPU code on the one GPU a test box has: a world of ONE rank makes every RCCL
call an N-GPU run makes (process-group setup with a device id, all_gather_into_tensor on uint8 / int64 device tensors,
all_reduce MAX / SUM, barrier, the gradient reducer's bucketed all_reduce) with the collectives degenerating to copies.
N > 1 is covered on CPU over gloo (tests/test_parallel_gloo.py, test_ddp_gloo.py); scaling itself cannot be measured here.
Each check runs in its own process (a process group is process-global) on a rendezvous port that was free when the test
started; a child that does not finish in time FAILS the test with what it printed (a communicator hang must not pass
silently).  The two-rank control flow of `bench.py --gpus 2` — bench.py starting its own ranks — is rehearsed on the one
GPU with PCC_BENCH_REHEARSE=1 (both ranks on cuda:0, gloo collectives)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _run(cmd, env_extra, timeout=240):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(p.pid, signal.SIGKILL)                      # the exact process group this test started
        out, err = p.communicate()
        pytest.fail(f"{' '.join(cmd)} did not finish within {timeout} s (RCCL communicator setup hang?)\n"
                    f"--- stdout ---\n{out[-3000:]}\n--- stderr ---\n{err[-3000:]}")
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)


def test_parallel_helpers_over_rccl_world_of_one():
    r = _run([sys.executable, "tools/rccl_single_rank_check.py"], {})
    assert r.returncode == 0 and "rccl single-rank check ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("partition", ["frames", "blocks"])
def test_bench_multi_gpu_branches_over_rccl_world_of_one(partition):
    r = _run([sys.executable, "bench.py", "--workload", "config1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
              "--no-x3-record", "--no-streamed-record", "--no-small-frame-record", "--partition", partition, "--block", "16"],
             {"PCC_BENCH_FORCE_DIST": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["steps"] == 2
    assert line["rccl"]["world"] == 1 and line["rccl"]["backend"].startswith("nccl") and len(line["rccl"]["devices"]) == 1
    if partition == "frames":
        assert line["blocks"]["decoded_points"] == line["blocks"]["points_per_frame"]


@pytest.mark.parametrize("reducer", ["all_reduce", "reduce_scatter"])
def test_training_step_reducer_over_rccl_world_of_one(reducer):
    """tools/train_bench.py with the process group up: GradBucketReducer's post-accumulate hooks launch their bucketed
    collectives (all-reduce, or reduce-scatter followed by the all-gather of finish()) on RCCL during backward,
    red.finish() waits for them, Adam steps; the loss stays finite"""
    r = _run([sys.executable, "tools/train_bench.py", "--batch", "2", "--block", "128", "--steps", "2", "--warmup", "1",
              "--reducer", reducer], {"PCC_BENCH_FORCE_DIST": "1"}, timeout=400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["last_loss"] == line["last_loss"]      # not NaN
    assert line["reducer"] == reducer


def test_bench_starts_its_own_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts two ranks itself (a torchrun child of a parent
    that has not touched the GPU) and relays rank 0's line.  One GPU here, so the ranks share cuda:0 and the collectives
    run over gloo (PCC_BENCH_REHEARSE=1): the control flow of an N-GPU run, not a measurement."""
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--workload", "config1", "--steps", "2", "--warmup", "1",
              "--no-cpu-baseline", "--no-x3-record", "--no-streamed-record", "--no-small-frame-record", "--block", "16"],
             {"PCC_BENCH_REHEARSE": "1"}, timeout=400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["steps"] == 2
    assert line["rccl"]["world"] == 2 and sorted(d["rank"] for d in line["rccl"]["devices"]) == [0, 1]
    assert len({d["pid"] for d in line["rccl"]["devices"]}) == 2                 # two processes
    assert line["config"]["parallelism"] == "frames x2"
    assert line["blocks"]["decoded_points"] == line["blocks"]["points_per_frame"]
    assert sum(line["blocks"]["cubes_per_rank"]) == line["blocks"]["cubes"] and min(line["blocks"]["cubes_per_rank"]) > 0
    # an N > 1 line is as complete as the N = 1 line: the roofline of the dominant class measured in this run,
    # the CPU baseline carried from the newest committed N = 1 record (labelled as replayed), every rank's own step time
    assert line["roofline"] is not None and line["roofline"]["bound"] == "mfma" and line["roofline"]["achieved"] > 0
    cb = line["cpu_baseline"]
    assert cb is not None and cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and "replayed_from" in cb
    assert cb["replayed_from"]["file"].startswith("profiles/r") and "cpu_model" in cb
    assert all(d["ms_per_step"] > 0 for d in line["rccl"]["devices"])
    lo, hi = line["rccl"]["ms_per_step_min_max_over_ranks"]
    assert 0 < lo <= hi


def test_bench_refuses_more_ranks_than_gpus():
    """--gpus 2 on a one-GPU box without the rehearsal switch: a non-zero exit and a message, never a one-rank run that
    prints n_gpus 1"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--workload", "config1", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode != 0 and "needs 2 visible GPUs" in r.stderr and not r.stdout.strip(), r.stdout + r.stderr


def test_train_bench_starts_its_own_two_ranks():
    r = _run([sys.executable, "tools/train_bench.py", "--gpus", "2", "--batch", "2", "--block", "128", "--steps", "2", "--warmup", "1"],
             {"PCC_BENCH_REHEARSE": "1"}, timeout=500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl"]["world"] == 2 and line["last_loss"] == line["last_loss"]
