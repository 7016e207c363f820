"""bench.py's bookkeeping that needs no GPU: the per-round profile lookup and the staleness stamp of the replayed
HBM-traffic figure (a profile taken on other kernel sources must be visible as stale)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_latest_profile_picks_the_newest_round_and_names_its_file():
    import bench
    tj = bench.latest_profile("traffic_dominant_kernel")
    assert tj is not None and tj["_file"].startswith("profiles/r") and tj["_file"].endswith("_traffic_dominant_kernel.json")
    rounds = sorted(f[:3] for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic_dominant_kernel.json"))
    assert os.path.basename(tj["_file"]).startswith(rounds[-1])
    for key in ("kernel", "traffic_bytes_per_launch", "commit", "kernel_source_sha256", "command"):
        assert key in tj, key
    assert bench.latest_profile("no_such_profile") is None


def test_kernel_source_hash_tracks_the_convolution_sources(tmp_path):
    import bench
    h = bench.kernel_source_sha256()
    assert len(h) == 64 and h == bench.kernel_source_sha256()
    # the committed traffic profile must have been taken on the sources in the tree (re-take the PMC passes after
    # touching csrc/conv.hip, common.h or the Makefile: tools/pmc_traffic.py)
    tj = bench.latest_profile("traffic_dominant_kernel")
    if tj["kernel_source_sha256"] != h:
        import pytest
        pytest.skip("profiles/*_traffic_dominant_kernel.json was taken on other kernel sources: bench.py reports "
                    "kernel_source_unchanged_since = false until the --pmc passes are re-taken (tools/pmc_traffic.py)")


def _bench(args, env_extra):
    import subprocess
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env.update({k: v for k, v in env_extra.items()})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=300)


def test_bench_never_runs_fewer_ranks_than_asked_for():
    """`python bench.py --gpus N` either runs N ranks or exits non-zero with a message.  No GPU in this
    container, so every route must refuse: the self-launcher (0 visible GPUs), a launcher world that does not match --gpus,
    and — with the rehearsal switch — the started ranks themselves (their failure is relayed as the exit code)."""
    r = _bench(["--gpus", "2"], {})
    assert r.returncode != 0 and "needs 2 visible GPUs" in r.stderr and not r.stdout.strip()
    r = _bench(["--gpus", "4"], {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in r.stderr and not r.stdout.strip()
    import torch
    if not torch.cuda.is_available():
        r = _bench(["--gpus", "2", "--workload", "config1", "--steps", "1", "--warmup", "0"], {"PCC_BENCH_REHEARSE": "1"})
        assert r.returncode != 0 and "[launch]" in r.stderr and "nproc-per-node=2" in r.stderr and not r.stdout.strip()
        assert "needs MI355X GPUs" in r.stderr                       # printed by the two ranks the launcher started


def test_kernel_source_hash_covers_the_coordinate_side_too():
    """the replayed HBM-traffic figure depends on the execution order (coords.hip / sort.hip / select.hip) as well as on
    the convolution kernel: all of them are in the staleness stamp"""
    import bench
    assert {"conv.hip", "common.h", "coords.hip", "sort.hip", "select.hip", "sort.h", "sort_small.h"} <= set(bench.KERNEL_SOURCES)


def test_visible_gpu_count_reads_the_environment_not_the_runtime(monkeypatch):
    """the launcher of `--gpus N` counts devices without touching the HIP runtime: the visible-devices variables
    decide when set; this container has no GPU nodes in /sys/class/kfd"""
    import bench
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    base = bench.visible_gpu_count()
    assert base >= 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == 0
    name = bench.cpu_model_name()
    assert name is None or (isinstance(name, str) and name)


def test_n_gt_1_line_carries_the_committed_cpu_baseline():
    """bench.py at N > 1 replays cpu_baseline from the newest committed N = 1 record: that record must exist and hold one"""
    import bench
    prev = bench.latest_profile("bench_config2")
    assert prev is not None and prev.get("cpu_baseline") and prev["cpu_baseline"]["value"] > 0 and prev["cpu_baseline"]["cores"] >= 1


def test_default_line_carries_the_committed_blas_order_baseline():
    """the default cpu_baseline is the kernel-order oracle (the faster CPU implementation); the BLAS-order oracle SURVEY §8d names rides
    along replayed from its committed record, which must exist, be slower, and hold its tolerance-class parity"""
    import bench
    blas = bench.latest_profile("cpu_baseline_blas_order")
    assert blas is not None and blas["cpu_baseline"]["parity"]["oracle_summation_order"] == "blas"
    d = blas["cpu_baseline"]["parity"]["abs_diff"]
    assert d["bpp"] <= 1e-3 and d["d1_psnr_db"] <= 1e-3 and d["y_psnr_db"] <= 1e-3
    main = bench.latest_profile("bench_config2")["cpu_baseline"]
    assert main["parity"]["oracle_summation_order"] == "kernel" and main["parity"]["streams_byte_equal"] is True
    assert main["parity"]["decoded_voxels_differing"] == 0 and main["parity"]["colours_differing"] == 0
    assert main["value"] > 2 * blas["cpu_baseline"]["value"]


def test_live_traffic_measurement_declines_inside_a_profiled_process(monkeypatch):
    """bench.py measures roofline.traffic itself (two rocprofv3 --pmc child passes of its own command) — but never from a process
    that is being profiled already (a nested profiler would be the exec-after-GPU-init the pool forbids), and its child command
    carries --no-live-pmc so that it cannot recurse"""
    import bench
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert bench.live_traffic("conv_mfma_buf_kernel<64, 128, 2, 2, 4, true") == "this process is itself being profiled"
    assert "--no-live-pmc" in bench.LIVE_PMC_FLAGS and "--no-cpu-baseline" in bench.LIVE_PMC_FLAGS
