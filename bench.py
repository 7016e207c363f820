#!/usr/bin/env python3
"""Benchmark of the hot path: ColorModel.compress + decompress of one 10-bit frame per step.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1: one rank per GPU over RCCL.  Either the driver starts the ranks (python -m torch.distributed.run ...
  bench.py --gpus N: RANK / WORLD_SIZE are in the environment) or bench.py does it itself: with --gpus N > 1 and no
  RANK it starts `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD process,
  before anything in this process has touched the GPU, relays rank 0's JSON line and exits with the child's code.  A
  world that does not match --gpus, or fewer visible GPUs than ranks, is an error (non-zero exit), never a silent
  one-rank run.  Frames are independent units (the codec is intra-only), so rank r codes its own frame each step (weak
  scaling, no data-path collective inside the codec) and the per-frame bitstreams are all-gathered over RCCL at the
  end of every step, as a whole-sequence encoder would collect them.

Metric (BASELINE.json): encode+decode Mpoints/s = points coded / (t_enc + t_dec), inputs resident
in HBM, in-memory API (strings returned; train.py:251-257), timing bracket as utils.py:448-464.
Workload: SURVEY.md §8d config 2 — synthetic 10-bit sphere shell, N = 850,824 (longdress_vox10_1300
has 857,966 points), q = (0.5, 0.5), seeded random weights of configs/Ours.yaml (no trained
weights are published).  All arithmetic fp32.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the pool's host driver shares device memory between processes (RCCL) through dmabuf only
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")        # see pcc_amd/__init__.py: two frames in flight = four busy streams beside the default one
import numpy as np
import torch

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
MFMA_BF16_PEAK_TFLOPS = 2516.6    # v_mfma_f32_32x32x16_bf16, dense (--bf16 runs only)
HBM_PEAK_GBS = 8000.0


def latest_profile(stem):
    """newest committed profiles/rNN_<stem>.json (named per round); its path is added under the key _file"""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{stem}.json")))
    if not hits:
        return None
    with open(hits[-1]) as f:
        j = json.load(f)
    j["_file"] = os.path.relpath(hits[-1], ROOT)
    return j


# what the dominant kernel's time and traffic depend on: the convolution kernel itself and the code that fixes its
# execution order and kernel maps (a change there changes the gather traffic: VERDICT r2)
KERNEL_SOURCES = ("conv.hip", "common.h", "coords.hip", "sort.hip", "sort.h", "sort_small.h", "select.hip")


def kernel_source_sha256():
    """hash of the sources the convolution launches are built from: a profile taken on other sources is stale"""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "learned-compression-of-point-cloud-geometry-and-attributes_amd", "csrc")
    for name in KERNEL_SOURCES:
        with open(os.path.join(base, name), "rb") as f:
            h.update(f.read())
    with open(os.path.join(base, "Makefile")) as f:          # the compiler flags, not the list of sources
        h.update("".join(line for line in f if line.startswith("HIPFLAGS")).encode())
    return h.hexdigest()


LIVE_PMC_FLAGS = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-blocks-record", "--no-x3-record", "--no-streamed-record",
                  "--no-small-frame-record", "--no-mid-frame-record", "--no-train-record", "--no-hbm-record", "--no-live-pmc"]


def live_traffic(kernel_substr, timeout_s=150):
    """HBM bytes per launch of the dominant kernel MEASURED IN THIS RUN: two child processes — `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE`, separate passes with the kernel trace only beside them, the program itself after `--`, as
    MI355X_MICROARCH.md prescribes — of this same script on the same workload (3 steps), reduced like tools/pmc_traffic.py
    (FETCH_SIZE x 2 on gfx950, KB -> bytes, averaged over the kernel's launches).  -> dict, or a string saying why not."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return "rocprofv3 not on PATH"
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCPROFILER_SDK_TOOL_LIBRARIES") or os.environ.get("ROCP_TOOL_LIBRARIES"):
        return "this process is itself being profiled"
    out = {}
    root = tempfile.mkdtemp(prefix="pcc_live_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(root, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                   "python3", os.path.abspath(__file__)] + LIVE_PMC_FLAGS
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return f"the {counter} pass did not finish in {timeout_s} s"
            if r.returncode != 0:
                return f"the {counter} pass failed (rc {r.returncode}): " + r.stderr.decode(errors="replace")[-300:]
            files = glob.glob(os.path.join(d, "**", "p_counter_collection.csv"), recursive=True)
            if not files:
                return f"the {counter} pass wrote no counter file"
            per, allk = {}, {}
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    if row["Counter_Name"] != counter:
                        continue
                    v = float(row["Counter_Value"])
                    a = allk.setdefault(row["Kernel_Name"], [set(), 0.0])          # every kernel: launches, KB over the whole pass
                    a[0].add(row["Dispatch_Id"])
                    a[1] += v
                    if kernel_substr in row["Kernel_Name"]:
                        per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + v
            out[counter + "_all"] = {k: (len(v[0]), v[1]) for k, v in allk.items()}
            if not per:
                return f"the {counter} pass saw no launch of {kernel_substr}"
            out[counter] = (len(per), sum(per.values()) / len(per) * 1024.0)
        # third pass: matrix-pipe busy cycles and the delivered clock of the same launches (tools/pmc_mfma.py's quotient:
        # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / launch duration).
        # Optional: its failure leaves the traffic figures standing.
        try:
            d = os.path.join(root, "MFMA")
            cmd = ["rocprofv3", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "--kernel-trace", "--output-format", "csv", "-d", d,
                   "-o", "p", "--", "python3", os.path.abspath(__file__)] + LIVE_PMC_FLAGS
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
            cc = glob.glob(os.path.join(d, "**", "p_counter_collection.csv"), recursive=True)
            kt = glob.glob(os.path.join(d, "**", "p_kernel_trace.csv"), recursive=True)
            if r.returncode == 0 and cc and kt:
                dur = {}
                with open(kt[0]) as f:
                    for row in csv.DictReader(f):
                        if kernel_substr in row["Kernel_Name"]:
                            dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
                busy = gui = 0.0
                with open(cc[0]) as f:
                    for row in csv.DictReader(f):
                        if row["Dispatch_Id"] in dur:
                            if row["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                                busy += float(row["Counter_Value"])
                            elif row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                                gui += float(row["Counter_Value"])
                secs = sum(dur.values())
                if dur and gui > 0 and secs > 0:
                    gui /= 8.0
                    out["MFMA"] = {"mfma_busy_frac": busy / (gui * 1024.0), "delivered_clock_ghz": gui / secs / 1e9,
                                   "mfma_busy_frac_of_2p4ghz": busy / 1024.0 / (secs * 2.4e9), "launches_mfma_pass": len(dur)}
        except Exception:
            pass
    finally:
        shutil.rmtree(root, ignore_errors=True)
    fb, wb = out["FETCH_SIZE"][1] * 2.0, out["WRITE_SIZE"][1]
    res = {"traffic_bytes_per_launch": fb + wb, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
           "launches_fetch_pass": out["FETCH_SIZE"][0], "launches_write_pass": out["WRITE_SIZE"][0],
           "command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE> --kernel-trace --output-format csv -- "
                      "python3 bench.py " + " ".join(LIVE_PMC_FLAGS)}
    res.update(out.get("MFMA", {}))
    res["per_kernel_kb"] = {"FETCH_SIZE": out["FETCH_SIZE_all"], "WRITE_SIZE": out["WRITE_SIZE_all"], "steps_in_run": 4}      # warm-up 1 + 3 steps
    return res


def free_port():
    """a TCP port nobody listens on right now (rendezvous of a self-launched run; fixed ports collide between
    concurrent runs on one host)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def visible_gpu_count():
    """GPUs this process could use, counted WITHOUT touching the HIP runtime: the visible-devices variables if set, else the
    GPU nodes of the kernel driver's topology (/sys/class/kfd: nodes with SIMDs).  torch.cuda.device_count() would do, but on
    ROCm it falls back to hipGetDeviceCount when amdsmi is missing, which initialises the runtime in the LAUNCHER process —
    harmless while the ranks are child processes, wrong to rely on."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    import glob
    n = 0
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(path) as f:
                for line in f:
                    if line.startswith("simd_count") and int(line.split()[1]) > 0:
                        n += 1
        except (OSError, ValueError):
            pass
    return n


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def launch_ranks(script, argv, n_gpus):
    """`--gpus N` with N > 1 and no RANK in the environment: start the N ranks as a child process group (torchrun, one
    rank per GPU, rendezvous on 127.0.0.1) and relay rank 0's stdout; returns the exit code for this process.  The
    launcher never touches the HIP runtime (visible_gpu_count reads the environment / sysfs), and the ranks are CHILDREN,
    never an exec of this process.  PCC_BENCH_REHEARSE=1 (every rank on cuda:0, gloo collectives — the control flow on a
    one-GPU box) skips the device-count check; otherwise fewer visible GPUs than ranks is an error (the ranks check again:
    a rank whose device does not exist exits non-zero)."""
    import subprocess
    rehearse = os.environ.get("PCC_BENCH_REHEARSE") == "1"
    have = visible_gpu_count()
    if not rehearse and have < n_gpus:
        print(f"{os.path.basename(script)}: --gpus {n_gpus} needs {n_gpus} visible GPUs, this host shows {have}; refusing to "
              f"run fewer ranks than asked for (PCC_BENCH_REHEARSE=1 rehearses the control flow with every rank on cuda:0 "
              f"over gloo — never a measurement)", file=sys.stderr, flush=True)
        return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.pop("PCC_BENCH_FORCE_DIST", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    print(f"[launch] {' '.join(cmd)}", file=sys.stderr, flush=True)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    line = next((ln for ln in reversed(lines) if ln.lstrip().startswith("{")), None)
    for ln in lines:
        if ln is not line:
            print(ln, file=sys.stderr)
    if r.returncode != 0 or line is None:
        print(f"[launch] the {n_gpus}-rank run failed (exit code {r.returncode}, "
              f"{'no result line' if line is None else 'result line present'})", file=sys.stderr, flush=True)
        return r.returncode or 3
    print(line, flush=True)
    return 0


def rank_devices(dist, dev, world, rehearse, extra=None):
    """what every rank runs on, all-gathered: the `rccl` record of the result line (proof that N ranks on N devices ran);
    `extra`: per-rank figures to carry along (its own ms_per_step: a straggler shows in one line)"""
    p = torch.cuda.get_device_properties(dev)
    mine = {"rank": int(os.environ.get("RANK", "0")), "device": str(dev), "name": p.name,
            "pci_bus_id": getattr(p, "pci_bus_id", None), "pid": os.getpid()}
    if extra:
        mine.update(extra)
    if dist is None:
        return {"world": 1, "backend": None, "devices": [mine]}
    got = [None] * world
    dist.all_gather_object(got, mine)
    return {"world": world, "backend": "gloo (rehearsal: every rank on cuda:0)" if rehearse else "nccl (RCCL)",
            "devices": got}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="config2", choices=["config1", "config2", "mid"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-order", default="kernel", choices=["kernel", "blas"],
                    help="summation order of the CPU baseline's oracle: kernel = hand-vectorised fused multiply-add chains (the faster CPU "
                         "implementation, ~45 s on the config-2 frame; the HIP path must reproduce its bytes), blas = torch-CPU sgemm (~130 s)")
    ap.add_argument("--cpu-sample", default="1024,260,0.5",
                    help="grid,radius,half_width of the CPU-baseline shell.  Default: the config-2 frame itself (N = 850,824: "
                         "~45 s on 16 host cores in kernel order, ~2.5 minutes in BLAS order), so cpu_baseline and cpu_baseline.parity are "
                         "on the headline workload; 256,100,0.5 is a bounded 125,672-point sample")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel-class table to stderr")
    ap.add_argument("--partition", default="frames", choices=["frames", "blocks"],
                    help="N > 1 sharding: one frame per rank (weak scaling, the default) or the cubes of ONE frame "
                         "spread over the ranks (strong scaling; different numbers than whole-frame coding, SURVEY.md 8e)")
    ap.add_argument("--block", type=int, default=512, help="cube edge of --partition blocks")
    ap.add_argument("--no-blocks-record", action="store_true",
                    help="skip the strong-scaling 'blocks' sub-record (one frame cut into cubes over the ranks) that "
                         "a frames run reports beside its value")
    ap.add_argument("--weights", default=None,
                    help="state_dict to load instead of the seeded initialisation (e.g. from tools/train.py); the headline "
                         "configuration is the seeded one")
    ap.add_argument("--bf16", action="store_true",
                    help="opt-in reduced-precision mode: bf16 operands on the wide convolutions (fp32 accumulation); NOT the "
                         "headline configuration — the reference computes in fp32 and so does the default run")
    ap.add_argument("--x3", action="store_true",
                    help="opt-in split-bf16 arithmetic (fp32 data, three-way bf16 split of both operands, six bf16 MFMA products with "
                         "fp32 accumulation) on the wide convolutions for the WHOLE run; NOT the headline configuration — a default "
                         "run reports this mode as the `split_bf16` sub-record beside the fp32 value")
    ap.add_argument("--no-x3-record", action="store_true", help="skip the `split_bf16` sub-record of a default run")
    ap.add_argument("--no-streamed-record", action="store_true", help="skip the `streamed` sub-record of a default run")
    ap.add_argument("--no-small-frame-record", action="store_true", help="skip the `small_frame` sub-record of a default run")
    ap.add_argument("--no-mid-frame-record", action="store_true", help="skip the `mid_frame` sub-record of a default run")
    ap.add_argument("--no-train-record", action="store_true", help="skip the `train_step` sub-record of a default run")
    ap.add_argument("--no-hbm-record", action="store_true", help="skip the `roofline_hbm` record of a default run")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc child passes, ~40 s): replay the committed figure")
    ap.add_argument("--file-mode", action="store_true",
                    help="after the timed region, also time compress(path=...) / decompress(path=...) (t_file, SURVEY.md 8d)")
    return ap.parse_args()


def cpu_baseline(model, state_dict, sample, dev, order="kernel"):
    """The oracle (CPU restatement, kind 'port') timed on a bounded sample of the same workload; the HIP codec then
    codes the SAME frame with the same weights and both results go into ``parity`` (the oracle run is the checker
    here, never the thing measured as ``value``).  ``order``: the oracle's summation order (oracle/nn.py) — "kernel" (default:
    every convolution as one fused multiply-add chain, hand-vectorised C on all host cores: the FASTER of the two CPU
    implementations, hence the baseline, and the one whose bytes the HIP path must reproduce) or "blas" (torch-CPU gather ->
    sgemm -> index_add_, three times slower; the independent restatement, compared within BASELINE's tolerances)."""
    from oracle import nn as oracle_nn
    from oracle.codec import Codec, count_bits
    from oracle.metrics import pc_metrics
    import pcc_amd
    was_order = oracle_nn.set_order(order)
    grid, radius, hw = sample.split(",")
    pts = pcc_amd.synthetic.sphere_shell(int(grid), float(radius), float(hw))
    qc, qf = pcc_amd.synthetic.uniform_qmap(pts[:, :3], 0.5, 0.5)
    # threads actually available to this process (a 1-GPU box gives a 16-core share), never more
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))
    torch.set_num_threads(threads)
    print(f"[bench] cpu_baseline: oracle on N={pts.shape[0]} points with {threads} threads ...", file=sys.stderr, flush=True)
    codec = Codec(state_dict)
    codec.update()
    t0 = time.time()
    strings, shape, k, coords = codec.compress(pts, qc, qf)
    t1 = time.time()
    print(f"[bench] cpu_baseline: encode {t1 - t0:.1f} s", file=sys.stderr, flush=True)
    o_rec = codec.decompress(coords, strings, shape, k)
    t2 = time.time()
    print(f"[bench] cpu_baseline: decode {t2 - t1:.1f} s", file=sys.stderr, flush=True)
    n = pts.shape[0]
    # the HIP path on the same frame
    x = torch.from_numpy(pts).to(dev)
    Q = pcc_amd.SparseTensor(coordinates=torch.from_numpy(qc).to(dev), features=torch.from_numpy(qf).to(dev), device=dev)
    h_strings, h_shape, h_k, h_coords = model.compress(x, Q)
    h_rec = model.decompress(coordinates=h_coords, strings=h_strings, shape=h_shape, k=h_k).cpu().numpy()
    res = int(grid) - 1
    key = lambda r_: r_[np.lexsort((r_[:, 2], r_[:, 1], r_[:, 0]))]
    h_rec, o_rec = key(h_rec), key(o_rec)          # one row order for both: equal clouds then give equal float32 sums in the metrics
    hm, om = pc_metrics(pts, h_rec, res), pc_metrics(pts, o_rec, res)
    flips = len(set(map(tuple, h_rec[:, :3].tolist())) ^ set(map(tuple, o_rec[:, :3].tolist())))
    parity = {"frame": f"{grid}^3 shell, N={n}", "structure_equal": bool(h_shape == shape and h_k == k),
              "bpp": {"hip": count_bits(h_strings) / n, "oracle": count_bits(strings) / n},
              "d1_psnr_db": {"hip": float(hm["sym_psnr_mse"]), "oracle": float(om["sym_psnr_mse"])},
              "y_psnr_db": {"hip": float(hm["sym_y_psnr"]), "oracle": float(om["sym_y_psnr"])},
              "decoded_voxels_differing": flips,
              "metric_resolution": res}
    parity["abs_diff"] = {"bpp": abs(parity["bpp"]["hip"] - parity["bpp"]["oracle"]),
                          "d1_psnr_db": abs(parity["d1_psnr_db"]["hip"] - parity["d1_psnr_db"]["oracle"]),
                          "y_psnr_db": abs(parity["y_psnr_db"]["hip"] - parity["y_psnr_db"]["oracle"])}
    parity["oracle_summation_order"] = order
    parity["streams_byte_equal"] = bool(h_strings[0][0] == strings[0][0] and h_strings[1][0] == strings[1][0])
    if flips == 0:
        parity["colours_differing"] = int((np.rint(h_rec[:, 3:6] * 255.0) != np.rint(o_rec[:, 3:6] * 255.0)).sum())
    oracle_nn.set_order(was_order)
    how = ("oracle/chain.c fused multiply-add chains on all cores + C rANS" if order == "kernel" else "torch-CPU sgemm + C rANS oracle")
    return {"value": n / (t2 - t0) / 1e6, "unit": "Mpoints/s", "cores": threads, "cpu_model": cpu_model_name(), "kind": "port",
            "sample": f"one {grid}^3 sphere-shell frame, N={n} points, q=(0.5,0.5), same weights; "
                      f"t_enc={t1 - t0:.2f}s t_dec={t2 - t1:.2f}s ({how})",
            "parity": parity}


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        # nothing in this process has touched the GPU yet: become the launcher of the N ranks
        sys.exit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: start exactly one rank per GPU "
              f"(python bench.py --gpus N launches them itself)", file=sys.stderr, flush=True)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs MI355X GPUs (torch.cuda.is_available() is False)", file=sys.stderr, flush=True)
        sys.exit(2)
    # PCC_BENCH_REHEARSE=1: rehearse the N > 1 control flow on a one-GPU box — every rank on cuda:0,
    # collectives over gloo on host tensors (never a measurement)
    rehearse = os.environ.get("PCC_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearse else dev          # device of the collectives' tensors
    dist = None
    # PCC_BENCH_FORCE_DIST=1: take the multi-GPU branches with a world of ONE rank over `nccl` — the collectives degenerate to
    # copies, but every RCCL call of an N-GPU run (setup, all_gather_into_tensor on uint8 / int64, all_reduce, barrier) is
    # made on the device; a one-GPU box cannot host two ranks on one device.  Never a measurement of scaling.
    force_dist = world == 1 and os.environ.get("PCC_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
    if world > 1 or force_dist:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import pcc_amd
    from pcc_amd import sparse as sp
    syn = pcc_amd.synthetic
    model = syn.make_model(0, dev)
    if args.weights:
        model.load_state_dict(torch.load(args.weights, map_location=dev))
    model.update()
    if args.bf16:
        sp.set_infer_bf16(True)
    if args.x3:
        sp.set_infer_x3(True)

    cfg = {"config1": syn.CONFIG1, "config2": syn.CONFIG2, "mid": dict(grid=256, radius=100.0, half_width=0.5)}[args.workload]
    cfg = dict(cfg)
    blocks_mode = args.partition == "blocks"
    if not blocks_mode:
        cfg["radius"] = cfg["radius"] - 0.25 * rank        # rank r codes its own frame
    pts = syn.sphere_shell(**cfg)
    N = pts.shape[0]
    qc, qf = syn.uniform_qmap(pts[:, :3], 0.5, 0.5)
    x = torch.from_numpy(pts).to(dev)
    q_coords = torch.from_numpy(qc).to(dev)
    q_feats = torch.from_numpy(qf).to(dev)

    from pcc_amd import parallel as par

    def gather_bitstreams(strings, shape, k):
        # frames are the sharded unit: every rank contributes its frame's container, all ranks end
        # up with the whole step's bitstreams (RCCL all-gather(v) over xGMI)
        return par.all_gather_bitstreams(par.pack_unit(strings, shape, k), cdev)

    t_enc = t_dec = 0.0
    step_ms = []                # encode + decode of every timed step (frames mode): a step that paid for allocator growth shows here
    last = {}

    def step_blocks(timed):
        """strong scaling: this rank codes and decodes its share of the frame's cubes, then all ranks exchange
        the cubes' containers"""
        nonlocal t_enc, t_dec
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, _, units = par.compress_blocks(model, x, q_feats, args.block, rank, world)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rec = par.decompress_blocks(model, units) if units else None
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if dist is not None:
            payload = b"".join(par.pack_items_unit(s_, sh, k_) for _, s_, sh, k_, _ in units)
            par.all_gather_bitstreams(payload, cdev)
        if timed:
            t_enc += t1 - t0
            t_dec += t2 - t1
        last.update(strings=[[b"".join(u[1][0][0] for u in units)], [b"".join(u[1][1][0] for u in units)]], rec=rec)

    # PCC_BENCH_MARK=1 (profiling runs only): a marker kernel nothing else launches, at the phase boundaries of every
    # step, for tools/trace_gaps.py to cut a rocprofv3 kernel trace into encode / decode windows
    mark_t = torch.ones(3, device=dev) if os.environ.get("PCC_BENCH_MARK") == "1" else None

    def mark():
        if mark_t is not None:
            torch.cuda.synchronize()
            torch.logcumsumexp(mark_t, 0)
            torch.cuda.synchronize()

    def step(timed):
        nonlocal t_enc, t_dec
        if blocks_mode:
            return step_blocks(timed)
        Q = pcc_amd.SparseTensor(coordinates=q_coords, features=q_feats, device=dev)   # fresh maps every step
        mark()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        strings, shape, k, coords = model.compress(x, Q)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        mark()
        t1b = time.perf_counter()
        rec = model.decompress(coordinates=coords, strings=strings, shape=shape, k=k)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        mark()
        if dist is not None:
            gather_bitstreams(strings, shape, k)
        if timed:
            t_enc += t1 - t0
            t_dec += t2 - t1b
            step_ms.append(round((t1 - t0 + t2 - t1b) * 1e3, 2))
        last.update(strings=strings, rec=rec, k=k)

    for _ in range(args.warmup):
        step(False)

    sp.PROFILER = []
    if args.breakdown:
        sp.PROFILER_MIN_ROWS = 0                  # the per-launch table wants every launch's time
    # The interpreter's cyclic collector: a FULL collection walks every container object alive — ~215 k after importing torch and
    # building the model — and takes 80-135 ms in this process; the per-launch records below (events, tuples) age into the old
    # generation and provoke exactly one during the timed steps (always the 9th of 20: 173-237 ms instead of 99; a plain
    # compress / decompress loop without the records ran 24 frames without one).  Everything alive now is moved out of the
    # collector's reach (gc.freeze(): what a long-running coder process should do after loading its model — README), and
    # automatic collection is off for the K timed steps, as timeit does (a collection of what the steps and this script's log
    # create still cost 10-20 ms when one fell into a step).  Nothing leaks meanwhile: a frame's coordinate maps no longer
    # reference each other in cycles (sparse._same_map), so their device memory goes back to the allocator by reference count —
    # `device_allocs_during_timed_steps` is 4 with the collector on or off (83 against 8 while the cycles existed).
    # PCC_BENCH_GC_OFF=0 leaves the collector on (A/B).
    import gc
    gc.collect()
    gc.freeze()
    if os.environ.get("PCC_BENCH_GC_OFF", "1") == "1":
        gc.disable()
    allocs_before = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    step0_len = None
    for _ in range(args.steps):
        n_before = len(sp.PROFILER)
        step(True)
        # The launch log holds every map's row masks and group masks (FLOP accounting) — of every timed step, so none of them
        # went back to the caching allocator and every step paid for new device blocks (152 hipMalloc calls in 20 timed steps).
        # The steps code the same frame and log the same launches in the same order: later steps point at the first step's tensors.
        if not blocks_mode:
            if step0_len is None:
                step0_len = len(sp.PROFILER) - n_before
            elif len(sp.PROFILER) - n_before == step0_len:
                log = sp.PROFILER
                for i_ in range(step0_len):
                    e_, f_ = log[n_before + i_], log[i_]
                    if e_[1:3] == f_[1:3] and e_[4] == f_[4]:
                        log[n_before + i_] = e_[:3] + (f_[3], e_[4], e_[5], e_[6], f_[7])
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    gc.enable()
    device_allocs_timed = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - allocs_before
    prof, sp.PROFILER = sp.PROFILER, None

    own_ms_per_step = elapsed / args.steps * 1e3
    rccl = rank_devices(dist, dev, world, rehearse, {"ms_per_step": round(own_ms_per_step, 3),
                                                       "step_ms_min_max": [min(step_ms), max(step_ms)] if step_ms else None})
    if world > 1:
        per_rank = [d_["ms_per_step"] for d_ in rccl["devices"]]
        rccl["ms_per_step_min_max_over_ranks"] = [min(per_rank), max(per_rank)]
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        ntot = torch.tensor([N], dtype=torch.int64, device=cdev)
        dist.all_reduce(ntot)
        n_total = N if blocks_mode else int(ntot.item())        # blocks: every rank holds the same frame
    else:
        n_total = N

    # ---- north_star's whole-frame mode beside the frames number: ONE frame (rank 0's) cut into cubes that are spread
    # over the ranks by point count (strong scaling; different numbers than whole-frame coding) ----
    blocks_record = None
    if not blocks_mode and not args.no_blocks_record:
        if rank == 0:
            xb, qb = x, q_feats
        else:
            cfg0 = dict(cfg)
            cfg0["radius"] = cfg["radius"] + 0.25 * rank
            p0 = syn.sphere_shell(**cfg0)
            xb = torch.from_numpy(p0).to(dev)
            qb = torch.from_numpy(syn.uniform_qmap(p0[:, :3], 0.5, 0.5)[1]).to(dev)
        nb_frame = xb.shape[0]
        b_steps = max(1, min(3, args.steps))

        def blocks_step():
            _, parts_, units = par.compress_blocks(model, xb, qb, args.block, rank, world)
            torch.cuda.synchronize()
            rec_ = par.decompress_blocks(model, units) if units else None
            payload = b"".join(par.pack_items_unit(s_, sh, k_) for _, s_, sh, k_, _ in units)
            if dist is not None:
                par.all_gather_bitstreams(payload, cdev)
            return units, parts_, (0 if rec_ is None else rec_.shape[0]), rec_

        blocks_step()                                               # warm-up: kernel maps of the cube sizes, caches
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        for _ in range(b_steps):
            b_units, b_parts, b_dec, b_rec = blocks_step()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        b_elapsed = time.perf_counter() - tb0
        b_bits = sum(pcc_amd.utils.count_bits(u[1]) for u in b_units)
        if dist is not None:
            tb = torch.tensor([b_elapsed], dtype=torch.float64, device=cdev)
            dist.all_reduce(tb, op=dist.ReduceOp.MAX)
            b_elapsed = float(tb.item())
            agg = torch.tensor([b_bits, b_dec], dtype=torch.int64, device=cdev)
            dist.all_reduce(agg)
            b_bits, b_dec = int(agg[0].item()), int(agg[1].item())
        blocks_record = {"value": nb_frame * b_steps / b_elapsed / 1e6, "unit": "Mpoints/s", "scaling": "strong",
                         "steps": b_steps, "ms_per_step": b_elapsed / b_steps * 1e3, "block": args.block,
                         "cubes": int(sum(len(p_) for p_ in b_parts)), "cubes_per_rank": [len(p_) for p_ in b_parts],
                         "bpp": b_bits / nb_frame, "decoded_points": b_dec, "points_per_frame": nb_frame,
                         "note": "one frame per step, its cubes coded as batch items of one compress call per rank "
                                 "(per-item k and top-k), containers all-gathered; a different partition gives "
                                 "different numbers than whole-frame coding (parity target: the oracle on the same items)"}

        if world == 1 and b_rec is not None:
            # the block partition changes the numbers — report them beside the whole-frame ones
            from pcc_amd.metrics import PointCloudMetric
            res_b = cfg["grid"] - 1
            mb, _ = PointCloudMetric(xb, b_rec, resolution=res_b).compute_pointcloud_metrics(drop_duplicates=True)
            mw, _ = PointCloudMetric(x, last["rec"], resolution=res_b).compute_pointcloud_metrics(drop_duplicates=True)
            w_bpp = pcc_amd.utils.count_bits(last["strings"]) / N
            blocks_record["d1_psnr_db"], blocks_record["y_psnr_db"] = float(mb["sym_psnr_mse"]), float(mb["sym_y_psnr"])
            blocks_record["delta_vs_whole_frame"] = {"bpp": blocks_record["bpp"] - w_bpp,
                                                     "d1_psnr_db": float(mb["sym_psnr_mse"]) - float(mw["sym_psnr_mse"]),
                                                     "y_psnr_db": float(mb["sym_y_psnr"]) - float(mw["sym_y_psnr"])}

    # ---- second record, never `value`: the same frame with split-bf16 arithmetic on the wide convolutions ----
    x3_record = None
    if not blocks_mode and not args.bf16 and not args.x3 and not args.no_x3_record and rank == 0 and world == 1:
        from pcc_amd.metrics import PointCloudMetric
        res = cfg["grid"] - 1

        def quality(rec_):
            m_, _ = PointCloudMetric(x, rec_, resolution=res).compute_pointcloud_metrics(drop_duplicates=True)
            return float(m_["sym_psnr_mse"]), float(m_["sym_y_psnr"])

        f32_bpp = pcc_amd.utils.count_bits(last["strings"]) / N
        f32_d1, f32_y = quality(last["rec"])
        sp.set_infer_x3(True)
        try:
            te = td = 0.0
            x_steps = max(1, min(5, args.steps))
            # four warm-up steps: the mode's own buffers change what the caching allocator has to find, and one step in the
            # first four paid for new device blocks with 55 ms (a 110 ms decode among 55 ms ones, step 3 of a trained-weights run)
            x_warm = 4
            x_step_ms = []
            for it in range(x_warm + x_steps):
                Qx = pcc_amd.SparseTensor(coordinates=q_coords, features=q_feats, device=dev)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                xs, xshape, xk, xc = model.compress(x, Qx)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                xrec = model.decompress(coordinates=xc, strings=xs, shape=xshape, k=xk)
                torch.cuda.synchronize(); t2 = time.perf_counter()
                if it >= x_warm:
                    te += t1 - t0; td += t2 - t1
                    x_step_ms.append(round((t2 - t0) * 1e3, 2))
        finally:
            sp.set_infer_x3(False)
        x_bpp = pcc_amd.utils.count_bits(xs) / N
        x_d1, x_y = quality(xrec)
        x3_record = {"value": N * x_steps / (te + td) / 1e6, "unit": "Mpoints/s", "steps": x_steps,
                     "ms_per_step": (te + td) / x_steps * 1e3, "t_enc_ms": te / x_steps * 1e3, "t_dec_ms": td / x_steps * 1e3,
                     "step_ms": x_step_ms,
                     "dtype": "f32 data and accumulation; products of the wide convolutions as 6 bf16 MFMA terms of an exact 3-way "
                              "bf16 split of both operands (dropped terms < 3 x 2^-24 |x||w| per product)",
                     "bpp": x_bpp, "d1_psnr_db": x_d1, "y_psnr_db": x_y,
                     "vs_f32_same_frame": {"bpp_abs_diff": abs(x_bpp - f32_bpp), "d1_psnr_db_abs_diff": abs(x_d1 - f32_d1),
                                           "y_psnr_db_abs_diff": abs(x_y - f32_y), "k_equal": bool(xk == last["k"])},
                     "note": "opt-in (PCC_INFER_X3=1 / bench.py --x3); the headline `value` is the fp32-multiply run above; encoder "
                             "and decoder must run in the same mode"}

    # ---- a small frame (BASELINE config 1's 4,904-point shell), never `value`: what a frame that cannot fill the chip costs,
    # with the launches shaped for it (small-launch convolution kernel, one-workgroup map / top-k / coordinate-set chains) and,
    # in the same process, with all of them switched off — the streams must be the same bytes ----
    small_frame_record = None
    if (args.workload == "config2" and not blocks_mode and not args.bf16 and not args.x3 and not args.no_small_frame_record
            and rank == 0 and world == 1 and not args.file_mode):
        try:
            s_pts = syn.sphere_shell(**syn.CONFIG1)
            s_qc, s_qf = syn.uniform_qmap(s_pts[:, :3], 0.5, 0.5)
            sx, sqc, sqf = torch.from_numpy(s_pts).to(dev), torch.from_numpy(s_qc).to(dev), torch.from_numpy(s_qf).to(dev)

            def small_steps(n_steps):
                te = td = 0.0
                out_ = None
                for _ in range(n_steps):
                    torch.cuda.synchronize()
                    a0 = time.perf_counter()
                    Qs = pcc_amd.SparseTensor(coordinates=sqc, features=sqf, device=dev)
                    ss, sshape, sk, sc = model.compress(sx, Qs)
                    torch.cuda.synchronize()
                    a1 = time.perf_counter()
                    srec = model.decompress(coordinates=sc, strings=ss, shape=sshape, k=sk)
                    torch.cuda.synchronize()
                    a2 = time.perf_counter()
                    te += a1 - a0
                    td += a2 - a1
                    out_ = (ss, srec)
                return te / n_steps * 1e3, td / n_steps * 1e3, out_

            def measure():
                small_steps(8)
                reps = sorted((small_steps(15) for _ in range(3)), key=lambda r: r[0] + r[1])
                return reps[1]                                  # the median repetition

            on_e, on_d, on_out = measure()
            was = (sp.set_conv_small_max(0), sp.set_small_map_max(0), sp.set_small_paths(0))
            try:
                off_e, off_d, off_out = measure()
            finally:
                sp.set_conv_small_max(was[0]); sp.set_small_map_max(was[1]); sp.set_small_paths(was[2])
            small_frame_record = {
                "workload": "config1: 64^3 sphere shell, N=%d points" % s_pts.shape[0], "ms_per_frame": on_e + on_d,
                "t_enc_ms": on_e, "t_dec_ms": on_d, "value": s_pts.shape[0] / (on_e + on_d) / 1e3, "unit": "Mpoints/s",
                "small_launch_paths_off": {"ms_per_frame": off_e + off_d, "t_enc_ms": off_e, "t_dec_ms": off_d},
                "same_bytes_and_points_either_way": bool(on_out[0] == off_out[0] and torch.equal(on_out[1], off_out[1])),
                "note": "median of three repetitions of 15 frames each; `small_launch_paths_off` = the same process with the "
                        "small-launch convolution kernel and the one-workgroup chains switched off (pcc_conv_small_max(0), "
                        "pcc_small_paths(0), no one-launch maps): the tile kernels and separate launches of a full-size frame"}
        except Exception as e:                                   # a sub-record must not take the headline down with it
            small_frame_record = {"error": repr(e)}

    # ---- third record, never `value`: a streamed sequence — two frames in flight on one GPU (two host threads, each on its
    # own HIP stream), so one frame's serial host range coder runs while the other's kernels do ----
    streamed_record = None
    if (not blocks_mode and not args.bf16 and not args.x3 and not args.no_streamed_record and rank == 0 and world == 1
            and not args.file_mode):
        import threading

        import queue

        # Two PERSISTENT workers (a streaming coder's shape): each owns a HIP stream and, through its thread id, its pinned
        # staging buffers, count word and prefetch side stream for the whole record.  (Workers created anew for every
        # repetition re-created all of that inside the timed span, and the record read 82 or 102-114 ms per frame from one
        # process to the next.)  Frames are handed out through a queue; the second worker's first frame of a run is held back
        # by half a sequential frame — frames of a sequence arrive one after the other, and two frames started together can
        # stay in lock-step: both in their convolutions, then both in their host range coders.
        stagger_s = 0.5 * (t_enc + t_dec) / max(args.steps, 1)
        jobs, results, errs = queue.Queue(), queue.Queue(), []

        def worker(wi):
            try:
                s_ = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(s_):
                    while True:
                        job = jobs.get()
                        if job is None:
                            break
                        if job == "hold":
                            time.sleep(stagger_s)
                            continue
                        Qs = pcc_amd.SparseTensor(coordinates=q_coords, features=q_feats, device=dev)
                        ss, sshape, sk, sc = model.compress(x, Qs)
                        srec = model.decompress(coordinates=sc, strings=ss, shape=sshape, k=sk)
                        results.put((srec.shape[0], pcc_amd.utils.count_bits(ss)))
            except BaseException as e:               # surfaced below: a worker's failure must fail the bench
                errs.append(e)
                results.put(None)

        workers_ = [threading.Thread(target=worker, args=(wi,), daemon=True) for wi in range(2)]
        [t.start() for t in workers_]

        def run_stream(_workers, n_frames):
            torch.cuda.synchronize()
            ts0 = time.perf_counter()
            jobs.put("frame")                            # the first worker starts at once,
            jobs.put("hold")                             # the second takes the hold, then frames like the first
            for _ in range(n_frames - 1):
                jobs.put("frame")
            done = []
            for _ in range(n_frames):
                r_ = results.get()
                if r_ is None or errs:
                    raise errs[0]
                done.append(r_)
            torch.cuda.synchronize()
            return time.perf_counter() - ts0, done

        _swi = sys.getswitchinterval()
        if os.environ.get("PCC_STREAM_SWITCH_INTERVAL"):
            sys.setswitchinterval(float(os.environ["PCC_STREAM_SWITCH_INTERVAL"]))
        try:
            # The two-thread record swings between runs (thread timing against the interpreter lock: the driver's round-2 run
            # and the committed profile disagreed by 25 %): three repetitions, the MEDIAN is the value, min / max beside it
            s_frames = max(8, min(32, 4 * args.steps))          # the half-frame stagger is inside the timed span: amortised over the run
            s_reps = 3
            run_stream(2, 4)                                  # per-thread warm-up (pinned staging, side streams, count words)
            runs = [run_stream(2, s_frames) for _ in range(s_reps)]
            per_frame = sorted(e_ / s_frames * 1e3 for e_, _ in runs)
            s_done = [d for _, done_ in runs for d in done_]
            f32_bits = pcc_amd.utils.count_bits(last["strings"])
            med = per_frame[len(per_frame) // 2]
            streamed_record = {"value": N / med / 1e3, "unit": "Mpoints/s", "frames": s_frames, "repetitions": s_reps,
                               "frames_in_flight": 2, "ms_per_frame": med,
                               "ms_per_frame_min_max": [per_frame[0], per_frame[-1]],
                               "streams_equal_to_sequential": bool(all(d == (last["rec"].shape[0], f32_bits) for d in s_done)),
                               "note": "whole frames, each encoded to bytes and decoded from them; median of the repetitions; a frame's "
                                       "latency is the sequential ms_per_step or more — only the throughput of a sequence gains "
                                       "(BASELINE config 4's shape on one GPU); the headline `value` is the one-frame-at-a-time run above"}
            if not args.no_x3_record:
                # the same sequence with the opt-in split-bf16 products (the `split_bf16` record's arithmetic): what the
                # fp32-class path sustains when both the range coder and a third of the matrix time are out of the way
                sp.set_infer_x3(True)
                try:
                    run_stream(2, 4)
                    xruns = sorted(run_stream(2, s_frames)[0] / s_frames * 1e3 for _ in range(s_reps))
                finally:
                    sp.set_infer_x3(False)
                xmed = xruns[len(xruns) // 2]
                streamed_record["with_split_bf16"] = {"value": N / xmed / 1e3, "unit": "Mpoints/s", "ms_per_frame": xmed,
                                                      "ms_per_frame_min_max": [xruns[0], xruns[-1]]}
        except Exception as e:                         # a sub-record must never take the headline line down with it
            import traceback
            traceback.print_exc(file=sys.stderr)
            streamed_record = {"error": repr(e)[:300]}
        finally:
            sys.setswitchinterval(_swi)
            [jobs.put(None) for _ in workers_]
            [t.join(timeout=30) for t in workers_]

    # ---- the HBM-bound (non-matrix) operators of the same frame, never `value`: coordinate sets, kernel maps, execution order,
    # top-k, pruning, row movers — each call bracketed by HIP events on the stream it is launched on (the frame's stream or the
    # map-prefetch side stream) in EXTRA steps after the timed ones (two events are ~9 us of host time per call: 3 ms per frame),
    # algorithmic bytes per call from the formulas beside each call site in pcc_amd/sparse.py ----
    roofline_hbm = None
    if args.workload == "config2" and not blocks_mode and rank == 0 and world == 1 and not args.no_hbm_record and not args.bf16 and not args.x3:
        try:
            h_steps = 3
            step(False)
            sp.COORD_PROFILER = []
            sp.PROFILER = []                  # the pair counts the byte formulas read are cached by the convolution log's entries
            for _ in range(h_steps):
                step(False)
            torch.cuda.synchronize()
            clog, sp.COORD_PROFILER, sp.PROFILER = sp.COORD_PROFILER, None, None
            ops = {}
            for name, rows, alg, e0, e1, on_main in clog:
                o = ops.setdefault(name, dict(calls=0, ms=0.0, ms_main=0.0, bytes=0.0, rows=0, max_ms=0.0, max_rows=0))
                ms = e0.elapsed_time(e1)
                o["calls"] += 1
                o["ms"] += ms
                o["ms_main"] += ms if on_main else 0.0
                o["bytes"] += float(alg() if callable(alg) else alg)
                o["rows"] += rows
                if ms > o["max_ms"]:
                    o["max_ms"], o["max_rows"] = ms, rows
            tj = latest_profile("hbm_kernel_traffic")
            classes_h = []
            for name, o in sorted(ops.items(), key=lambda kv: -kv[1]["ms"]):
                gbps = o["bytes"] / (o["ms"] * 1e-3) / 1e9 if o["ms"] > 0 else 0.0
                rec = {"operator": name, "calls_per_step": o["calls"] / h_steps, "ms_per_step": o["ms"] / h_steps,
                       "ms_per_step_on_main_stream": o["ms_main"] / h_steps,
                       "algorithmic_mb_per_step": o["bytes"] / h_steps / 1e6, "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": gbps / HBM_PEAK_GBS, "largest_call": {"rows": o["max_rows"], "ms": o["max_ms"]}}
                if tj is not None and name in tj.get("operators", {}):
                    t_ = tj["operators"][name]
                    rec["traffic"] = t_.get("hbm_bytes_per_step")
                    rec["traffic_kernels"] = t_.get("kernels")
                classes_h.append(rec)
            tot = sum(o["ms"] for o in ops.values()) / h_steps
            tot_main = sum(o["ms_main"] for o in ops.values()) / h_steps
            roofline_hbm = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "steps": h_steps,
                            "ms_per_step": tot, "ms_per_step_on_main_stream": tot_main,
                            "ms_per_step_on_prefetch_stream": tot - tot_main,
                            "algorithmic_mb_per_step": sum(o["bytes"] for o in ops.values()) / h_steps / 1e6,
                            "achieved": sum(o["bytes"] for o in ops.values()) / max(sum(o["ms"] for o in ops.values()), 1e-9) / 1e6,
                            "operators": classes_h,
                            "traffic_source": None if tj is None else {
                                "file": tj["_file"], "taken_at_commit": tj.get("commit"),
                                "kernel_source_unchanged_since": tj.get("kernel_source_sha256") == kernel_source_sha256(),
                                "how": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, per kernel name, FETCH x2 "
                                       "gfx950 correction (tools/pmc_traffic.py --all); replayed from the committed file"},
                            "note": "per OPERATOR call (a call is one to ten kernels: an execution order = offset counts, keys, four radix "
                                    "passes, group masks), HIP events on the launching stream in 3 extra steps after the timed ones; "
                                    "times of calls on the prefetch stream overlap the convolutions of the main stream"}
            roofline_hbm["frac"] = roofline_hbm["achieved"] / HBM_PEAK_GBS
        except Exception as e:                                   # a sub-record must not take the headline down with it
            import traceback
            traceback.print_exc(file=sys.stderr)
            sp.COORD_PROFILER = None
            roofline_hbm = {"error": repr(e)[:300]}

    # ---- a mid-size frame (125,672 points: one rank's share of the config-2 frame in whole-frame-over-8-GPUs mode), never
    # `value`: the frame every rank of the spatial-block mode sees ----
    mid_frame_record = None
    if (args.workload == "config2" and not blocks_mode and not args.bf16 and not args.x3 and not args.no_mid_frame_record
            and rank == 0 and world == 1 and not args.file_mode):
        try:
            m_pts = syn.sphere_shell(grid=256, radius=100.0, half_width=0.5)
            m_qc, m_qf = syn.uniform_qmap(m_pts[:, :3], 0.5, 0.5)
            mx, mqc, mqf = torch.from_numpy(m_pts).to(dev), torch.from_numpy(m_qc).to(dev), torch.from_numpy(m_qf).to(dev)

            def mid_steps(n_steps):
                te = td = 0.0
                for _ in range(n_steps):
                    torch.cuda.synchronize()
                    a0 = time.perf_counter()
                    Qm = pcc_amd.SparseTensor(coordinates=mqc, features=mqf, device=dev)
                    ms_, mshape, mk, mc = model.compress(mx, Qm)
                    torch.cuda.synchronize()
                    a1 = time.perf_counter()
                    model.decompress(coordinates=mc, strings=ms_, shape=mshape, k=mk)
                    torch.cuda.synchronize()
                    a2 = time.perf_counter()
                    te += a1 - a0
                    td += a2 - a1
                return te / n_steps * 1e3, td / n_steps * 1e3

            mid_steps(4)
            reps = sorted((mid_steps(10) for _ in range(3)), key=lambda r: r[0] + r[1])
            m_e, m_d = reps[1]
            pro_rata = elapsed / args.steps * 1e3 * m_pts.shape[0] / N
            mid_frame_record = {"workload": "256^3 sphere shell, N=%d points" % m_pts.shape[0], "ms_per_frame": m_e + m_d,
                                "ms_per_frame_min_max": [reps[0][0] + reps[0][1], reps[2][0] + reps[2][1]],
                                "t_enc_ms": m_e, "t_dec_ms": m_d, "value": m_pts.shape[0] / (m_e + m_d) / 1e3, "unit": "Mpoints/s",
                                "pro_rata_of_the_full_frame_ms": pro_rata, "over_pro_rata": (m_e + m_d) / pro_rata,
                                "note": "median of three repetitions of 10 frames; the share of the config-2 frame one of eight ranks "
                                        "codes in the spatial-block mode: its time over the pro-rata time is what strong scaling can "
                                        "reach at best before any communication"}
        except Exception as e:
            mid_frame_record = {"error": repr(e)[:300]}

    # ---- the training step (BASELINE config 5's shape on one GPU), never `value`: tools/train_bench.py as a
    # CHILD process (its own model in training mode, Adam, 8 cubes of 256^3 = ~812 k points per step), fp32 and bf16 operands ----
    train_record = None
    if (args.workload == "config2" and not blocks_mode and not args.bf16 and not args.x3 and not args.no_train_record
            and rank == 0 and world == 1 and not args.file_mode):
        import subprocess
        train_record = {}
        torch.cuda.empty_cache()
        for tag, env_add in (("f32", {}), ("bf16_operands", {"PCC_TRAIN_BF16": "1"})):
            try:
                env = dict(os.environ, **env_add)
                for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PCC_BENCH_FORCE_DIST", "PCC_BENCH_REHEARSE"):
                    env.pop(k_, None)
                r_ = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_bench.py"), "--batch", "8", "--block", "256",
                                     "--steps", "5", "--warmup", "3"], env=env, capture_output=True, text=True, timeout=150)
                line = next((ln for ln in reversed(r_.stdout.splitlines()) if ln.lstrip().startswith("{")), None)
                if r_.returncode != 0 or line is None:
                    train_record[tag] = {"error": (r_.stderr or r_.stdout)[-300:]}
                    continue
                j_ = json.loads(line)
                train_record[tag] = {"ms_per_step": j_["ms_per_step"], "value": j_["value"] / 1e6, "unit": "Mpoints/s",
                                     "points_per_step": j_["points_per_step"], "last_loss": j_["last_loss"], "dtype": j_["dtype"]}
            except Exception as e:
                train_record[tag] = {"error": repr(e)[:300]}
        train_record["note"] = ("tools/train_bench.py --batch 8 --block 256 --steps 5 --warmup 3 in a child process: forward + loss + "
                                "backward + Adam of configs/Ours.yaml on 8 cubes cut from the config-2 frame, one MI355X")

    # ---- per-kernel-class accounting from the HIP events recorded around every conv launch ----
    classes = {}
    pop_cache = {}

    def active_slots(gmask, n_out):
        """issued rows / 32: sum over MFMA row tiles of the number of kernel offsets the tile executes (a 16-row tile = 1/2)"""
        if gmask is None:
            return (n_out + 31) // 32
        key = gmask.data_ptr()
        if key not in pop_cache and gmask.dtype == torch.int16:
            # compacted-offset kernel: list lengths per (group, offset); a list executes ceil(length / 32) tiles
            pop_cache[key] = int(((gmask.cpu().numpy().astype(np.int64) + 31) // 32).sum())
        if key not in pop_cache:
            # counted on the host: 27 tiny device reductions per map would fill the kernel trace
            bits = int(np.unpackbits(gmask.cpu().numpy().view(np.uint8)).sum())
            pop_cache[key] = bits / 2.0 if gmask.numel() == (n_out + 15) // 16 and n_out > 16 else bits
        return pop_cache[key]

    launches = []
    for entry in prof:
        name, cin, cout, pairs, n_out, e0, e1, gmask = entry
        name = sp.profiled_name(entry)
        p = int(pairs.item()) if torch.is_tensor(pairs) else int(pairs)
        c = classes.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, gather_bytes=0.0, exec_flops=0.0, untimed_launches=0,
                                          untimed_flops=0.0))
        ex = 0.0 if name.startswith("narrow") else 2.0 * 32 * active_slots(gmask, n_out) * cin * ((cout + 31) // 32 * 32)
        if e0 is None:
            # launches below sp.PROFILER_MIN_ROWS rows carry no events: their FLOPs count towards the step's total only, so that a
            # class's rate, average launch time and bytes per launch all describe the same (timed) launches
            c["untimed_launches"] += 1
            c["untimed_flops"] += 2.0 * p * cin * cout
            launches.append((0.0, name, cin, cout, n_out, p, ex))
            continue
        ms = e0.elapsed_time(e1)
        launches.append((ms, name, cin, cout, n_out, p, ex))
        c["exec_flops"] += ex
        c["launches"] += 1
        c["ms"] += ms
        c["flops"] += 2.0 * p * cin * cout
        c["gather_bytes"] += 4.0 * (p * cin + n_out * cout)
    timed_classes = {n_: c for n_, c in classes.items() if c["launches"]}
    dom_name = max(timed_classes, key=lambda n: timed_classes[n]["ms"]) if timed_classes else None
    roofline = None
    if dom_name:
        d = classes[dom_name]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        # HBM traffic per launch comes from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this
        # same command, reduced by tools/pmc_traffic.py and committed under profiles/
        traffic, traffic_src, live_pmc = None, None, None
        tj = latest_profile("traffic_dominant_kernel")
        if args.workload == "config2" and tj is not None and tj["kernel"] in dom_name:
            traffic = tj["traffic_bytes_per_launch"]
            traffic_src = {"file": tj["_file"], "taken_at_commit": tj.get("commit"),
                           "command": tj.get("command"),
                           "kernel_source_unchanged_since": tj.get("kernel_source_sha256") == kernel_source_sha256(),
                           "how": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, FETCH x2 gfx950 "
                                  "correction (tools/pmc_traffic.py); replayed from the committed file, not measured in this run"}
        # ... unless it can be measured here and now (default N = 1 run): the same two passes as children of this process
        if (traffic_src is not None and rank == 0 and world == 1 and not args.no_live_pmc and not args.bf16 and not args.x3
                and not blocks_mode and not args.weights and os.environ.get("PCC_BENCH_MARK") != "1"):
            print("[bench] roofline.traffic: two rocprofv3 --pmc child passes ...", file=sys.stderr, flush=True)
            live = live_traffic(dom_name.rstrip(">"))
            live_pmc = live if isinstance(live, dict) else None
            if isinstance(live, dict):
                replayed = traffic
                traffic = live["traffic_bytes_per_launch"]
                traffic_src = {"how": "MEASURED IN THIS RUN: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (kernel trace only beside them) "
                                      "of this command as child processes, FETCH x2 gfx950 correction, averaged over the kernel's launches",
                               "command": live["command"], "fetch_bytes_per_launch": live["fetch_bytes_per_launch"],
                               "write_bytes_per_launch": live["write_bytes_per_launch"], "launches": [live["launches_fetch_pass"], live["launches_write_pass"]],
                               "committed_figure": {"file": traffic_src["file"], "traffic_bytes_per_launch": replayed,
                                                    "kernel_source_unchanged_since": traffic_src["kernel_source_unchanged_since"]}}
            else:
                traffic_src["live_measurement_skipped"] = live
        peak = MFMA_BF16_PEAK_TFLOPS if "[bf16]" in dom_name else MFMA_F32_PEAK_TFLOPS
        if "[x3]" in dom_name:
            peak = MFMA_BF16_PEAK_TFLOPS / 6.0      # six bf16 products per algorithmic multiply-add
        if "[bf16]" in dom_name or "[x3]" in dom_name:
            traffic = traffic_src = None               # the PMC passes were taken on the fp32 kernel
        roofline = {"bound": "mfma", "kernel": dom_name, "achieved": achieved, "peak": peak,
                    "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                    "traffic_unit": "bytes per launch (HBM, PMC)", "traffic_source": traffic_src,
                    "algorithmic_gather_bytes_per_launch": d["gather_bytes"] / d["launches"],
                    "launches_per_step": d["launches"] / args.steps,
                    "avg_launch_ms": d["ms"] / d["launches"],
                    "algorithmic_gflop_per_launch": d["flops"] / d["launches"] / 1e9,
                    "share_of_step_time": d["ms"] * 1e-3 / elapsed,
                    # the same launches seen from the memory side: measured HBM bytes (PMC) over the measured duration
                    "hbm_gbps": None if traffic is None else traffic / (d["ms"] / d["launches"] * 1e-3) / 1e9,
                    "hbm_peak_gbps": 8000.0}
        # the peak assumes 2.4 GHz; the delivered clock of this kernel class under load was measured in a separate
        # rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace pass of this command (profiles/r01_clock_under_load.json)
        cj = latest_profile("clock_under_load")
        if args.workload == "config2" and "[bf16]" not in dom_name and "[x3]" not in dom_name and cj is not None:
            for key, val in cj.get("bench_config2_frame", {}).items():
                if key.replace(" ", "").startswith(dom_name.replace(" ", "").rstrip(">")) and "mean_clock_GHz" in val:
                    roofline["peak_assumes_clock_ghz"] = 2.4
                    roofline["delivered_clock_ghz"] = val["mean_clock_GHz"]
                    roofline["frac_at_delivered_clock"] = achieved / (peak * val["mean_clock_GHz"] / 2.4)
                    roofline["clock_source"] = {"file": cj["_file"], "taken_at_commit": cj.get("commit"),
                                                "how": "GRBM_GUI_ACTIVE / launch duration, separate --pmc pass"}
        # the matrix pipe seen by the hardware counters: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs) of
        # this kernel template, from a separate rocprofv3 --pmc pass of this command (tools/gpu_pmc_job.sh -> tools/pmc_mfma.py)
        mj = latest_profile("mfma_util")
        if args.workload == "config2" and "[bf16]" not in dom_name and "[x3]" not in dom_name and mj is not None:
            want = dom_name.replace(" ", "").rstrip(">")
            for key, val in mj.get("kernels", {}).items():
                if key.startswith(want):
                    roofline["mfma_busy_frac"] = val["mfma_busy_frac"]
                    roofline["mfma_busy_frac_of_2p4ghz"] = val["mfma_busy_frac_of_2p4ghz"]
                    roofline["mfma_counter_source"] = {
                        "file": mj["_file"], "taken_at_commit": mj.get("commit"),
                        "kernel_source_unchanged_since": mj.get("kernel_source_sha256") == kernel_source_sha256(),
                        "how": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), one rocprofv3 --pmc pass with "
                               "--kernel-trace; busy share of SIMD-cycles at the delivered clock, and the same cycles against 2.4 GHz; "
                               "replayed from the committed file, not measured in this run"}
                    break
        if roofline is not None and live_pmc is not None and "mfma_busy_frac" in live_pmc:
            how = ("MEASURED IN THIS RUN: one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE child pass of this command (kernel trace "
                   "only beside it), over the dominant kernel's launches")
            for key, src in (("mfma_busy_frac", "mfma_counter_source"), ("delivered_clock_ghz", "clock_source")):
                committed = roofline.get(key)
                roofline[key] = live_pmc[key]
                roofline[src] = {"how": how, "launches": live_pmc["launches_mfma_pass"],
                                 "committed_figure": {"value": committed, "file": (roofline.get(src) or {}).get("file")}}
            roofline["mfma_busy_frac_of_2p4ghz"] = live_pmc["mfma_busy_frac_of_2p4ghz"]
        if roofline_hbm is not None and live_pmc is not None and "per_kernel_kb" in live_pmc:
            # the same FETCH_SIZE / WRITE_SIZE child passes, per kernel name, grouped by operator like tools/pmc_hbm_ops.py
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from pmc_hbm_ops import operator_of, short
                pk = live_pmc["per_kernel_kb"]
                steps_run = pk["steps_in_run"]
                per_op = {}
                for name in set(pk["FETCH_SIZE"]) | set(pk["WRITE_SIZE"]):
                    op = operator_of(name)
                    if op is None:
                        continue
                    nf, f_kb = pk["FETCH_SIZE"].get(name, (0, 0.0))
                    nw, w_kb = pk["WRITE_SIZE"].get(name, (0, 0.0))
                    fb_, wb_ = f_kb * 1024 * 2 / steps_run, w_kb * 1024 / steps_run
                    o = per_op.setdefault(op, {"hbm_bytes_per_step": 0.0, "kernels": {}})
                    o["hbm_bytes_per_step"] += fb_ + wb_
                    o["kernels"][short(name)] = {"launches_per_step": max(nf, nw) / steps_run, "fetch_bytes_per_step": fb_, "write_bytes_per_step": wb_}
                for rec in roofline_hbm["operators"]:
                    if rec["operator"] in per_op:
                        rec["traffic"] = per_op[rec["operator"]]["hbm_bytes_per_step"]
                        rec["traffic_kernels"] = per_op[rec["operator"]]["kernels"]
                committed = roofline_hbm.get("traffic_source")
                roofline_hbm["traffic_source"] = {
                    "how": "MEASURED IN THIS RUN: the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this command (roofline.traffic_source), "
                           "per kernel name over the whole pass / its 4 steps, FETCH x2 gfx950 correction, grouped by operator (tools/pmc_hbm_ops.py)",
                    "committed_file": None if committed is None else committed.get("file")}
            except Exception as e:              # the replayed figures stay
                roofline_hbm.setdefault("traffic_source", {})["live_measurement_skipped"] = repr(e)[:200]
            roofline["peak_assumes_clock_ghz"] = 2.4
            roofline["frac_at_delivered_clock"] = roofline["achieved"] / (roofline["peak"] * live_pmc["delivered_clock_ghz"] / 2.4)
    if args.breakdown and rank == 0:
        tot_ms = sum(c["ms"] for c in classes.values())
        for n_, c in sorted(classes.items(), key=lambda kv: -kv[1]["ms"]):
            print(f"  {n_:52s} launches/step {c['launches'] / args.steps:6.1f}  ms/step {c['ms'] / args.steps:8.2f} "
                  f"GFLOP/step {c['flops'] / args.steps / 1e9:9.1f}  TFLOP/s {c['flops'] / max(c['ms'], 1e-9) / 1e9:7.2f}"
                  f"  issued-MFMA TFLOP/s {c['exec_flops'] / max(c['ms'], 1e-9) / 1e9:7.2f}",
                  file=sys.stderr)
        per_step = len(launches) // args.steps
        for ms, n_, cin, cout, n_out, p, ex in sorted(launches[:per_step], key=lambda t: -t[0])[:int(os.environ.get("PCC_BENCH_TOP", "16"))]:
            if ms <= 0.0:
                continue
            print(f"    {n_[:34]:34s} {cin:4d}->{cout:<4d} rows {n_out:8d} nbrs/row {p / max(n_out, 1):5.1f}  {ms:7.3f} ms  "
                  f"alg {2.0 * p * cin * cout / ms / 1e9:6.1f} TF/s  issued {ex / ms / 1e9:6.1f} TF/s", file=sys.stderr)
        print(f"  conv total {tot_ms / args.steps:.2f} ms/step of {elapsed / args.steps * 1e3:.2f} ms/step; "
              f"t_enc {t_enc / args.steps * 1e3:.1f} ms  t_dec {t_dec / args.steps * 1e3:.1f} ms", file=sys.stderr)

    bits = pcc_amd.utils.count_bits(last["strings"])
    if blocks_mode and dist is not None:                      # every rank holds a share of the frame's streams
        bt = torch.tensor([bits], dtype=torch.int64, device=cdev)
        dist.all_reduce(bt)
        bits = int(bt.item())
    bpp = bits / N
    file_mode = None
    if args.file_mode and rank == 0:
        # t_file: the same frame through the container on disk (28-byte header, PCO1 coordinate payload, y, z)
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "frame.bin")
            te, tdc = [], []
            for it in range(4):
                Q = pcc_amd.SparseTensor(coordinates=q_coords, features=q_feats, device=dev)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                model.compress(x, Q, path=path)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                rec_f = model.decompress(path=path)
                torch.cuda.synchronize(); t2 = time.perf_counter()
                if it:
                    te.append(t1 - t0); tdc.append(t2 - t1)
            size = os.path.getsize(path)
            with open(path, "rb") as f:
                head = f.read(28)
        import struct
        coord_bytes = struct.unpack(">7i", head)[1]
        file_mode = {"t_enc_ms": 1e3 * sum(te) / len(te), "t_dec_ms": 1e3 * sum(tdc) / len(tdc), "file_bytes": size,
                     "coordinate_payload_bytes": coord_bytes, "bpp_file": 8.0 * size / N,
                     "decoded_points": int(rec_f.shape[0]),
                     "note": "coordinate payload = this build's lossless octree (PCO1), not G-PCC"}
    out = {
        "metric": "encode+decode Mpoints/sec",
        "value": n_total * args.steps / elapsed / 1e6,
        "unit": "Mpoints/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if blocks_mode else "weak",
        "vs_baseline": None,
        "dtype": ("bf16 operands on the wide convolutions, fp32 accumulation (opt-in mode, not the headline)" if args.bf16 else
                  "f32 data; wide-convolution products as 6 bf16 MFMA terms of an exact 3-way split (opt-in mode, not the headline)" if args.x3
                  else "f32"),
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: {cfg['grid']}^3 voxel sphere shell r={cfg['radius']}, N={N} points/frame, "
                               f"q=(0.5,0.5), configs/Ours.yaml, {'weights from ' + os.path.basename(args.weights) if args.weights else 'seeded random weights'}, in-memory compress+decompress, "
                               + (f"ONE frame per step cut into {args.block}^3 cubes spread over the ranks by point count"
                                  if blocks_mode else "one frame per rank per step"),
                   "points_per_frame": N,
                   "parallelism": (f"blocks x{world}" if blocks_mode else f"frames x{world}") if world > 1 else "single"},
        "t_enc_ms": t_enc / args.steps * 1e3,
        "t_dec_ms": t_dec / args.steps * 1e3,
        "step_ms": step_ms,
        "device_allocs_during_timed_steps": device_allocs_timed,     # hipMalloc calls of the caching allocator inside the timed region
        "gc": "objects alive after warm-up frozen (gc.freeze()), automatic collection off during the timed steps: a full collection "
              "otherwise walks ~215 k interpreter objects (80-135 ms) and the per-launch records of this script provoke one per 20 "
              "steps; no device memory depends on the collector (device_allocs_during_timed_steps)",
        "bpp": bpp,
        "conv_gflop_per_step": sum(c["flops"] + c["untimed_flops"] for c in classes.values()) / args.steps / 1e9,
        "roofline": roofline,
        "rccl": rccl,
    }
    if blocks_record is not None:
        out["blocks"] = blocks_record
    if x3_record is not None:
        out["split_bf16"] = x3_record
    if streamed_record is not None:
        out["streamed"] = streamed_record
    if roofline_hbm is not None:
        out["roofline_hbm"] = roofline_hbm
    if small_frame_record is not None:
        out["small_frame"] = small_frame_record
    if mid_frame_record is not None:
        out["mid_frame"] = mid_frame_record
    if train_record is not None:
        out["train_step"] = train_record
    if file_mode is not None:
        out["file_mode"] = file_mode
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sd = {n_: t.detach().cpu() for n_, t in model.state_dict().items()}
        out["cpu_baseline"] = cpu_baseline(model, sd, args.cpu_sample, dev, args.cpu_order)
        if args.cpu_order == "kernel":
            # the slower CPU implementation SURVEY.md §8d names (torch-CPU sgemm per offset + index_add_) is not re-run by the default
            # command (2.5 minutes); its newest committed measurement (`python bench.py --cpu-order blas`) rides along, labelled
            other = latest_profile("cpu_baseline_blas_order")
            if other and other.get("cpu_baseline"):
                ob = other["cpu_baseline"]
                out["cpu_baseline"]["blas_order_oracle"] = {
                    "value": ob["value"], "unit": ob["unit"], "cores": ob["cores"], "cpu_model": ob.get("cpu_model"), "sample": ob["sample"],
                    "parity_abs_diff": ob.get("parity", {}).get("abs_diff"), "decoded_voxels_differing": ob.get("parity", {}).get("decoded_voxels_differing"),
                    "replayed_from": {"file": other["_file"], "commit": other.get("commit"), "note": "not re-measured in this run"}}
    elif rank == 0:
        # N > 1 (and --no-cpu-baseline): the baseline is measured on rank 0 at N = 1 only (minutes of host work); carry the newest
        # committed N = 1 record along, labelled as replayed, so that a scaling line is as complete as the N = 1 line
        prev = latest_profile("bench_config2")
        cb = (prev or {}).get("cpu_baseline")
        if cb:
            cb = dict(cb)
            cb.setdefault("cpu_model", None)                 # records of rounds 1-3 did not name the CPU
            cb["replayed_from"] = {"file": prev["_file"], "commit": prev.get("commit"),
                                   "note": "measured by the N = 1 run of this command on one MI355X box's host cores; not re-measured in this run"}
        out["cpu_baseline"] = cb
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
