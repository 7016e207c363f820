#!/usr/bin/env python3
"""Generates tests/golden/*.json|npz — the golden vectors of the hot path.

The reference cannot be imported here (``ModuleNotFoundError: No module named 'MinkowskiEngine'``;
compressai, open3d, bitstream are absent as well) and ships no fixtures, so the vectors are produced
by the CPU oracle on seeded inputs and committed together with this script.  They pin the oracle
against accidental drift and give the GPU tests size-independent facts to check.

    python tests/golden/make_golden.py        # rewrites the fixtures in place
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import pcc_amd                                   # noqa: E402  (model definition + seeded init only; no HIP calls)
from oracle import coords as oc                  # noqa: E402
from oracle.codec import Codec, count_bits       # noqa: E402
from oracle.entropy import GaussianConditional   # noqa: E402
from oracle.metrics import pc_metrics            # noqa: E402
from oracle import rans as crans                 # noqa: E402


def sha(b):
    return hashlib.sha256(b).hexdigest()


def config1():
    syn = pcc_amd.synthetic
    torch.set_num_threads(1)                     # fixed summation order inside MKL for reproducible bytes
    model = syn.make_model(seed=0, device="cpu")
    codec = Codec(model.state_dict())
    codec.update()
    pts = syn.sphere_shell(**syn.CONFIG1)
    qc, qf = syn.uniform_qmap(pts[:, :3], 0.5, 0.5)
    strings, shape, k, coords = codec.compress(pts, qc, qf)
    rec = codec.decompress(coords, strings, shape, k)
    met = pc_metrics(pts, rec)
    c = np.concatenate([np.zeros((pts.shape[0], 1)), pts[:, :3]], axis=1).astype(np.int32)
    sizes, cc = [], c
    for ts in (1, 2, 4, 8, 16):
        cc = oc.stride_map(cc, ts)
        sizes.append(int(cc.shape[0]))
    order = np.lexsort((rec[:, 2], rec[:, 1], rec[:, 0]))
    out = {
        "workload": "32^3 sphere shell, centre 15.5, |r-15| < 0.875, q=(0.5,0.5), seeded_init(seed=0), torch threads = 1",
        "n_points": int(pts.shape[0]),
        "n_per_stride": sizes,
        "k": k,
        "shape": shape,
        "len_y": len(strings[0][0]),
        "len_z": len(strings[1][0]),
        "sha256_y": sha(strings[0][0]),
        "sha256_z": sha(strings[1][0]),
        "bpp": count_bits(strings) / pts.shape[0],
        "d1_psnr": float(met["sym_psnr_mse"]),
        "y_psnr": float(met["sym_y_psnr"]),
        "latent_coords_sha256": sha(np.ascontiguousarray(coords[oc.sort_order(coords)]).tobytes()),
        "recon_geometry_sha256": sha(np.ascontiguousarray(rec[order, :3].astype(np.int32)).tobytes()),
        "parameter_count": int(sum(p.numel() for p in model.parameters())),
        "state_dict_sha256": sha(b"".join(v.detach().cpu().numpy().tobytes() for _, v in sorted(model.state_dict().items()))),
    }
    return out


def integer_kats():
    c = np.array([[0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 2, 0], [0, 1, 1, 1], [1, 0, 0, 0]], dtype=np.int32)
    return {
        "coords": c.tolist(),
        "kernel_map_k3_s1": oc.kernel_map(c, c, 3, 1).tolist(),
        "stride_map_ts1": oc.stride_map(c, 1).tolist(),
        "children_k2_ts2": oc.children(c * np.array([1, 2, 2, 2]), 2, 2).tolist(),
        "children_k3_ts2_count": int(oc.children(c * np.array([1, 2, 2, 2]), 2, 3).shape[0]),
        "sort_order": oc.sort_order(c[::-1]).tolist(),
    }


def entropy_kats():
    gc = GaussianConditional()
    gc.update()
    rng = np.random.default_rng(7)
    n = 3000
    idx = rng.integers(0, 40, n).astype(np.int32)
    sym = np.rint(rng.normal(0, 1, n) * gc.scale_table.numpy()[idx]).astype(np.int32)
    sym[::211] = rng.integers(-5000, 5000, sym[::211].shape)
    data = crans.encode_with_indexes(sym, idx, gc.cdf, gc.cdf_length, gc.offset)
    return {
        "gc_cdf_sha256": sha(np.ascontiguousarray(gc.cdf).tobytes()),
        "gc_cdf_length": gc.cdf_length.tolist(),
        "gc_offset": gc.offset.tolist(),
        "rans_seed": 7, "rans_n": n, "rans_bytes": len(data), "rans_sha256": sha(data),
        "pmf_cases": {"[0.5,0.25,0.25]": crans.pmf_to_quantized_cdf([0.5, 0.25, 0.25]).tolist(),
                      "[1.0,0.0]": crans.pmf_to_quantized_cdf([1.0, 0.0]).tolist(),
                      "[0.25,0.0,0.5,0.25]": crans.pmf_to_quantized_cdf([0.25, 0.0, 0.5, 0.25]).tolist()},
    }


KERNEL_ORDER_FRAMES = {           # name -> (shell, (q_g, q_a)); the shells of tests/test_hip_codec.py and tests/test_exact_parity.py
    "config1_32": (dict(grid=32, radius=15.0, half_width=0.875), (0.5, 0.5)),
    "shell_64_q01_02": (dict(grid=64, radius=27.0, half_width=0.6), (0.1, 0.2)),
    "shell_96": (dict(grid=96, radius=40.0, half_width=0.5), (0.5, 0.5)),
}


def recon_sha(rec):
    """sha256 of the decoded cloud in canonical (x, y, z) order: int32 coordinates, then the colours as 8-bit integers"""
    order = np.lexsort((rec[:, 2], rec[:, 1], rec[:, 0]))
    geo = np.ascontiguousarray(rec[order, :3].astype(np.int32)).tobytes()
    col = np.ascontiguousarray(np.rint(rec[order, 3:6] * 255.0).astype(np.uint8)).tobytes()
    return sha(geo), sha(col)


def kernel_order_frames():
    """The oracle in "kernel" summation order (oracle/chain.c: one fused multiply-add chain per output element, no BLAS):
    its bytes do not depend on the host's BLAS, thread count or vector width, and the product must reproduce them EXACTLY —
    streams, latent coordinates, decoded geometry and 8-bit colours (tests/test_golden.py on the CPU for the oracle,
    tests/test_exact_parity.py on the GPU for the HIP path)."""
    from oracle import nn as on
    syn = pcc_amd.synthetic
    model = syn.make_model(seed=0, device="cpu")
    codec = Codec(model.state_dict())
    codec.update()
    out = {}
    was = on.set_order("kernel")
    try:
        for name, (shell, (qg, qa)) in KERNEL_ORDER_FRAMES.items():
            pts = syn.sphere_shell(**shell)
            qc, qf = syn.uniform_qmap(pts[:, :3], qg, qa)
            strings, shape, k, coords = codec.compress(pts, qc, qf)
            rec = codec.decompress(coords, strings, shape, k)
            geo, col = recon_sha(rec)
            out[name] = {"shell": shell, "q": [qg, qa], "n_points": int(pts.shape[0]), "k": k, "shape": shape,
                         "len_y": len(strings[0][0]), "len_z": len(strings[1][0]),
                         "sha256_y": sha(strings[0][0]), "sha256_z": sha(strings[1][0]),
                         "latent_coords_sha256": sha(np.ascontiguousarray(coords[oc.sort_order(coords)]).tobytes()),
                         "recon_geometry_sha256": geo, "recon_colour_sha256": col}
    finally:
        on.set_order(was)
    return out


TWO_HYPERPRIOR_FRAMES = {         # the two-hyperprior variant (model/model.py:22-24; pcc_amd.synthetic.TWO_HYPERPRIOR_CONFIG)
    "config1_32": (dict(grid=32, radius=15.0, half_width=0.875), (0.5, 0.5)),
    "shell_64_q02_04": (dict(grid=64, radius=27.0, half_width=0.6), (0.2, 0.4)),
}


def two_hyperprior_frames():
    """The oracle of the two-hyperprior variant in "kernel" summation order: four streams ([y, z] of the latents, [y, z] of the
    stride-8 q-map), shapes, k, latent coordinates, decoded cloud — the bytes tests/test_golden.py (CPU, the oracle) and
    tests/test_two_hyperprior.py (GPU, the HIP path) must reproduce."""
    from oracle import nn as on
    syn = pcc_amd.synthetic
    model = syn.make_model(seed=0, device="cpu", config=syn.TWO_HYPERPRIOR_CONFIG)
    codec = Codec(model.state_dict(), syn.TWO_HYPERPRIOR_CONFIG)
    codec.update()
    out = {}
    was = on.set_order("kernel")
    try:
        for name, (shell, (qg, qa)) in TWO_HYPERPRIOR_FRAMES.items():
            pts = syn.sphere_shell(**shell)
            qc, qf = syn.uniform_qmap(pts[:, :3], qg, qa)
            strings, shape, k, coords = codec.compress(pts, qc, qf)
            rec = codec.decompress(coords, strings, shape, k)
            geo, col = recon_sha(rec)
            flat = [strings[0][0][0], strings[0][1][0], strings[1][0][0], strings[1][1][0]]       # y, z of the latents; y, z of the q-map
            out[name] = {"shell": shell, "q": [qg, qa], "n_points": int(pts.shape[0]), "k": k, "shape": shape,
                         "len": [len(b) for b in flat], "sha256": [sha(b) for b in flat],
                         "q_symbols_min_max": [int(torch.round(codec.last_q["y"].F - codec.last_q["means"][0].t()).min()),
                                               int(torch.round(codec.last_q["y"].F - codec.last_q["means"][0].t()).max())],
                         "latent_coords_sha256": sha(np.ascontiguousarray(coords[oc.sort_order(coords)]).tobytes()),
                         "recon_geometry_sha256": geo, "recon_colour_sha256": col}
    finally:
        on.set_order(was)
    return out


if __name__ == "__main__":
    fixtures = {"config1_oracle": config1(), "integer_kats": integer_kats(), "entropy_kats": entropy_kats(),
                "kernel_order_frames": kernel_order_frames(), "two_hyperprior_frames": two_hyperprior_frames()}
    for name, obj in fixtures.items():
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(obj, f, indent=1, sort_keys=True)
        print("wrote", name)
