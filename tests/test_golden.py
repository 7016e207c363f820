"""The CPU oracle against the committed golden vectors (tests/golden/, made by make_golden.py).

Integer / byte facts must match exactly.  Facts that pass through fp32 sgemm (stream bytes, bpp,
PSNR) are compared exactly when the host reproduces the generating machine's summation order and
within a tolerance otherwise (a different CPU may flip a few symbols on rounding boundaries).
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import coords as oc
from oracle import rans as crans
from oracle.codec import count_bits
from oracle.entropy import GaussianConditional
from oracle.metrics import pc_metrics

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLD, name + ".json")) as f:
        return json.load(f)


def sha(b):
    return hashlib.sha256(b).hexdigest()


def test_integer_kats():
    g = load("integer_kats")
    c = np.asarray(g["coords"], dtype=np.int32)
    assert oc.kernel_map(c, c, 3, 1).tolist() == g["kernel_map_k3_s1"]
    assert oc.stride_map(c, 1).tolist() == g["stride_map_ts1"]
    c2 = c * np.array([1, 2, 2, 2])
    assert oc.children(c2, 2, 2).tolist() == g["children_k2_ts2"]
    assert oc.children(c2, 2, 3).shape[0] == g["children_k3_ts2_count"]
    assert oc.sort_order(c[::-1]).tolist() == g["sort_order"]


def test_entropy_kats():
    g = load("entropy_kats")
    gc = GaussianConditional()
    gc.update()
    assert gc.cdf_length.tolist() == g["gc_cdf_length"] and gc.offset.tolist() == g["gc_offset"]
    assert sha(np.ascontiguousarray(gc.cdf).tobytes()) == g["gc_cdf_sha256"]
    for key, want in g["pmf_cases"].items():
        assert crans.pmf_to_quantized_cdf(json.loads(key)).tolist() == want
    rng = np.random.default_rng(g["rans_seed"])
    n = g["rans_n"]
    idx = rng.integers(0, 40, n).astype(np.int32)
    sym = np.rint(rng.normal(0, 1, n) * gc.scale_table.numpy()[idx]).astype(np.int32)
    sym[::211] = rng.integers(-5000, 5000, sym[::211].shape)
    data = crans.encode_with_indexes(sym, idx, gc.cdf, gc.cdf_length, gc.offset)
    assert len(data) == g["rans_bytes"] and sha(data) == g["rans_sha256"]
    assert (crans.decode_with_indexes(data, idx, gc.cdf, gc.cdf_length, gc.offset) == sym).all()


def test_product_host_coder_matches_golden_stream(pcc):
    """libpcc_hip.so's host rANS (no GPU needed) must emit the golden bytes too."""
    from pcc_amd import entropy as pe
    g = load("entropy_kats")
    gc = GaussianConditional()
    gc.update()
    rng = np.random.default_rng(g["rans_seed"])
    n = g["rans_n"]
    idx = rng.integers(0, 40, n).astype(np.int32)
    sym = np.rint(rng.normal(0, 1, n) * gc.scale_table.numpy()[idx]).astype(np.int32)
    sym[::211] = rng.integers(-5000, 5000, sym[::211].shape)
    data = pe._rans_encode(sym, idx, gc.cdf, gc.cdf_length, gc.offset)
    assert sha(data) == g["rans_sha256"]
    assert (pe._rans_decode(data, idx, gc.cdf, gc.cdf_length, gc.offset) == sym).all()


def test_config1_oracle_end_to_end(pcc, seeded_state_dict, oracle_codec):
    g = load("config1_oracle")
    assert sha(b"".join(v.numpy().tobytes() for _, v in sorted(seeded_state_dict.items()))) == g["state_dict_sha256"]
    syn = pcc.synthetic
    torch.set_num_threads(1)
    pts = syn.sphere_shell(**syn.CONFIG1)
    assert pts.shape[0] == g["n_points"]
    qc, qf = syn.uniform_qmap(pts[:, :3], 0.5, 0.5)
    strings, shape, k, coords = oracle_codec.compress(pts, qc, qf)
    assert k == g["k"] and shape == g["shape"]                                    # transforms.py:89-127 order
    assert sha(np.ascontiguousarray(coords[oc.sort_order(coords)]).tobytes()) == g["latent_coords_sha256"]
    rec = oracle_codec.decompress(coords, strings, shape, k)
    assert rec.shape == (g["n_points"], 6)
    met = pc_metrics(pts, rec)
    exact = sha(strings[0][0]) == g["sha256_y"] and sha(strings[1][0]) == g["sha256_z"]
    if exact:
        order = np.lexsort((rec[:, 2], rec[:, 1], rec[:, 0]))
        assert sha(np.ascontiguousarray(rec[order, :3].astype(np.int32)).tobytes()) == g["recon_geometry_sha256"]
        assert abs(met["sym_psnr_mse"] - g["d1_psnr"]) < 1e-9 and abs(met["sym_y_psnr"] - g["y_psnr"]) < 1e-6
    else:   # other CPU / BLAS summation order
        assert abs(len(strings[0][0]) - g["len_y"]) <= 0.01 * g["len_y"] + 8
        assert abs(count_bits(strings) / pts.shape[0] - g["bpp"]) <= 5e-3 * g["bpp"]
        assert abs(met["sym_psnr_mse"] - g["d1_psnr"]) < 0.05 and abs(met["sym_y_psnr"] - g["y_psnr"]) < 0.05


def test_kernel_order_oracle_reproduces_its_golden_bytes(pcc, oracle_codec):
    """The oracle in "kernel" summation order (oracle/chain.c) is plain fused multiply-add chains — no BLAS, so its bytes
    are the same on every host: streams, latent coordinates, decoded geometry and colours must equal the committed hashes
    exactly.  The GPU twin of this test (tests/test_exact_parity.py) holds the HIP path to the same hashes."""
    import sys
    sys.path.insert(0, GOLD)
    from make_golden import KERNEL_ORDER_FRAMES, recon_sha
    from oracle import nn as on
    g = load("kernel_order_frames")
    assert set(g) == set(KERNEL_ORDER_FRAMES)
    syn = pcc.synthetic
    was = on.set_order("kernel")
    try:
        for name, (shell, (qg, qa)) in KERNEL_ORDER_FRAMES.items():
            want = g[name]
            pts = syn.sphere_shell(**shell)
            qc, qf = syn.uniform_qmap(pts[:, :3], qg, qa)
            strings, shape, k, coords = oracle_codec.compress(pts, qc, qf)
            assert (pts.shape[0], k, shape) == (want["n_points"], want["k"], want["shape"]), name
            assert (sha(strings[0][0]), sha(strings[1][0])) == (want["sha256_y"], want["sha256_z"]), name
            assert sha(np.ascontiguousarray(coords[oc.sort_order(coords)]).tobytes()) == want["latent_coords_sha256"], name
            rec = oracle_codec.decompress(coords, strings, shape, k)
            assert recon_sha(rec) == (want["recon_geometry_sha256"], want["recon_colour_sha256"]), name
    finally:
        on.set_order(was)


def test_two_hyperprior_oracle_reproduces_its_golden_bytes(pcc):
    """The two-hyperprior variant (model/model.py:22-24, model/entropy_models.py:104-250) in "kernel" order: four streams, shapes, k,
    latent coordinates and the decoded cloud equal the committed hashes (tests/test_two_hyperprior.py holds the HIP path to them)."""
    import sys
    sys.path.insert(0, GOLD)
    from make_golden import TWO_HYPERPRIOR_FRAMES, recon_sha
    from oracle import nn as on
    from oracle.codec import Codec
    g = load("two_hyperprior_frames")
    assert set(g) == set(TWO_HYPERPRIOR_FRAMES)
    syn = pcc.synthetic
    model = syn.make_model(seed=0, device="cpu", config=syn.TWO_HYPERPRIOR_CONFIG)
    codec = Codec(model.state_dict(), syn.TWO_HYPERPRIOR_CONFIG)
    codec.update()
    assert float(codec.aux_loss()) > 0
    was = on.set_order("kernel")
    try:
        for name, (shell, (qg, qa)) in TWO_HYPERPRIOR_FRAMES.items():
            want = g[name]
            pts = syn.sphere_shell(**shell)
            qc, qf = syn.uniform_qmap(pts[:, :3], qg, qa)
            strings, shape, k, coords = codec.compress(pts, qc, qf)
            assert (pts.shape[0], k, shape) == (want["n_points"], want["k"], want["shape"]), name
            flat = [strings[0][0][0], strings[0][1][0], strings[1][0][0], strings[1][1][0]]
            assert [sha(b) for b in flat] == want["sha256"], name
            assert want["q_symbols_min_max"][1] - want["q_symbols_min_max"][0] >= 8, "the q-map's symbols span several bins"
            assert sha(np.ascontiguousarray(coords[oc.sort_order(coords)]).tobytes()) == want["latent_coords_sha256"], name
            rec = codec.decompress(coords, strings, shape, k)
            assert recon_sha(rec) == (want["recon_geometry_sha256"], want["recon_colour_sha256"]), name
    finally:
        on.set_order(was)


def test_chain_convolution_equals_its_scalar_statement():
    """oracle/chain.c: the vectorised, windowed, threaded chain against the four-line scalar loop it restates, bit for bit,
    over thin / MFMA-order / narrow shapes, kernel sizes 1, 2, 3, sparse and dense neighbourhoods"""
    from oracle import chain
    rng = np.random.default_rng(5)
    for (n_in, n_out, cin, cout, K, mfma, dens) in [(300, 500, 64, 64, 27, 1, 0.4), (300, 200, 128, 1, 27, 1, 0.8), (3000, 1999, 16, 2, 27, 0, 0.1),
                                                    (997, 997, 128, 128, 1, 1, 1.0), (500, 1000, 128, 192, 8, 1, 0.12), (40, 40, 2, 128, 27, 0, 0.5),
                                                    (7, 5, 32, 3, 27, 1, 0.5), (64, 64, 192, 256, 27, 1, 0.05)]:
        fin = rng.standard_normal((n_in, cin)).astype(np.float32)
        w = rng.standard_normal((K, cin, cout)).astype(np.float32)
        nbr = None
        if K > 1:
            nbr = rng.integers(0, n_in, (n_out, K))
            nbr[rng.random((n_out, K)) > dens] = -1
        else:
            n_out = n_in
        a = chain.conv_chain(fin, w, nbr, n_out, mfma)
        b = chain.conv_chain(fin, w, nbr, n_out, mfma, scalar=True)
        assert np.array_equal(a, b), (n_in, n_out, cin, cout, K, mfma)
    # the visit order matters (else the two orders would be one): MFMA order differs from ascending order somewhere
    fin = rng.standard_normal((50, 64)).astype(np.float32)
    w = rng.standard_normal((1, 64, 32)).astype(np.float32)
    assert not np.array_equal(chain.conv_chain(fin, w, None, 50, True), chain.conv_chain(fin, w, None, 50, False))
    # and a hand-checkable case: one row, one offset, 8 channels of powers of two whose sum depends on the order
    x = np.array([[2.0 ** 24, 1.0, 1.0, 1.0, -2.0 ** 24, 1.0, 1.0, 1.0]], np.float32)
    w1 = np.ones((1, 8, 1), np.float32)
    asc = chain.conv_chain(x, w1, None, 1, False)[0, 0]       # ((2^24 + 1 + 1 + 1) - 2^24) + 1 + 1 + 1: the first three ones are absorbed
    vis = chain.conv_chain(x, w1, None, 1, True)[0, 0]        # 2^24 - 2^24 first (channels 0, 4), then six ones
    assert (asc, vis) == (3.0, 6.0), (asc, vis)
