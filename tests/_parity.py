"""Shared rules of the HIP-vs-oracle codec comparisons.

Two oracles, two kinds of statement (oracle/nn.py):

* **"kernel" order — equality.**  The oracle sums every convolution as the one fused multiply-add chain the kernels
  document (oracle/chain.c).  Against it the HIP codec must produce the SAME BYTES: y and z streams, latent coordinates,
  k, decoded voxels and 8-bit colours (``assert_exact``).  Every ``compare_codec`` call checks this unless told not to.

* **"blas" order — the contract's tolerances.**  The independent restatement (gather -> sgemm -> index_add_, MKL's
  summation order).  BASELINE.json asks for bpp and D1 / Y-PSNR within 1e-3 (dB) on identical inputs.  On the BASELINE
  configurations (``strict=True``: configs 1, 2, 3) that bound is asserted DIRECTLY — no allowance per differing voxel —
  on the decoder fed identical latents (always), and on the end-to-end result whenever the two encoders coded the same
  symbols.  What the two fp32 implementations legitimately decide differently (a latent whose y - mu sits on .5 within
  their ~1e-7 difference, a scale on a table boundary, a top-k near-tie) is COUNTED — ``n_sym`` latents a whole step
  apart, ``flips_same`` voxels with the decoder fed identical latents, ``flips`` voxels end to end — recorded per case
  in gpurun_out/parity_counts.json and held against the committed tests/golden/parity_counts.json: a count may not
  exceed twice its committed value (+ 2).
  The one case the direct end-to-end bound cannot cover: ``n_sym > 0``.  Then the two sides hold DIFFERENT (equally
  valid) encodings of the frame — one of the 16 config-3 operating points (the 17 k-point frame at q = (0.4, 0.8): one
  latent of 0.5 M a step apart, which seeded random weights amplify to 0.0099 dB D1 / 0.039 dB Y; the config-2 frame
  also has one such latent and stays at 7.7e-5 / 2.6e-4 dB, which its test asserts directly) — and "within 1e-3 dB" of
  each other is not a property
  either implementation could have: the reference's own CPU and GPU builds would differ the same way.  For such a case
  the test REQUIRES the equality with the kernel-order oracle (which shows the HIP side's encoding is the documented
  arithmetic's, bit for bit), keeps the direct bound on the decoder fed identical latents, and bounds the end-to-end
  difference by 0.05 dB per differently-coded latent.
  The older allowances (a dB bound per differing voxel / per differently-rounded latent) remain only for the adversarial
  clouds and model variants (``strict=False``: tests/test_random_clouds.py, test_config_variants.py), where seeded random
  weights amplify one flipped latent of a 1,500-point cloud into more than 1e-3 dB.
"""
import json
import os

import numpy as np

D1_WORST_SQ = 64.0      # geometry term = mean over axes of squared offsets: an 8-voxel miss on every axis
Y_WORST_SQ = 1.0        # luma in [0, 1]
CONTRACT_DB = 1e-3      # BASELINE.json: D1-PSNR / Y-PSNR within 1e-3 dB
CONTRACT_BPP = 1e-3     # BASELINE.json: bpp within 1e-3

_HERE = os.path.dirname(os.path.abspath(__file__))
COUNTS_GOLDEN = os.path.join(_HERE, "golden", "parity_counts.json")
COUNTS_OUT = os.path.join(os.path.dirname(_HERE), "gpurun_out", "parity_counts.json")
_counts_seen = {}


def flip_bound_db(n_flips, n_points, mse, worst_sq):
    if n_flips == 0:
        return 0.0
    return 10.0 * np.log10(1.0 + n_flips * worst_sq / (n_points * max(mse, 1e-12)))


def voxel_flips(rec, o_rec):
    """voxels that are in one decoded set and not in the other"""
    def keys(r):
        c = np.asarray(r)[:, :3].astype(np.int64)
        return np.unique(((c[:, 0] + (1 << 20)) << 42) | ((c[:, 1] + (1 << 20)) << 21) | (c[:, 2] + (1 << 20)))
    return int(np.setxor1d(keys(rec), keys(o_rec), assume_unique=True).size)


def assert_psnr_parity(m, om, flips, n_points, tag=None):
    """non-strict rule — m / om: pc_metrics of the HIP and the oracle reconstruction against the same source"""
    for key, mse_key, worst in (("sym_psnr_mse", "mse", D1_WORST_SQ), ("sym_y_psnr", "y_mse", Y_WORST_SQ)):
        if not (np.isfinite(m[key]) or np.isfinite(om[key])):
            continue                                         # both lossless
        tol = CONTRACT_DB + flip_bound_db(flips, n_points, min(om["AB_" + mse_key], om["BA_" + mse_key]), worst)
        assert abs(m[key] - om[key]) <= tol, (tag, key, m[key], om[key], flips, tol)


def assert_contract(bpp, o_bpp, m, om, tag=None):
    """strict rule: BASELINE's bounds, directly"""
    assert abs(bpp - o_bpp) <= CONTRACT_BPP, (tag, "bpp", bpp, o_bpp)
    for key in ("sym_psnr_mse", "sym_y_psnr"):
        if np.isfinite(m[key]) or np.isfinite(om[key]):
            assert abs(m[key] - om[key]) <= CONTRACT_DB, (tag, key, m[key], om[key])


SYMBOL_FLIP_DB = 0.05       # non-strict end-to-end allowance per differently-rounded latent (seeded random weights amplify one step
                            # of one latent into ~0.01 dB on a 17 k-point frame; measured, tools/parity_diag.py)


def record_counts(tag, counts):
    """store the discrete-decision counts of one case and hold them against the committed values (module docstring)"""
    key = tag if isinstance(tag, str) else json.dumps(tag, default=str)
    _counts_seen[key] = counts
    try:
        os.makedirs(os.path.dirname(COUNTS_OUT), exist_ok=True)
        with open(COUNTS_OUT, "w") as f:
            json.dump(_counts_seen, f, indent=1, sort_keys=True)
    except OSError:
        pass
    committed = {}
    if os.path.exists(COUNTS_GOLDEN):
        with open(COUNTS_GOLDEN) as f:
            committed = json.load(f)
    want = committed.get(key)
    if want is None:
        return                                    # a new case: its first run defines the committed value
    for name, value in counts.items():
        if name in want and isinstance(value, int):
            assert value <= 2 * int(want[name]) + 2, (key, name, value, "committed", want[name])


def canonical(rec):
    """decoded cloud in (x, y, z) order as (int32 coordinates, uint8 colours)"""
    order = np.lexsort((rec[:, 2], rec[:, 1], rec[:, 0]))
    return rec[order, :3].astype(np.int32), np.rint(rec[order, 3:6] * 255.0).astype(np.uint8)


def assert_exact(oracle_codec, pts, qc, qf, strings, shape, k, coords, rec, tag=None, batch=None, rec_item=None):
    """the HIP result against the oracle in "kernel" summation order: the same bytes (module docstring)"""
    from oracle import coords as oc
    from oracle import nn as on
    was = on.set_order("kernel")
    try:
        o_strings, o_shape, o_k, o_coords = oracle_codec.compress(pts, qc, qf, batch=batch)
        assert shape == o_shape and k == o_k, (tag, "kernel order: shape / k")
        got_c = np.asarray(coords)
        assert np.array_equal(got_c[oc.sort_order(got_c)], o_coords[oc.sort_order(o_coords)]), (tag, "kernel order: latent coordinates")
        assert strings[1][0] == o_strings[1][0], (tag, "kernel order: z stream differs")
        assert strings[0][0] == o_strings[0][0], (tag, "kernel order: y stream differs")
        o_rec = oracle_codec.decompress(o_coords, o_strings, o_shape, o_k)
        if rec_item is not None:                                   # batch items may overlap in (x, y, z): compare per item
            o_item = oracle_codec.last_batch
            for i in np.unique(o_item):
                geo, col = canonical(rec[rec_item == i])
                o_geo, o_col = canonical(o_rec[o_item == i])
                assert np.array_equal(geo, o_geo) and np.array_equal(col, o_col), (tag, "kernel order: item", int(i))
            return
        geo, col = canonical(rec)
        o_geo, o_col = canonical(o_rec)
        assert np.array_equal(geo, o_geo), (tag, "kernel order: decoded voxel sets differ", voxel_flips(rec, o_rec))
        assert np.array_equal(col, o_col), (tag, "kernel order: decoded colours differ", int((col != o_col).sum()))
    finally:
        on.set_order(was)


class _Rows:
    """(C, F) of an oracle tensor that was computed elsewhere (tests/_config2_blas_worker.py)"""

    def __init__(self, C, F):
        import torch
        self.C, self.F = C, torch.from_numpy(np.ascontiguousarray(F))


def compare_codec(pcc, model, oracle_codec, pts, qc, qf, tag=None, dev="cuda:0", strict=False, exact=True, oracle_results=None):
    """HIP codec vs CPU oracle on one frame (module docstring).  Returns a dict of what was measured.
    ``oracle_results``: the BLAS-order oracle's outputs for this very frame computed by a background process (a dict of arrays:
    tests/_config2_blas_worker.py) instead of calling ``oracle_codec`` here."""
    import torch
    from oracle import nn as on
    from oracle.codec import count_bits
    from oracle.metrics import pc_metrics
    assert on.ORDER == "blas", "compare_codec's tolerances belong to the BLAS-order oracle"
    N = pts.shape[0]
    x = torch.from_numpy(pts).to(dev)
    Q = pcc.SparseTensor(coordinates=torch.from_numpy(qc).to(dev), features=torch.from_numpy(qf).to(dev), device=dev)
    strings, shape, k, coords = model.compress(x, Q)
    if oracle_results is None:
        o_strings, o_shape, o_k, o_coords = oracle_codec.compress(pts, qc, qf)
    else:
        R = oracle_results
        o_strings = [[R["y_stream"].tobytes()], [R["z_stream"].tobytes()]]
        o_shape, o_k, o_coords = [int(v) for v in R["shape"]], [[int(v) for v in row] for row in R["k"]], R["coords"]
    # exact: structure
    assert shape == o_shape and k == o_k, tag
    got_c = coords.cpu().numpy()
    assert got_c.shape == o_coords.shape and set(map(tuple, got_c.tolist())) == set(map(tuple, o_coords.tolist())), tag
    bpp, o_bpp = count_bits(strings) / N, count_bits(o_strings) / N
    if not strict:
        assert abs(bpp - o_bpp) <= 2e-3 * o_bpp + 1e-3, (tag, bpp, o_bpp)
    # each side decodes its own stream
    c8 = pcc.CoordMap(coords.to(torch.int32).contiguous(), 8, nbatch=1)
    with torch.no_grad():
        y_hat, Q_hat = model.entropy_model.decompress([c8, c8.down().down()], strings, shape)
    rec = model.reconstruct(y_hat, Q_hat, k).cpu().numpy()
    if oracle_results is None:
        o_rec = oracle_codec.decompress(o_coords, o_strings, o_shape, o_k)
        o_y, o_Q = oracle_codec.last_dec["y_hat"], oracle_codec.last_dec["Q_hat"]
    else:
        o_rec = oracle_results["rec"]
        o_y, o_Q = _Rows(oracle_results["y_C"], oracle_results["y_F"]), _Rows(oracle_results["Q_C"], oracle_results["Q_F"])
    assert rec.shape == o_rec.shape == (N, 6), tag
    # stage 1, the encoders' discrete decisions: decoded latents equal except for whole quantisation steps (counted)
    assert np.array_equal(y_hat.C.cpu().numpy(), o_y.C), tag                   # both in canonical (bitstream) order
    d = (y_hat.F.cpu() - o_y.F).abs()
    stepped = d > 0.5
    n_sym = int(stepped.sum())
    assert bool(((d[stepped] - torch.round(d[stepped])).abs() < 1e-3).all()), tag          # whole steps
    assert float(d[~stepped].max()) <= 1e-4 * max(1.0, float(o_y.F.abs().max())), (tag, float(d[~stepped].max()))   # the means agree
    if strings[1] != o_strings[1]:
        n_sym += 1                                                             # a hyper-latent coded differently
    # stage 2, the decoder on identical latents: the oracle's y_hat / Q_hat through the HIP synthesis
    yo = pcc.SparseTensor(o_y.F.to(dev).contiguous(), coordinate_map=pcc.CoordMap(torch.from_numpy(np.ascontiguousarray(o_y.C, dtype=np.int32)).to(dev), 8, nbatch=1))
    Qo = pcc.SparseTensor(o_Q.F.to(dev).contiguous(), coordinate_map=pcc.CoordMap(torch.from_numpy(np.ascontiguousarray(o_Q.C, dtype=np.int32)).to(dev), 8, nbatch=1))
    rec_same = model.reconstruct(yo, Qo, k).cpu().numpy()
    flips_same = voxel_flips(rec_same, o_rec)
    om = pc_metrics(pts, o_rec)
    m_same = pc_metrics(pts, rec_same)
    flips = voxel_flips(rec, o_rec)
    m = pc_metrics(pts, rec)
    counts = dict(n=int(N), n_sym=n_sym, flips_same=int(flips_same), flips=int(flips))
    if strict:
        # BASELINE's bounds directly: the decoder on identical latents, and end to end on own streams
        assert_contract(o_bpp, o_bpp, m_same, om, (tag, "decoder on identical latents"))
        if n_sym == 0:
            assert_contract(bpp, o_bpp, m, om, tag)
        else:
            # different encodings of the frame (module docstring): the kernel-order equality below is then mandatory
            assert exact, (tag, "n_sym > 0 needs the kernel-order comparison (exact=True, or \"elsewhere\" naming the test that makes it)")
            assert abs(bpp - o_bpp) <= CONTRACT_BPP, (tag, "bpp", bpp, o_bpp)
            for key in ("sym_psnr_mse", "sym_y_psnr"):
                if np.isfinite(m[key]) or np.isfinite(om[key]):
                    assert abs(m[key] - om[key]) <= CONTRACT_DB + SYMBOL_FLIP_DB * n_sym, (tag, key, m[key], om[key], n_sym)
        counts["d_d1_db"] = round(abs(float(m["sym_psnr_mse"]) - float(om["sym_psnr_mse"])), 6) if np.isfinite(m["sym_psnr_mse"]) else 0.0
        counts["d_y_db"] = round(abs(float(m["sym_y_psnr"]) - float(om["sym_y_psnr"])), 6) if np.isfinite(m["sym_y_psnr"]) else 0.0
        counts["d_bpp"] = round(abs(bpp - o_bpp), 6)
        record_counts(tag, counts)
    else:
        assert n_sym <= max(2, int(2e-5 * d.numel())), (tag, "latents rounded differently", n_sym)
        assert flips_same <= max(4, int(2e-3 * N)), (tag, "decoder on identical latents", flips_same)
        assert_psnr_parity(m_same, om, flips_same, N, (tag, "decoder on identical latents"))
        if n_sym == 0:
            # more near-ties than in stage 2: here the two decoders also start from latents whose means differ in the last bits
            assert flips <= max(8, int(5e-3 * N)), (tag, flips)
            assert_psnr_parity(m, om, flips, N, tag)                           # 1e-3 dB (+ the voxel-flip bound above)
        else:
            for key in ("sym_psnr_mse", "sym_y_psnr"):
                if np.isfinite(m[key]) or np.isfinite(om[key]):
                    assert abs(m[key] - om[key]) <= CONTRACT_DB + SYMBOL_FLIP_DB * n_sym, (tag, key, m[key], om[key], n_sym)
    if exact is True:                      # "elsewhere": the same frame's equality test lives in another test (named by the caller)
        assert_exact(oracle_codec, pts, qc, qf, strings, shape, k, got_c, rec, tag)
    return dict(bpp=bpp, o_bpp=o_bpp, m=m, om=om, flips=flips, flips_same=flips_same, n_sym=n_sym,
                streams_equal=(strings == o_strings), d_d1=abs(m["sym_psnr_mse"] - om["sym_psnr_mse"]),
                d_y=abs(m["sym_y_psnr"] - om["sym_y_psnr"]))
