"""Quality metrics on the GPU: D1 (point-to-point) PSNR, Y/U/V PSNR, and Bjontegaard deltas.

``PointCloudMetric`` mirrors /root/reference/metrics/metric.py:6-189 (same result keys, same
formulas, same two modes of ``compute_pointcloud_metrics``); the open3d KD-tree association
(metric.py:36-43) is replaced by an exact voxel-hash search (csrc/metrics.hip), so clouds are given
as ``[N, 6]`` tensors on the GPU (xyz voxel indices, rgb in [0, 1]) instead of open3d objects or
PLY paths.  Equidistant neighbours — the rule on a lattice, and resolved by KD-tree visiting order
in the reference — resolve to the smallest (x, y, z) here.

``Bjontegaard_Model`` / ``Bjontegaard_Delta`` mirror metrics/bjontegaard.py:6-79 (cubic fits in the
log10-rate domain; host-side numpy, a handful of points per curve).
"""
import math

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr
from .sparse import CoordMap

_RADII = (2, 8, 32, 128, 1024)      # widening schedule of the shell search


def _as_cloud(pc, device):
    if not torch.is_tensor(pc):
        pc = torch.as_tensor(np.asarray(pc))
    if pc.dim() != 2 or pc.shape[1] < 6:
        raise ValueError("a cloud is a [N, 6] array: x, y, z, r, g, b")
    pc = pc.to(device)
    xyz = pc[:, :3]
    ixyz = torch.round(xyz.double()).to(torch.int32)
    if not torch.equal(ixyz.to(xyz.dtype), xyz):
        raise ValueError("metrics run on voxelised clouds: coordinates must be integers")
    coords = torch.cat([torch.zeros((pc.shape[0], 1), dtype=torch.int32, device=device), ixyz], dim=1).contiguous()
    return coords, pc[:, 3:6].to(torch.float64).contiguous()


def _drop_duplicated_points(coords, rgb):
    """first occurrence wins, like o3d's remove_duplicated_points (metric.py:18-21)"""
    key = (coords[:, 1].long() << 42) | ((coords[:, 2].long() & 0x1FFFFF) << 21) | (coords[:, 3].long() & 0x1FFFFF)
    uniq, inv = torch.unique(key, return_inverse=True)
    if uniq.numel() == key.numel():
        return coords, rgb
    first = torch.full((uniq.numel(),), key.numel(), dtype=torch.long, device=key.device)
    first.scatter_reduce_(0, inv, torch.arange(key.numel(), device=key.device), reduce="amin")
    first = torch.sort(first).values
    return coords[first].contiguous(), rgb[first].contiguous()


def nearest_neighbours(query_coords, target_map, target_rgb=None):
    """For every query voxel: (row of the nearest target voxel, squared distance, number of
    equidistant nearest voxels, sum of their colours or None).  Exact."""
    L = _lib.lib()
    keys, vals, cap = target_map.table()
    dev = query_coords.device
    nq = int(query_coords.shape[0])
    idx = torch.empty(nq, dtype=torch.int32, device=dev)
    d2 = torch.empty(nq, dtype=torch.int64, device=dev)
    ties = torch.empty(nq, dtype=torch.int32, device=dev)
    rgb64 = None if target_rgb is None else target_rgb.to(torch.float64).contiguous()
    tsum = None if rgb64 is None else torch.empty((nq, 3), dtype=torch.float64, device=dev)
    if target_map.n == 0:
        raise ValueError("nearest_neighbours: empty target cloud")
    todo = None                                    # rows still unresolved (None = all)
    for radius in _RADII:
        if todo is None:
            q, o_idx, o_d2, o_t, o_s = query_coords, idx, d2, ties, tsum
        else:
            q = query_coords[todo].contiguous()
            m = int(q.shape[0])
            o_idx = torch.empty(m, dtype=torch.int32, device=dev)
            o_d2 = torch.empty(m, dtype=torch.int64, device=dev)
            o_t = torch.empty(m, dtype=torch.int32, device=dev)
            o_s = None if rgb64 is None else torch.empty((m, 3), dtype=torch.float64, device=dev)
        check(L.pcc_nn_search(ptr(q), q.shape[0], ptr(keys), ptr(vals), cap, target_map.stride, ptr(rgb64), radius, ptr(o_idx), ptr(o_d2), ptr(o_t),
                              ptr(o_s), _lib.stream()))
        if todo is not None:
            idx[todo], d2[todo], ties[todo] = o_idx, o_d2, o_t
            if tsum is not None:
                tsum[todo] = o_s
        todo = torch.nonzero(idx < 0).flatten()
        if todo.numel() == 0:
            break
    else:
        raise RuntimeError("nearest_neighbours: %d queries farther than %d voxels from the target" % (todo.numel(), _RADII[-1]))
    return idx, d2, ties, tsum


def rgb_to_yuv(rgb):
    """BT.709 after the reference's truncating 8-bit cast (metric.py:171-189); rgb float64 in [0,1]."""
    scale = bool(rgb.max() <= 1.0)
    if scale:
        rgb = (rgb * 255).to(torch.uint8)
    c = rgb.to(torch.float64)
    yuv = torch.stack([0.2126 * c[:, 0] + 0.7152 * c[:, 1] + 0.0722 * c[:, 2],
                       -0.1146 * c[:, 0] - 0.3854 * c[:, 1] + 0.5 * c[:, 2],
                       0.5 * c[:, 0] - 0.4542 * c[:, 1] - 0.0458 * c[:, 2]], dim=1).to(torch.float32)
    if scale:
        yuv = yuv / _scalar(255.0, yuv)      # a tensor divisor: torch turns x / python_float into x * (1 / float)
        yuv[:, 1] += 0.5
        yuv[:, 2] += 0.5
    return yuv


def _scalar(v, like):
    return torch.tensor(v, dtype=like.dtype, device=like.device)


def _round_to_8bit(c):
    """clip(round(c * 255) / 255, 0, 1) with numpy's true division (metric.py:149-150)"""
    return torch.clamp(torch.round(c * 255.0) / _scalar(255.0, c), 0.0, 1.0)


def _psnr(peak_sq, mse):
    return math.inf if mse <= 0 else 10 * math.log10(peak_sq / mse)


class PointCloudMetric:
    """metrics/metric.py:6-189.  ``source`` / ``reconstruction``: [N, 6] tensors or arrays."""

    def __init__(self, source, reconstruction, resolution=1023, drop_duplicates=True, device="cuda:0"):
        self.report = {}
        self.resolution = resolution
        sc, srgb = _as_cloud(source, device)
        rc, rrgb = _as_cloud(reconstruction, device)
        if drop_duplicates:
            sc, srgb = _drop_duplicated_points(sc, srgb)
            rc, rrgb = _drop_duplicated_points(rc, rrgb)
        self.source_points, self.source_colors = sc, srgb
        self.recons_points, self.recons_colors = rc, rrgb
        self.source_map = CoordMap(sc, 1, nbatch=1)
        self.recons_map = CoordMap(rc, 1, nbatch=1)
        self.source_2_recons = nearest_neighbours(sc, self.recons_map, rrgb)
        self.recons_2_source = nearest_neighbours(rc, self.source_map, srgb)

    def get_result(self):
        return self.report

    def compute_pointcloud_metrics(self, drop_duplicates=False):
        """metric.py:60-83.  As in the reference, ``drop_duplicates=True`` SKIPS the averaging over
        equidistant neighbours; the default averages."""
        result, error_vectors = {}, {}
        ab, ev_ab = self.compute_metrics(mirror=False, drop_duplicates=drop_duplicates)
        ba, ev_ba = self.compute_metrics(mirror=True, drop_duplicates=drop_duplicates)
        result.update(ab)
        result.update(ba)
        error_vectors["colorAB"] = ev_ba        # the reference overwrites the AB vector with BA (metric.py:69-70)
        for name in ("mse", "hausdorff", "psnr_mse", "psnr_hausdorff", "y_mse", "u_mse", "v_mse", "y_psnr", "u_psnr", "v_psnr"):
            result["sym_" + name] = min(result["AB_" + name], result["BA_" + name])
        return result, error_vectors

    def compute_metrics(self, mirror=False, drop_duplicates=False):
        if not mirror:
            a_colors, b_colors, assoc, prefix = self.source_colors, self.recons_colors, self.source_2_recons, "AB_"
        else:
            a_colors, b_colors, assoc, prefix = self.recons_colors, self.source_colors, self.recons_2_source, "BA_"
        idx, d2, ties, tsum = assoc
        result = {}
        l2 = d2.to(torch.float64)
        l2 = l2 / _scalar(3.0, l2)                          # mean over the three axes (metric.py:113)
        mse = float(l2.mean())
        haus = float(l2.max())
        peak = float(self.resolution) ** 2
        result[prefix + "mse"] = mse
        result[prefix + "hausdorff"] = haus
        result[prefix + "psnr_mse"] = _psnr(peak, mse)
        result[prefix + "psnr_hausdorff"] = _psnr(peak, haus)
        b_ordered = b_colors.index_select(0, idx.long())
        if not drop_duplicates:
            # metric.py:121-146: where more than one neighbour sits at the nearest distance, the first one's
            # colour plus the colours of ALL of them (the first again included), divided by their number + 1
            many = ties > 1
            avg = (b_ordered + tsum.to(torch.float64)) / (ties.to(torch.float64) + 1.0).unsqueeze(1)
            b_ordered = torch.where(many.unsqueeze(1), avg, b_ordered)
        a_yuv = rgb_to_yuv(_round_to_8bit(a_colors))
        b_yuv = rgb_to_yuv(_round_to_8bit(b_ordered))
        err = (a_yuv - b_yuv) ** 2
        e = err.to(torch.float64).mean(dim=0).tolist()
        for i, ch in enumerate("yuv"):
            result[prefix + ch + "_mse"] = e[i]
            result[prefix + ch + "_psnr"] = _psnr(1.0, e[i])
        result[prefix + "yuv_mse"] = sum(e) / 3.0
        result[prefix + "yuv_psnr"] = _psnr(1.0, result[prefix + "yuv_mse"])
        return result, err


class Bjontegaard_Model:
    """metrics/bjontegaard.py:41-79: cubic least-squares fits PSNR(log10 R) and log10 R(PSNR)."""

    def __init__(self, bitrates, psnr_values):
        self.bitrates = np.asarray(bitrates, dtype=np.float64)
        self.psnr_values = np.asarray(psnr_values, dtype=np.float64)
        log_rate = np.log10(self.bitrates)
        self.parameters_PSNR = np.polyfit(log_rate, self.psnr_values, 3)
        self.parameters_Rate = np.polyfit(self.psnr_values, log_rate, 3)

    def evaluate(self, R):
        return np.polyval(self.parameters_PSNR, np.log10(R))

    def evaluate_rate(self, D):
        return np.polyval(self.parameters_Rate, D)

    def get_plot_data(self):
        xs = np.linspace(self.bitrates.min(), self.bitrates.max(), 100)
        return self.bitrates, self.psnr_values, xs, self.evaluate(xs)


class Bjontegaard_Delta:
    """metrics/bjontegaard.py:6-36: mean gap between two fitted curves over their common interval."""

    @staticmethod
    def _mean_gap(p1, p2, lo, hi):
        i1, i2 = np.polyint(p1), np.polyint(p2)
        return ((np.polyval(i2, hi) - np.polyval(i1, hi)) - (np.polyval(i2, lo) - np.polyval(i1, lo))) / (hi - lo)

    def compute_BD_PSNR(self, model1, model2):
        l1, l2 = np.log10(model1.bitrates), np.log10(model2.bitrates)
        return self._mean_gap(model1.parameters_PSNR, model2.parameters_PSNR, max(l1.min(), l2.min()), min(l1.max(), l2.max()))

    def compute_BD_Rate(self, model1, model2):
        d1, d2 = model1.psnr_values, model2.psnr_values
        expo = self._mean_gap(model1.parameters_Rate, model2.parameters_Rate, max(d1.min(), d2.min()), min(d1.max(), d2.max()))
        return 10 ** expo - 1
