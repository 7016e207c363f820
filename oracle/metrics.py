"""Oracle: D1 (point-to-point) PSNR and Y/U/V PSNR, CPU (numpy + scipy cKDTree).

Restates /root/reference/metrics/metric.py:61-189 (``PointCloudMetric`` with
``drop_duplicates=True`` as called at train.py:263-264): nearest-neighbour
association in both directions, geometry MSE = mean over points of the mean
squared coordinate difference (metric.py:113-119), colours rounded to 8 bit and
converted with the BT.709 matrix after a truncating uint8 cast
(metric.py:149-150,171-189), symmetric value = min over the two directions
(metric.py:72-83).  open3d's KD-tree is replaced by scipy's; ties between
equidistant neighbours (KD-tree visiting order in the reference: unpinned) resolve to the smallest (x, y, z).
Test infrastructure only — see oracle/__init__.py.
"""
import numpy as np
from scipy.spatial import cKDTree


def rgb_to_yuv(rgb):
    rgb = np.asarray(rgb, dtype=np.float64)
    scale = rgb.max() <= 1.0
    if scale:
        rgb = (rgb * 255).astype(np.uint8)
    yuv = np.empty(rgb.shape, dtype=np.float32)
    yuv[..., 0] = 0.2126 * rgb[..., 0] + 0.7152 * rgb[..., 1] + 0.0722 * rgb[..., 2]
    yuv[..., 1] = -0.1146 * rgb[..., 0] - 0.3854 * rgb[..., 1] + 0.5 * rgb[..., 2]
    yuv[..., 2] = 0.5 * rgb[..., 0] - 0.4542 * rgb[..., 1] - 0.0458 * rgb[..., 2]
    if scale:
        yuv = yuv / 255.0
        yuv[..., 1] += 0.5
        yuv[..., 2] += 0.5
    return yuv


def _dedupe(pc):
    pts = pc[:, :3]
    if pts.size and np.all(pts == np.floor(pts)) and np.abs(pts).max() < (1 << 20):      # integer grids: packed keys (np.unique over rows is 10x slower)
        c = pts.astype(np.int64) + (1 << 20)
        _, first = np.unique((c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2], return_index=True)
    else:
        _, first = np.unique(pts, axis=0, return_index=True)
    return pc[np.sort(first)]


def _one_way(a, b, resolution, average_ties=False, kmax=32):
    """nearest neighbour of every row of a in b.  Equidistant candidates (common on a lattice; the
    reference takes whichever its KD-tree visits first, metric.py:36-43) resolve to the smallest
    (x, y, z).  average_ties restates metric.py:121-146: where several neighbours share the nearest
    distance the colour becomes (first + sum of all of them) / (count + 1)."""
    tree = cKDTree(b[:, :3])
    k = min(kmax, b.shape[0])
    bkey = (b[:, 0].astype(np.int64) << 42) + (b[:, 1].astype(np.int64) << 21) + b[:, 2].astype(np.int64)

    def resolve(q, kk):
        dist, cand = tree.query(q, k=kk, workers=-1)              # all host cores: the same neighbours, found in parallel
        if kk == 1:
            dist, cand = dist[:, None], cand[:, None]
        d2c = np.rint(dist ** 2).astype(np.int64)                  # integer grids: exact squared distances
        tie = d2c == d2c[:, :1]
        ck = np.where(tie, bkey[cand], np.iinfo(np.int64).max)
        return cand, tie, cand[np.arange(q.shape[0]), ck.argmin(axis=1)]

    if average_ties or k <= 2:
        cand, tie, nn = resolve(a[:, :3], k)
    else:
        # two neighbours first: a point whose nearest neighbour is unique (nearly every point of a codec's output) is settled;
        # only the points whose two nearest are equidistant need the whole tie set (up to kmax candidates) — the same choice
        cand2, tie2, nn = resolve(a[:, :3], 2)
        again = np.nonzero(tie2[:, 1])[0]
        if again.size:
            _, _, nn_again = resolve(a[again, :3], k)
            nn[again] = nn_again
        cand = tie = None
    d = ((a[:, :3] - b[nn, :3]) ** 2).mean(axis=1)
    res = {"mse": d.mean(), "hausdorff": d.max()}
    res["psnr_mse"] = 10 * np.log10(resolution ** 2 / res["mse"]) if res["mse"] > 0 else np.inf
    res["psnr_hausdorff"] = 10 * np.log10(resolution ** 2 / res["hausdorff"]) if res["hausdorff"] > 0 else np.inf
    b_col = b[nn, 3:6].copy()
    if average_ties:
        cnt = tie.sum(axis=1)
        assert (cnt < k).all() or b.shape[0] <= kmax, "tie set may be truncated: raise kmax"
        tsum = (b[cand, 3:6] * tie[:, :, None]).sum(axis=1)
        many = cnt > 1
        b_col[many] = (b_col[many] + tsum[many]) / (cnt[many] + 1)[:, None]
    a_yuv = rgb_to_yuv(np.clip(np.round(a[:, 3:6] * 255.0) / 255.0, 0.0, 1.0))
    b_yuv = rgb_to_yuv(np.clip(np.round(b_col * 255.0) / 255.0, 0.0, 1.0))
    e = ((a_yuv - b_yuv) ** 2).mean(axis=0)
    for i, ch in enumerate("yuv"):
        res[f"{ch}_mse"] = e[i]
        res[f"{ch}_psnr"] = 10 * np.log10(1 / e[i]) if e[i] > 0 else np.inf
    return res


def pc_metrics(source, recon, resolution=1023, average_ties=False):
    """source/recon: float [N,6] (xyz voxel coords, rgb in [0,1]).  average_ties=False is the
    reference's compute_pointcloud_metrics(drop_duplicates=True), True its default."""
    a = _dedupe(np.asarray(source, dtype=np.float64))
    b = _dedupe(np.asarray(recon, dtype=np.float64))
    ab = _one_way(a, b, resolution, average_ties)
    ba = _one_way(b, a, resolution, average_ties)
    out = {}
    for kk in ab:
        out["AB_" + kk] = ab[kk]
        out["BA_" + kk] = ba[kk]
        out["sym_" + kk] = min(ab[kk], ba[kk])
    return out
