"""Evaluation harness: one frame through file-mode compress / decompress with the reference's timing
bracket (mirror of ``compress_model_ours``, /root/reference/utils.py:418-472, as driven by
evaluate.py:55-216), plus the quality numbers the sweep records.

Differences in form only: clouds are ``[N, 6]`` GPU tensors instead of open3d objects, and the
metrics come from metrics.PointCloudMetric on the GPU instead of the external ``pc_error`` binary
(utils.py:206-290) — same quantities (D1 / Y / U / V PSNR, symmetric = worse direction).
"""
import os
import time

import numpy as np
import torch

from .metrics import PointCloudMetric
from .sparse import SparseTensor


def build_q_map(points, q_g, q_a, device):
    """utils.py:436-445: scalar q -> uniform map; per-point arrays -> that map.  Channels [q_g, q_a]."""
    n = points.shape[0]
    coords = torch.cat([torch.zeros((n, 1), device=device), points.to(device, dtype=torch.float32)], dim=1)
    if isinstance(q_a, (float, int, np.floating)):
        feats = torch.cat([torch.ones((n, 1), device=device) * float(q_g), torch.ones((n, 1), device=device) * float(q_a)], dim=1)
    else:
        feats = torch.cat([torch.as_tensor(q_g).reshape(n, 1), torch.as_tensor(q_a).reshape(n, 1)], dim=1).to(device, torch.float32)
    return SparseTensor(coordinates=coords, features=feats, device=device)


def compress_model_ours(experiment, model, data, q_a, q_g, device, base_path):
    """-> (source [N,6], reconstruction [N',6], bpp, t_compress, t_decompress); the bitstream goes
    through ``<base_path>/<experiment>/tmp/bitstream.bin`` like the reference's."""
    points = data["src"]["points"].to(device, dtype=torch.float)
    colors = data["src"]["colors"].to(device, dtype=torch.float)
    source = torch.cat([points, colors], dim=2)[0]
    n = source.shape[0]
    bin_dir = os.path.join(base_path, experiment, "tmp")
    os.makedirs(bin_dir, exist_ok=True)
    bin_path = os.path.join(bin_dir, "bitstream.bin")
    q_map = build_q_map(points[0], q_g, q_a, device)

    torch.cuda.synchronize()
    t0 = time.time()
    model.compress(source, q_map, path=bin_path)
    torch.cuda.synchronize()
    t_compress = time.time() - t0

    torch.cuda.synchronize()
    t0 = time.time()
    reconstruction = model.decompress(path=bin_path)
    torch.cuda.synchronize()
    t_decompress = time.time() - t0

    bpp = os.path.getsize(bin_path) * 8 / n
    return source, reconstruction, bpp, t_compress, t_decompress


def evaluate_frame(experiment, model, data, q_a, q_g, device, base_path, resolution=1023):
    """one row of the sweep table (evaluate.py:100-160): rate, times, D1 and colour PSNRs"""
    src, rec, bpp, t_c, t_d = compress_model_ours(experiment, model, data, q_a, q_g, device, base_path)
    res, _ = PointCloudMetric(src, rec, resolution=resolution, device=device).compute_pointcloud_metrics(drop_duplicates=True)
    return {"q_g": float(np.mean(q_g)), "q_a": float(np.mean(q_a)), "bpp": bpp, "t_compress": t_c, "t_decompress": t_d,
            "n_source": int(src.shape[0]), "n_decoded": int(rec.shape[0]),
            "sym_p2p_psnr": res["sym_psnr_mse"], "sym_y_psnr": res["sym_y_psnr"], "sym_u_psnr": res["sym_u_psnr"],
            "sym_v_psnr": res["sym_v_psnr"]}
