"""TEST INFRASTRUCTURE ONLY — CPU restatement of the training forward and the losses, differentiated by
torch autograd (the reference differentiates through MinkowskiEngine / compressai on CUDA:
/root/reference/train.py:171-221, model/model.py:51-93, loss.py:67-195).

``forward_train`` = oracle.codec.Codec.forward_eval with compressai's training-mode quantisation
(additive U(-0.5, 0.5) noise, entropy_models.py:313,330) and LowerBound gradients; the noise is
supplied by the caller so that the GPU path can be fed the same draws.
"""
import math

import numpy as np
import torch

from . import coords as oc
from .codec import analysis, h_a, h_q, h_s, synthesis
from .entropy import LIKELIHOOD_BOUND, SCALE_BOUND
from .nn import SparseTensor


class _LowerBound(torch.autograd.Function):
    """compressai.ops.LowerBound: max(x, b); the gradient also passes below the bound when it raises x"""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x)
        ctx.bound = bound
        return torch.clamp(x, min=bound)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ((x >= ctx.bound) | (g < 0)).to(g.dtype) * g, None


def leaf_state_dict(state_dict):
    """float tensors of a state_dict as autograd leaves"""
    return {k: torch.as_tensor(v).detach().to(torch.float32).cpu().clone().requires_grad_(True)
            for k, v in state_dict.items() if torch.as_tensor(v).dtype.is_floating_point}


def _hyper_train(eb, gc, p, y, noise, with_q=False):
    """one hyperprior in training mode (entropy_models.py:145-169 / :309-337): h_a, noisy z and its likelihood, h_s, noisy y and its
    likelihood -> y_hat, y_lik, z_lik, (Q_hat from h_q when ``with_q``)"""
    z = h_a(p.sub("h_a"), y)
    v = z.F.t().unsqueeze(0).permute(1, 0, 2)                        # [C, 1, N]
    v = v + noise(tuple(v.shape), z.C)
    z_lik, _, _ = eb._likelihood_raw(v)
    z_lik = _LowerBound.apply(z_lik, LIKELIHOOD_BOUND).permute(1, 0, 2)
    z_hat = SparseTensor(z.C, v.permute(1, 0, 2)[0].t().contiguous(), 32)
    params = h_s(p.sub("h_s"), z_hat)
    Q_hat = h_q(p.sub("h_q"), z_hat) if with_q else None
    scales, means = params.features_at_coordinates(y.C).chunk(2, dim=1)
    scales, means = scales.t().unsqueeze(0), means.t().unsqueeze(0)
    yin = y.F.t().unsqueeze(0)
    yout = yin + noise(tuple(yin.shape), y.C)
    a = torch.abs(yout - means)
    s = _LowerBound.apply(scales, SCALE_BOUND)
    Phi = gc._Phi
    y_lik = _LowerBound.apply(Phi((0.5 - a) / s) - Phi((-0.5 - a) / s), LIKELIHOOD_BOUND)
    return SparseTensor(y.C, yout[0].t().contiguous(), 8), y_lik, z_lik, Q_hat


def forward_train(codec, coords, colors, Q_coords, Q_feats, noise):
    """codec: oracle.codec.Codec built on leaf_state_dict(...) (``leaves=True``); noise(shape, coords) -> tensor, called per
    hyperprior for z ([C, 1, N32]) then y ([1, C, N8]) with the points' coordinates (row order is implementation-specific); the
    two-hyperprior variant (model/model.py:75-78) runs the latents' model first, then the q-map's."""
    coords = oc.to_int_coords(coords)
    N = coords.shape[0]
    feats = torch.cat([torch.ones((N, 1)), torch.as_tensor(colors, dtype=torch.float32)], dim=1)
    x = SparseTensor(coords, feats, 1)
    Q = SparseTensor(oc.to_int_coords(Q_coords), torch.as_tensor(Q_feats, dtype=torch.float32), 1)
    y, Q8, k = analysis(codec.p.sub("g_a"), x, Q, getattr(codec, "cfg_a", None))
    if getattr(codec, "two", False):
        y_hat, y_lik, zy_lik, _ = _hyper_train(codec.em_y.eb, codec.gc, codec.p.sub("entropy_model"), y, noise)
        Q_hat, q_lik, zq_lik, _ = _hyper_train(codec.em_q.eb, codec.gc, codec.p.sub("entropy_model_map"), Q8, noise)
        likelihoods = {"y": [y_lik, q_lik], "z": [zy_lik, zq_lik]}
    else:
        y_hat, y_lik, z_lik, Q_hat = _hyper_train(codec.eb, codec.gc, codec.p.sub("entropy_model"), y, noise, with_q=True)
        likelihoods = {"y": y_lik, "z": z_lik}
    x_hat, points, preds = synthesis(codec.p.sub("g_s"), y_hat, Q_hat, k, coords=coords, cfg=getattr(codec, "cfg_s", None))
    return {"prediction": x_hat, "points": points, "occ_predictions": preds, "likelihoods": likelihoods, "k": k}


def avg_pool(x, out_coords, out_stride):
    """ME.MinkowskiAvgPooling(kernel_size=3): mean over the inputs present in the 3^3 window (loss.py:154-155)"""
    nbr = oc.kernel_map(x.C, out_coords, 3, x.stride)
    valid = torch.from_numpy(nbr >= 0)
    idx = torch.from_numpy(np.where(nbr >= 0, nbr, 0)).long()
    sel = x.F[idx.reshape(-1)].reshape(nbr.shape[0], nbr.shape[1], -1) * valid.unsqueeze(2)
    cnt = valid.sum(dim=1, keepdim=True).clamp(min=1)
    return SparseTensor(out_coords, sel.sum(dim=1) / cnt, out_stride)


def losses(gt_coords, gt_colors, out, lambda_map, alpha=0.5, gamma=2.0):
    """configs/Ours.yaml:58-73: Multiscale_FocalLoss + ColorLoss(L2) + BPPLoss(y) + BPPLoss(z) -> (total, parts)"""
    gt_coords = oc.to_int_coords(gt_coords)
    n = gt_coords.shape[0]
    parts = {}
    for key in ("y", "z"):
        bits = 0.0
        for lik in out["likelihoods"][key]:
            bits = bits + torch.log(lik).sum() / (-math.log(2) * n)
        parts["bpp-" + key] = bits.mean()
    pred_colors = out["prediction"].features_at_coordinates(gt_coords)
    lam = lambda_map.features_at_coordinates(gt_coords)
    parts["ColorLoss"] = (((torch.as_tensor(gt_colors, dtype=torch.float32) - pred_colors) ** 2) * lam[:, 1].unsqueeze(1)).mean()
    focal_total = 0.0
    q = lambda_map
    for pred, pts in zip(reversed(out["occ_predictions"]), reversed(out["points"])):
        pts_c = pts if isinstance(pts, np.ndarray) else pts.C
        overlap = torch.from_numpy(np.isin(oc.pack(pred.C), oc.pack(pts_c)))
        p_z = torch.sigmoid(pred.F[:, 0])
        pt = torch.clip(torch.where(overlap, p_z, 1 - p_z), 1e-2, 1)
        a_z = torch.where(overlap, torch.tensor(alpha), torch.tensor(1 - alpha))
        focal = -a_z * (1 - pt) ** gamma * torch.log(pt)
        q_avg = avg_pool(q, pred.C, pred.stride)
        q = avg_pool(q, oc.stride_map(q.C, q.stride), q.stride * 2)
        focal_total = focal_total + (focal * q_avg.F[:, 0]).mean()
    parts["Multiscale_FocalLoss"] = focal_total
    return sum(parts.values()), parts
