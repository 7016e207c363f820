"""Training losses (mirror of /root/reference/loss.py:7-195: ``Loss``, ``BPPLoss``, ``ColorLoss``,
``FocalLoss``, ``Multiscale_FocalLoss``), on this package's SparseTensor.

Elementwise arithmetic on per-point vectors is plain torch (autograd); the sparse pieces run on the
HIP coordinate kernels: the reference's ``torch.isin`` on packed coordinates becomes a voxel-hash
lookup, ``MinkowskiAvgPooling`` (loss.py:154-155) an average over the kernel map's existing neighbours.
"""
import math

import torch
import torch.nn.functional as F

from .sparse import SparseTensor


def avg_pool(x, out_map, kernel_size=3):
    """ME.MinkowskiAvgPooling(kernel_size, stride = out stride / in stride): mean of the inputs that exist
    in the kernel window of every output voxel (zero where there is none)."""
    nbr, _, _ = x.map.kernel_map(out_map, kernel_size)
    valid = nbr >= 0
    sel = x.F.index_select(0, nbr.clamp(min=0).reshape(-1).long()).reshape(nbr.shape[0], nbr.shape[1], -1)
    sel = sel * valid.unsqueeze(2).to(sel.dtype)
    cnt = valid.sum(dim=1, keepdim=True).clamp(min=1).to(sel.dtype)
    return SparseTensor(sel.sum(dim=1) / cnt, coordinate_map=out_map)


class BPPLoss:
    def __init__(self, config):
        self.weight = config["weight"]
        self.identifier = config["id"]
        self.key = config["key"]

    def __call__(self, gt, pred):
        loss = 0.0
        num_points = gt.C.shape[0]
        for likelihood in pred["likelihoods"][self.key]:
            loss = loss + torch.log(likelihood).sum() / (-math.log(2) * num_points)
        return loss.mean() * self.weight


class ColorLoss:
    def __init__(self, config):
        self.identifier = config["id"]
        self.loss_func = torch.nn.L1Loss(reduction="none") if config["loss"] == "L1" else torch.nn.MSELoss(reduction="none")

    def __call__(self, gt, pred):
        pred_colors = pred["prediction"].features_at_coordinates(gt.C.float())
        color_loss = self.loss_func(gt.F, pred_colors)
        color_loss = color_loss * pred["q_map"].features_at_coordinates(gt.C.float())[:, 1].unsqueeze(1)
        return color_loss.mean()


def _focal(logits, overlapping, alpha, gamma):
    p_z = torch.sigmoid(logits)
    pt_z = torch.where(overlapping, p_z, 1 - p_z)
    alpha_z = torch.where(overlapping, torch.full_like(p_z, alpha), torch.full_like(p_z, 1 - alpha))
    pt_z = torch.clip(pt_z, 1e-2, 1)
    return -alpha_z * (1 - pt_z) ** gamma * torch.log(pt_z)


class FocalLoss:
    def __init__(self, config):
        self.identifier = config["id"]
        self.alpha, self.gamma = config["alpha"], config["gamma"]

    def __call__(self, gt, pred):
        prediction = pred["prediction"]
        overlapping = gt.map.lookup(prediction.C) >= 0
        return _focal(prediction.F[:, 0] + 0.5, overlapping, self.alpha, self.gamma).mean() * pred["lambdas"][0][0]


class Multiscale_FocalLoss:
    def __init__(self, config):
        self.identifier = config["id"]
        self.alpha, self.gamma = config["alpha"], config["gamma"]

    def __call__(self, gt, pred):
        predictions = list(reversed(pred["occ_predictions"]))      # finest scale first (loss.py:160-161)
        points = list(reversed(pred["points"]))
        q_map = pred["q_map"]
        loss = 0.0
        for prediction, coords in zip(predictions, points):
            overlapping = coords.map.lookup(prediction.C) >= 0
            focal = _focal(prediction.F[:, 0], overlapping, self.alpha, self.gamma)
            q_avgs = avg_pool(q_map, prediction.map, 3)             # pooled onto the candidates
            q_map = avg_pool(q_map, q_map.map.down(), 3)            # next scale
            loss = loss + (focal * q_avgs.F[:, 0]).mean()
        return loss


class Loss:
    """loss.py:7-64: sum of the configured losses -> (total, {id: value})."""
    TYPES = {"BPPLoss": BPPLoss, "ColorLoss": ColorLoss, "FocalLoss": FocalLoss, "Multiscale_FocalLoss": Multiscale_FocalLoss}

    def __init__(self, config):
        self.losses = {}
        for ident, setting in config.items():
            setting = dict(setting, id=ident)
            cls = self.TYPES.get(setting["type"])
            if cls is None:
                print("Not found {}".format(setting["type"]))
                continue
            self.losses[ident] = cls(setting)

    def __call__(self, gt, pred):
        total, parts = 0, {}
        for loss in self.losses.values():
            item = loss(gt, pred)
            parts[loss.identifier] = item
            total = total + item
        return total, parts


OURS_LOSS = {            # configs/Ours.yaml:58-73
    "Multiscale_FocalLoss": {"type": "Multiscale_FocalLoss", "alpha": 0.5, "gamma": 2.0},
    "ColorLoss": {"type": "ColorLoss", "loss": "L2"},
    "bpp-y": {"type": "BPPLoss", "key": "y", "weight": 1.0},
    "bpp-z": {"type": "BPPLoss", "key": "z", "weight": 1.0},
}
