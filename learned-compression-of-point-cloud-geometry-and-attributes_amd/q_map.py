"""Quality-map builders (mirror of /root/reference/data/q_map.py:143-291, class ``Q_Map``).

A quality map is a sparse tensor on the frame's coordinates with two channels ``[q_g, q_a]`` in
[0, 1] (geometry, attribute; order fixed by utils.py:439).  ``Q_Map(config)(geometry)`` draws, per
batch item, either a linear gradient along a random axis or a uniform map with random levels — the
reference's training-time generator, driven by Python's ``random`` like the reference so that a
seeded run draws the same maps — and returns it together with the lambda map used by the losses.
``uniform_map`` / ``gradient_map`` / ``view_dependent_map`` / ``roi_map`` build the fixed maps of the
evaluation scripts (utils.py:436-445, evaluate_view_dep.py:207-260).  Elementwise work on [N, 2] tensors; plain torch.
"""
import math
import random

import torch

from .sparse import SparseTensor


def uniform_map(coords_map, q_g, q_a):
    n = coords_map.n
    f = torch.empty((n, 2), dtype=torch.float32, device=coords_map.device)
    f[:, 0] = float(q_g)
    f[:, 1] = float(q_a)
    return SparseTensor(f, coordinate_map=coords_map)


def gradient_map(coords_map, axis, lo=0.0, hi=1.0):
    """q rises linearly from ``lo`` to ``hi`` along coordinate axis 1..3 (both channels)."""
    c = coords_map.coords[:, axis].to(torch.float32)
    t = torch.clamp((c - c.min()) / (c.max() - c.min() + 1e-10), 0, 1)
    q = lo + (hi - lo) * t
    return SparseTensor(q.unsqueeze(1).repeat(1, 2).contiguous(), coordinate_map=coords_map)


def view_dependent_map(coords_map, q_g, q_a, axis, lo, hi):
    """evaluate_view_dep.py:207-215: quality falls off along a viewing axis — score = clip((p[axis] - lo) /
    (hi - lo), 0, 1), channels [q_g * score, q_a * score]"""
    c = coords_map.coords[:, axis].to(torch.float32)
    score = torch.clamp((c - float(lo)) / (float(hi) - float(lo)), 0, 1)
    return SparseTensor(torch.stack([float(q_g) * score, float(q_a) * score], dim=1).contiguous(), coordinate_map=coords_map)


def roi_map(coords_map, q_g, q_a, axis, plane):
    """evaluate_view_dep.py:254-260: region of interest — full quality where p[axis] >= plane, zero below"""
    score = (coords_map.coords[:, axis] >= plane).to(torch.float32)
    return SparseTensor(torch.stack([float(q_g) * score, float(q_a) * score], dim=1).contiguous(), coordinate_map=coords_map)


class Q_Map:
    def __init__(self, config):
        self.mode = config["mode"]
        if self.mode == "exponential":
            self.a_A = math.log2(config["lambda_A_max"] + config["lambda_A_min"])
            self.b_A = config["lambda_A_min"] - 1
            self.a_G = math.log2(config["lambda_G_max"] + config["lambda_G_min"])
            self.b_G = config["lambda_G_min"] - 1
        elif self.mode == "quadratic":
            self.a_A = config["lambda_A_max"] - config["lambda_A_min"]
            self.b_A = config["lambda_A_min"]
            self.a_G = config["lambda_G_max"] - config["lambda_G_min"]
            self.b_G = config["lambda_G_min"]
        else:
            raise ValueError("Unknown Q_map mode")

    def __call__(self, geometry):
        """geometry: SparseTensor -> (q_map, lambda_map), both on geometry's coordinate map"""
        coords = geometry.C
        feats = torch.zeros((coords.shape[0], 2), dtype=torch.float32, device=coords.device)
        for b in torch.unique(coords[:, 0]).tolist():
            mask = coords[:, 0] == b
            feats[mask] = self.random_q_map(coords[mask])
        q_map = SparseTensor(feats, coordinate_map=geometry.map)
        return q_map, self.scale_q_map(q_map)

    def scale_q_map(self, q_map):
        f = q_map.F.clone()
        if self.mode == "exponential":
            f[:, 0] = 2 ** (f[:, 0] * self.a_G) + self.b_G
            f[:, 1] = 2 ** (f[:, 1] * self.a_A) + self.b_A
        else:
            f[:, 0] = f[:, 0] ** 2 * self.a_G + self.b_G
            f[:, 1] = f[:, 1] ** 2 * self.a_A + self.b_A
        return SparseTensor(f, coordinate_map=q_map.map)

    def random_q_map(self, coordinates):
        return self.gradient(coordinates) if random.choice(range(2)) == 0 else self.uniform(coordinates)

    @staticmethod
    def gradient(coordinates):
        direction = random.randint(1, 3)
        c = coordinates[:, direction].to(torch.float32)
        q = torch.clamp((c - c.min()) / (c.max() - c.min() + 1e-10), 0, 1)
        return q.unsqueeze(1).repeat(1, 2)

    @staticmethod
    def uniform(coordinates):
        scale_geometry = random.uniform(0, 1)
        scale_attribute = random.uniform(0, 1)
        q = torch.ones((coordinates.shape[0], 2), dtype=torch.float32, device=coordinates.device)
        q[:, 0] *= scale_geometry
        q[:, 1] *= scale_attribute
        return q
