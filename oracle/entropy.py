"""Oracle: factorized entropy bottleneck + Gaussian conditional (torch-CPU).

Restates compressai==1.2.4 ``EntropyBottleneck`` / ``GaussianConditional``
(requirements.txt:9; constructed at /root/reference/model/entropy_models.py:269-270,
used at :313,330,352-353,371-372,393,407-408) following SURVEY.md Appendix
B.2-B.4.  compressai is not under /root/reference and not installable here:
parity at this boundary is unpinned (see oracle/__init__.py).

Tensors follow the reference layout ``(1, C, N)`` (``z.F.t().unsqueeze(0)``).
Test infrastructure only.
"""
import math

import numpy as np
import scipy.stats
import torch

from . import rans as crans

SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256.0, 64
LIKELIHOOD_BOUND = 1e-9
SCALE_BOUND = 0.11
TAIL_MASS = 1e-9


def default_scale_table():
    return torch.exp(torch.linspace(math.log(SCALES_MIN), math.log(SCALES_MAX), SCALES_LEVELS))


class EntropyBottleneck:
    """B.2: filters (3,3,3,3), init_scale 10, tail_mass 1e-9."""

    def __init__(self, params):
        # params: nn.Params view with _matrix{i}, _bias{i}, _factor{i}, quantiles
        self.m = [params.get(f"_matrix{i}") for i in range(5)]
        self.b = [params.get(f"_bias{i}") for i in range(5)]
        self.f = [params.get(f"_factor{i}") for i in range(4)]
        self.quantiles = params.get("quantiles")
        self.C = self.quantiles.shape[0]
        self._cdf = None

    def medians(self):
        return self.quantiles[:, 0, 1]

    def logits_cumulative(self, v):
        # v: [C, 1, M]
        for i in range(5):
            v = torch.matmul(torch.nn.functional.softplus(self.m[i]), v) + self.b[i]
            if i < 4:
                v = v + torch.tanh(self.f[i]) * torch.tanh(v)
        return v

    def _likelihood_raw(self, v):
        lo = self.logits_cumulative(v - 0.5)
        up = self.logits_cumulative(v + 0.5)
        s = -torch.sign(lo + up)
        return torch.abs(torch.sigmoid(s * up) - torch.sigmoid(s * lo)), lo, up

    def forward_eval(self, x):
        """x [1,C,N] -> (x_hat [1,C,N], likelihood [1,C,N]) in eval mode."""
        med = self.medians().reshape(1, -1, 1)
        v = torch.round(x - med) + med
        L, _, _ = self._likelihood_raw(v[0].reshape(self.C, 1, -1))
        L = torch.clamp(L.reshape(1, self.C, -1), min=LIKELIHOOD_BOUND)
        return v, L

    def aux_loss(self):
        target = math.log(2 / TAIL_MASS - 1)
        t = torch.tensor([-target, 0.0, target])
        logits = self.logits_cumulative(self.quantiles)
        return torch.abs(logits - t).sum()

    def update(self):
        med = self.medians()
        minima = torch.clamp(torch.ceil(med - self.quantiles[:, 0, 0]).int(), min=0)
        maxima = torch.clamp(torch.ceil(self.quantiles[:, 0, 2] - med).int(), min=0)
        self.offset = (-minima).numpy().astype(np.int32)
        pmf_start = med - minima
        pmf_length = maxima + minima + 1
        max_length = int(pmf_length.max())
        samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
        pmf, lo, up = self._likelihood_raw(samples)
        pmf = pmf[:, 0, :]
        tail = torch.sigmoid(lo[:, 0, :1]) + torch.sigmoid(-up[:, 0, -1:])
        self.cdf = _pmf_to_cdf(pmf, tail, pmf_length, max_length)
        self.cdf_length = (pmf_length + 2).numpy().astype(np.int32)

    def symbols(self, x):
        med = self.medians().reshape(1, -1, 1)
        return torch.round(x - med).to(torch.int32)

    def compress(self, x):
        sym = self.symbols(x)[0]                      # [C, N]
        C, N = sym.shape
        idx = np.repeat(np.arange(C, dtype=np.int32), N)
        return [crans.encode_with_indexes(sym.reshape(-1).numpy(), idx, self.cdf, self.cdf_length, self.offset)]

    def decompress(self, strings, n):
        C = self.C
        idx = np.repeat(np.arange(C, dtype=np.int32), n)
        vals = crans.decode_with_indexes(strings[0], idx, self.cdf, self.cdf_length, self.offset)
        out = torch.from_numpy(vals.astype(np.float32)).reshape(1, C, n)
        return out + self.medians().reshape(1, -1, 1)


def _pmf_to_cdf(pmf, tail, pmf_length, max_length):
    cdf = np.zeros((len(pmf_length), max_length + 2), dtype=np.int32)
    for i in range(len(pmf_length)):
        L = int(pmf_length[i])
        prob = torch.cat((pmf[i, :L], tail[i].reshape(-1)), dim=0).to(torch.float32).numpy()
        c = crans.pmf_to_quantized_cdf(prob, 16)
        cdf[i, : c.size] = c
    return cdf


class GaussianConditional:
    """B.3: scale_bound 0.11, tail_mass 1e-9, default 64-level scale table."""

    def __init__(self, scale_table=None):
        self.scale_table = default_scale_table() if scale_table is None else torch.as_tensor(scale_table, dtype=torch.float32)

    @staticmethod
    def _Phi(x):
        return 0.5 * torch.erfc(-(2 ** -0.5) * x)

    def update(self):
        mult = -scipy.stats.norm.ppf(TAIL_MASS / 2)
        center = torch.ceil(self.scale_table * mult).int()
        pmf_length = 2 * center + 1
        max_length = int(pmf_length.max())
        samples = torch.abs(torch.arange(max_length).int() - center[:, None]).float()
        s = self.scale_table[:, None].float()
        upper = self._Phi((0.5 - samples) / s)
        lower = self._Phi((-0.5 - samples) / s)
        pmf = upper - lower
        tail = 2 * lower[:, :1]
        self.cdf = _pmf_to_cdf(pmf, tail, pmf_length, max_length)
        self.offset = (-center).numpy().astype(np.int32)
        self.cdf_length = (pmf_length + 2).numpy().astype(np.int32)

    def build_indexes(self, scales):
        s = torch.clamp(scales, min=SCALE_BOUND)
        idx = torch.full(s.shape, len(self.scale_table) - 1, dtype=torch.int32)
        for t in self.scale_table[:-1]:
            idx -= (s <= t).int()
        return idx

    def likelihood(self, v, scales, means):
        a = torch.abs(v - means)
        s = torch.clamp(scales, min=SCALE_BOUND)
        L = self._Phi((0.5 - a) / s) - self._Phi((-0.5 - a) / s)
        return torch.clamp(L, min=LIKELIHOOD_BOUND)

    def forward_eval(self, y, scales, means):
        v = torch.round(y - means) + means
        return v, self.likelihood(v, scales, means)

    def compress(self, y, indexes, means):
        sym = torch.round(y - means).to(torch.int32)
        return [crans.encode_with_indexes(sym.reshape(-1).numpy(), indexes.reshape(-1).numpy(),
                                          self.cdf, self.cdf_length, self.offset)]

    def decompress(self, strings, indexes, means):
        vals = crans.decode_with_indexes(strings[0], indexes.reshape(-1).numpy(), self.cdf, self.cdf_length, self.offset)
        return torch.from_numpy(vals.astype(np.float32)).reshape(indexes.shape) + means
