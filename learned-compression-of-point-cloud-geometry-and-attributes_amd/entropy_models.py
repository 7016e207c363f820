"""Mean-scale hyperpriors on sparse tensors.

Mirror of ``MeanScaleHyperprior_Map`` (/root/reference/model/entropy_models.py:253-414: h_a, h_s,
h_q — the q-map decoded from z_hat —, factorized bottleneck on z, Gaussian conditional on y, one rANS
stream each; every shipped config) and of ``MeanScaleHyperprior`` (:104-250: the same without h_q;
model/model.py:22-24 instantiates it twice when the config has an "entropy_model_map" section).

Differences in *how* (never in *what*):
* the reference's ``Sorted*`` shims (entropy_models.py:12-102) exist to make h_s reproducible
  between encoder and decoder; here every convolution output is a pure function of its own
  neighbourhood with a fixed accumulation order (csrc/conv.hip), so plain layers are used for h_s
  and no re-sorting is needed.  Parameter names are unchanged.
* tensors are never physically sorted: the canonical (b,x,y,z) order (utils.sort_tensor,
  utils.py:155-180) is applied as a column permutation of the symbol planes just before rANS.
* h_s's last convolution is evaluated only at y's coordinates — the only rows the reference
  reads (entropy_models.py:364,401) — which also makes (scales | means) row-aligned with y.
"""
import torch
import torch.nn as nn

from . import sparse as sp
from .entropy import EntropyBottleneck, GaussianConditional, get_scale_table
from .sparse import (ConvChain, CoordMap, MinkowskiConvolution, MinkowskiConvolutionTranspose,
                     MinkowskiGenerativeConvolutionTranspose, MinkowskiLeakyReLU, MinkowskiReLU, SparseTensor)


def _c(cin, cout, k=3, s=1, bias=False):
    return MinkowskiConvolution(in_channels=cin, out_channels=cout, kernel_size=k, stride=s, bias=bias, dimension=3)


class MeanScaleHyperprior(nn.Module):
    """``MeanScaleHyperprior`` (/root/reference/model/entropy_models.py:104-250): h_a, factorized z, h_s, Gaussian y — the entropy model
    of the two-hyperprior variant of ColorModel (model/model.py:22-24: one instance codes y, a second the stride-8 q-map).  The layers
    of MeanScaleHyperprior_Map without h_q; same parameter names as the reference, same remarks as in the module docstring."""

    def __init__(self, config):
        super().__init__()
        Cb, Ch = config["C_bottleneck"], config["C_hyper_bottleneck"]
        self.C_bottleneck = Cb
        self.entropy_bottleneck = EntropyBottleneck(Ch)
        self.gaussian_conditional = GaussianConditional(None)
        self.h_a = ConvChain(
            _c(Cb, Ch), MinkowskiLeakyReLU(),
            _c(Ch, Ch, 3, 2), _c(Ch, Ch), MinkowskiLeakyReLU(),
            _c(Ch, Ch, 3, 2), _c(Ch, Ch))
        gT = lambda cin, cout, k: MinkowskiGenerativeConvolutionTranspose(
            in_channels=cin, out_channels=cout, kernel_size=k, stride=2, bias=True, dimension=3)
        self.h_s = ConvChain(
            _c(Ch, Ch, bias=True), gT(Ch, Ch, 2), MinkowskiLeakyReLU(),
            _c(Ch, Ch, bias=True), gT(Ch, Cb * 3 // 2, 2), MinkowskiLeakyReLU(),
            _c(Cb * 3 // 2, Cb * 2, bias=True))

    def update(self, scale_table=None, force=False):
        """CompressionModel.update: EB tables + the 64-level Gaussian table (model/model.py:30-36)"""
        if scale_table is None:
            scale_table = get_scale_table()
        updated = self.gaussian_conditional.update_scale_table(scale_table, force=force)
        updated |= self.entropy_bottleneck.update(force=force)
        return updated

    def aux_loss(self):
        return self.entropy_bottleneck.loss()

    def _params_at(self, z_hat, y_map):
        return self.h_s(z_hat, last_out_map=y_map).F

    def forward(self, y):
        """entropy_models.py:145-169 -> y_hat, (L_y, L_z)"""
        z = self.h_a(y)
        if self.training:
            from . import entropy as _e
            _e.NOISE_ROWS = z.C
            z_hat_f, z_lik = self.entropy_bottleneck(z.F.t().unsqueeze(0))
            z_hat = SparseTensor(z_hat_f[0].t().contiguous(), coordinate_map=z.map)
            scales, means = self._params_at(z_hat, y.map).chunk(2, dim=1)
            _e.NOISE_ROWS = y.C
            y_hat_f, y_lik = self.gaussian_conditional(y.F.t().unsqueeze(0), scales.t().unsqueeze(0), means=means.t().unsqueeze(0))
            return SparseTensor(y_hat_f[0].t().contiguous(), coordinate_map=y.map), (y_lik, z_lik)
        z_hat_f, z_lik = self.entropy_bottleneck(z.F.t().unsqueeze(0))
        z_hat = SparseTensor(z_hat_f[0].t().contiguous(), coordinate_map=z.map)
        y_hat_f, y_lik = self.gaussian_conditional.forward_features(y.F, self._params_at(z_hat, y.map))
        return SparseTensor(y_hat_f, coordinate_map=y.map), (y_lik.unsqueeze(0), z_lik)

    def compress(self, y):
        """entropy_models.py:172-212 -> (points, strings, shape)"""
        z = self.h_a(y)
        perm_y = y.map.sort_permutation()
        perm_z = z.map.sort_permutation()
        shape = [z.map.n]
        finish_z, z_hat_f = self.entropy_bottleneck.compress_features_begin(z.F, perm=perm_z)
        z_hat = SparseTensor(z_hat_f, coordinate_map=z.map)
        finish_y = self.gaussian_conditional.compress_features_begin(y.F, self._params_at(z_hat, y.map), perm=perm_y)
        points = [y.C.index_select(0, perm_y.long()), z.C.index_select(0, perm_z.long())]
        z_strings = finish_z()
        y_strings = finish_y()
        return points, [y_strings, z_strings], shape

    def decompress(self, points, strings, shape):
        """entropy_models.py:215-250 -> y_hat on the canonically sorted stride-8 coordinates; ``points`` = [coords8, coords32]
        or CoordMaps (already canonically sorted ones are taken as they are: the two models of a codec share them)"""
        assert isinstance(strings, list) and len(strings) == 2
        y_sorted, z_sorted = (_canonical_map(m, s) for m, s in zip(points, (8, 32)))
        y_strings, z_strings = strings
        finish_z = self.entropy_bottleneck.decompress_features_async(z_strings, int(shape[0]), z_sorted.device)
        z_hat = SparseTensor(finish_z(), coordinate_map=z_sorted)
        finish_y = self.gaussian_conditional.decompress_features_async(y_strings, self._params_at(z_hat, y_sorted), self.C_bottleneck)
        return SparseTensor(finish_y(), coordinate_map=y_sorted)


def _canonical_map(m, stride):
    """coordinates or a CoordMap -> the CoordMap of the same set in canonical (b, x, y, z) order = bitstream order (utils.sort_points)"""
    if not isinstance(m, CoordMap):
        m = CoordMap(sp._as_int_coords(m), stride)
    if getattr(m, "_canonical", False):
        return m
    out = CoordMap(m.coords.index_select(0, m.sort_permutation().long()), stride, nbatch=m._nbatch)
    out._canonical = True
    return out


class MeanScaleHyperprior_Map(nn.Module):
    def __init__(self, config):
        super().__init__()
        Cb, Ch, Cq = config["C_bottleneck"], config["C_hyper_bottleneck"], config["C_Q"]
        self.C_bottleneck = Cb
        self.entropy_bottleneck = EntropyBottleneck(Ch)
        self.gaussian_conditional = GaussianConditional(None)
        self.h_a = ConvChain(
            _c(Cb, Ch), MinkowskiLeakyReLU(),
            _c(Ch, Ch, 3, 2), _c(Ch, Ch), MinkowskiLeakyReLU(),
            _c(Ch, Ch, 3, 2), _c(Ch, Ch))
        gT = lambda cin, cout, k: MinkowskiGenerativeConvolutionTranspose(
            in_channels=cin, out_channels=cout, kernel_size=k, stride=2, bias=True, dimension=3)
        cT = lambda cin, cout, k: MinkowskiConvolutionTranspose(
            in_channels=cin, out_channels=cout, kernel_size=k, stride=2, bias=True, dimension=3)
        self.h_s = ConvChain(
            _c(Ch, Ch, bias=True), gT(Ch, Ch, 2), MinkowskiLeakyReLU(),
            _c(Ch, Ch, bias=True), gT(Ch, Cb * 3 // 2, 2), MinkowskiLeakyReLU(),
            _c(Cb * 3 // 2, Cb * 2, bias=True))
        self.h_q = ConvChain(
            _c(Ch, Ch, bias=True), cT(Ch, Ch, 3), MinkowskiReLU(),
            _c(Ch, Ch, bias=True), cT(Ch, Ch, 3), MinkowskiReLU(),
            _c(Ch, Cq, bias=True))

    # compressai CompressionModel surface ------------------------------------------------------------
    def update(self, scale_table=None, force=False):
        """model/model.py:30-36 -> CompressionModel.update(force=True): EB tables + the default
        64-level Gaussian table."""
        if scale_table is None:
            scale_table = get_scale_table()
        updated = self.gaussian_conditional.update_scale_table(scale_table, force=force)
        updated |= self.entropy_bottleneck.update(force=force)
        return updated

    def aux_loss(self):
        return self.entropy_bottleneck.loss()

    # ------------------------------------------------------------------------------------------------
    def _params_at(self, z_hat, y_map):
        """h_s(z_hat) evaluated at y's coordinates -> [N_y, 2*C] = (scales | means)."""
        return self.h_s(z_hat, last_out_map=y_map).F

    def forward(self, y):
        """entropy_models.py:309-337: -> y_hat, Q_hat, (L_y, L_z).  Eval: rounding + HIP likelihood kernels;
        training: additive noise and differentiable likelihoods (compressai's "noise" mode)."""
        z = self.h_a(y)
        if self.training:
            from . import entropy as _e
            _e.NOISE_ROWS = z.C
            z_hat_f, z_lik = self.entropy_bottleneck(z.F.t().unsqueeze(0))
            z_hat = SparseTensor(z_hat_f[0].t().contiguous(), coordinate_map=z.map)
            params = self._params_at(z_hat, y.map)
            Q_hat = self.h_q(z_hat)
            scales, means = params.chunk(2, dim=1)
            _e.NOISE_ROWS = y.C
            y_hat_f, y_lik = self.gaussian_conditional(y.F.t().unsqueeze(0), scales.t().unsqueeze(0), means=means.t().unsqueeze(0))
            return SparseTensor(y_hat_f[0].t().contiguous(), coordinate_map=y.map), Q_hat, (y_lik, z_lik)
        z_in = z.F.t().unsqueeze(0)
        z_hat_f, z_lik = self.entropy_bottleneck(z_in)
        z_hat = SparseTensor(z_hat_f[0].t().contiguous(), coordinate_map=z.map)
        params = self._params_at(z_hat, y.map)
        Q_hat = self.h_q(z_hat)
        y_hat_f, y_lik = self.gaussian_conditional.forward_features(y.F, params)
        y_hat = SparseTensor(y_hat_f, coordinate_map=y.map)
        return y_hat, Q_hat, (y_lik.unsqueeze(0), z_lik)

    def compress(self, y):
        """entropy_models.py:341-381 -> (points, strings, shape)."""
        z = self.h_a(y)
        perm_y = y.map.sort_permutation()
        perm_z = z.map.sort_permutation()
        shape = [z.map.n]
        # the host range coder of z runs while the GPU computes h_s and prepares the y symbols: everything the
        # GPU still has to do is enqueued before the first host wait
        finish_z, z_hat_f = self.entropy_bottleneck.compress_features_begin(z.F, perm=perm_z)
        z_hat = SparseTensor(z_hat_f, coordinate_map=z.map)
        params = self._params_at(z_hat, y.map)
        finish_y = self.gaussian_conditional.compress_features_begin(y.F, params, perm=perm_y)
        # (the sorted point lists are enqueued before the host starts coding: behind it they were four small launches at the very end)
        points = [y.C.index_select(0, perm_y.long()), z.C.index_select(0, perm_z.long())]
        z_strings = finish_z()
        y_strings = finish_y()
        return points, [y_strings, z_strings], shape

    def decompress(self, points, strings, shape):
        """entropy_models.py:384-414 -> (y_hat, Q_hat); ``points`` = [coords8, coords32] or CoordMaps."""
        assert isinstance(strings, list) and len(strings) == 2
        y_map, z_map = points
        if not isinstance(y_map, CoordMap):
            y_map = CoordMap(sp._as_int_coords(y_map), 8)
        if not isinstance(z_map, CoordMap):
            z_map = CoordMap(sp._as_int_coords(z_map), 32)
        # canonical order = bitstream order (utils.sort_points)
        y_sorted = CoordMap(y_map.coords.index_select(0, y_map.sort_permutation().long()), 8, nbatch=y_map._nbatch)
        z_sorted = CoordMap(z_map.coords.index_select(0, z_map.sort_permutation().long()), 32, nbatch=z_map._nbatch)
        y_strings, z_strings = strings
        # the z stream needs nothing from the GPU: its host decode runs while the GPU builds what depends on coordinates
        # only — the tables and kernel maps of h_s (z -> 16 -> 8, evaluated at y) and of the first h_q layer
        early = self.__dict__.pop("_early_z", None)                   # ColorModel.decompress starts it before the coordinate sets
        if early is not None and early[0] is strings:                 # (a decode started for THESE strings: an earlier call may have failed)
            finish_z = early[1]
        else:
            finish_z = self.entropy_bottleneck.decompress_features_async(z_strings, int(shape[0]), z_sorted.device)
        z_sorted.mfma_kernel_map(z_sorted, 3)
        z16 = z_sorted.up(2)
        z_sorted.mfma_kernel_map(z16, 2, True)
        z16.mfma_kernel_map(z16, 3)
        z16.mfma_kernel_map(z16.up(2), 2, True)
        z16.up(2).mfma_kernel_map(y_sorted, 3)
        z_hat = SparseTensor(finish_z(), coordinate_map=z_sorted)
        # h_s first: its output (scales | means) is what the serial rANS decode of y is waiting for.  While
        # the host decodes (~15 ms for 2.5 M symbols) the GPU runs everything that does not need y: h_q
        # and the coordinate sets / kernel maps of the first synthesis stage.
        params = self._params_at(z_hat, y_sorted)
        finish_y = self.gaussian_conditional.decompress_features_async(y_strings, params, self.C_bottleneck)
        Q_hat = self.h_q(z_hat)
        self._prefetch_synthesis_maps(y_sorted, Q_hat.map)
        y_hat_f = finish_y()
        return SparseTensor(y_hat_f, coordinate_map=y_sorted), Q_hat

    def start_z_decode(self, strings, shape, device):
        """The z stream needs nothing but its bytes: its host decode (0.6-0.8 ms on the config-2 frame) can start before the first
        coordinate set of a decompress is built; decompress() picks the pending result up."""
        self.__dict__["_early_z"] = (strings, self.entropy_bottleneck.decompress_features_async(strings[1], int(shape[0]), device))

    @staticmethod
    def _prefetch_synthesis_maps(y_map, q_map):
        """Fill the kernel-map caches g_s will hit first (pure functions of the coordinates)."""
        y_map.mfma_kernel_map(y_map, 3)                    # cond_conv, pre_conv, scale_1 on the y set
        q_map.mfma_kernel_map(q_map, 3)                    # q_predict_1 on the dilated q-map support
        q_map.kernel_map(q_map, 3)                            # q_pre_conv (thin convolutions)
        q_map.mfma_kernel_map(y_map, 3)                    # q_predict_1's last conv evaluated at y
        cand = y_map.up(3)                                    # up_1 candidates
        y_map.mfma_kernel_map(cand, 3, True)
        cand.mfma_kernel_map(cand, 3)
        cand.kernel_map(cand, 3)
