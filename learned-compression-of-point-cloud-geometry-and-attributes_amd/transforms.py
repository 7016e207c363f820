"""Analysis (g_a) and generative-stride synthesis (g_s) transforms.

Module tree / parameter names follow /root/reference/model/transforms.py:8-304
(SURVEY.md Appendix A).  Forward passes are organised for the MI355X path:

* the q-map pyramid is re-indexed onto the main branch's coordinate maps whenever both hold the
  same coordinate set (always true for the reference's callers, utils.py:436-445), so beta/gamma
  need no lookup and every 3x3x3 kernel map is built once per coordinate set and shared;
* layers whose output the reference only reads at some coordinates (q_predict_1's last conv,
  q_up_i) are evaluated at exactly those coordinates — same values, less work.
"""
import torch
import torch.nn as nn

from . import sparse as sp
from .blocks import join_analysis_level, prefetch_analysis_maps, prefetch_up_maps, ConditionEncoder, GenerativeUpBlock, ScaledBlock, _conv
from .sparse import ConvChain, CoordMap, MinkowskiConvolution, MinkowskiReLU, SparseTensor


def _align(Q, target_map):
    """Return Q re-indexed onto ``target_map`` if both hold the same coordinate set, else None."""
    if Q.map is target_map:
        return Q
    if Q.map.n != target_map.n or Q.map.stride != target_map.stride:
        return None
    idx = target_map.lookup(Q.C)
    if not bool((idx >= 0).all()):      # one small host sync per frame
        return None
    return SparseTensor(sp.scatter_rows(Q.F, idx, target_map.n), coordinate_map=target_map)


class AnalysisTransform(nn.Module):
    """model/transforms.py:8-128."""

    def __init__(self, config):
        super().__init__()
        C_in, N1, N2, N3 = config["C_in"], config["N1"], config["N2"], config["N3"]
        self.condition_ablation = config.get("condition_ablation")
        if config["source_condition"]:
            self.cond_conv = ConvChain(_conv(C_in, 2), MinkowskiReLU(), _conv(2, 2))
        else:
            self.cond_conv = None
        self.pre_conv = ConvChain(_conv(C_in, N1), MinkowskiReLU())
        self.down_1 = _conv(N1, N2, 3, 2)
        self.down_2 = _conv(N2, N3, 3, 2)
        self.down_3 = _conv(N3, N3, 3, 2)
        self.scale_1 = ScaledBlock(N2, encode=True, scale=True)
        self.scale_2 = ScaledBlock(N3, encode=True, scale=True)
        self.scale_3 = ScaledBlock(N3, encode=True, scale=True)
        self.post_conv = _conv(N3, N3)
        self.condition_encoder = ConditionEncoder(C_in=2, N_scales=[N2, N2, N3], N_features=[2, 2, 2, 2],
                                                  condition_ablation=self.condition_ablation)

    def count_per_batch(self, x):
        return x.map.count_per_batch()

    def forward(self, x, Q):
        k = [self.count_per_batch(x)]
        aligned = _align(Q, x.map)
        if aligned is not None:
            Q = aligned
        if self.cond_conv:
            Q_plus = self.cond_conv(x)
            if Q.map is x.map:
                Q = SparseTensor(Q.F + Q_plus.F, coordinate_map=x.map)
            else:
                Q = SparseTensor(Q.F + Q_plus.features_at_coordinates(Q.C), coordinate_map=Q.map)
        x_in_map = x.map
        prefetched = Q.map is x_in_map
        if prefetched:
            prefetch_analysis_maps(x_in_map)       # a helper thread builds every coarser set and map on the side stream, level by level
        x = self.pre_conv(x)
        ce = self.condition_encoder
        Q = ce.begin(Q)
        # level by level: the q-map's pyramid stage, the stride-2 convolution, the ScaledBlock — the reference runs the whole pyramid
        # first (blocks.py:240-249); the values are the same, and a level's (heavy) convolutions start as soon as ITS maps are there
        downs, scales = (self.down_1, self.down_2, self.down_3), (self.scale_1, self.scale_2, self.scale_3)
        for i in range(3):
            if prefetched:
                join_analysis_level(x_in_map, i + 1)
            Q, beta_gamma = ce.stage(i, Q)
            x = downs[i](x)
            x = scales[i](x, beta_gamma)
            if i < 2:
                k.append(self.count_per_batch(x))
        x = self.post_conv(x)
        if prefetched:
            join_analysis_level(x_in_map, -1)      # h_a / h_s's sets and maps (the entropy model looks them up next)

        if Q.map is x.map:
            Q8 = Q
        else:
            Q8 = SparseTensor(Q.features_at_coordinates(x.C), coordinate_map=x.map)
        k.reverse()
        return x, Q8, k


class SparseSynthesisTransform(nn.Module):
    """model/transforms.py:134-304."""

    def __init__(self, config):
        super().__init__()
        C_out, N1, N2, N3 = config["C_out"], config["N1"], config["N2"], config["N3"]
        dense = config.get("dense", True)
        if config["source_condition"]:
            self.cond_conv = ConvChain(_conv(N1, N1 // 2), MinkowskiReLU(), _conv(N1 // 2, 2))
        else:
            self.cond_conv = None
        self.pre_conv = ConvChain(_conv(N1, N1), MinkowskiReLU())
        self.up_1 = GenerativeUpBlock(N1, N1, predict=True, dense=dense)
        self.up_2 = GenerativeUpBlock(N1, N2, predict=True, dense=dense)
        self.up_3 = GenerativeUpBlock(N2, N3, predict=True, dense=dense)
        self.scale_1 = ScaledBlock(N1, encode=False, scale=True)
        self.scale_2 = ScaledBlock(N1, encode=False, scale=True)
        self.scale_3 = ScaledBlock(N2, encode=False, scale=True)
        self.post_conv = ConvChain(_conv(N3, N3), MinkowskiReLU(), _conv(N3, N3 // 2), MinkowskiReLU(),
                                   _conv(N3 // 2, C_out))
        self.q_pre_conv = ConvChain(_conv(2, 16), MinkowskiReLU(), _conv(16, 16, 1), MinkowskiReLU(), _conv(16, 2))
        self.condition_ablation = config.get("condition_ablation")
        self.q_up_1 = GenerativeUpBlock(2, 2, condition_ablation=self.condition_ablation)
        self.q_up_2 = GenerativeUpBlock(2, 2, condition_ablation=self.condition_ablation)
        self.q_up_3 = GenerativeUpBlock(2, 2, condition_ablation=self.condition_ablation)
        self.q_predict_1 = ConvChain(_conv(2, N1), MinkowskiReLU(), _conv(N1, N1), MinkowskiReLU(), _conv(N1, N1 * 2))
        self.q_predict_2 = ConvChain(_conv(2, N1), MinkowskiReLU(), _conv(N1, N1), MinkowskiReLU(), _conv(N1, N1 * 2))
        self.q_predict_3 = ConvChain(_conv(2, N2), MinkowskiReLU(), _conv(N2, N2), MinkowskiReLU(), _conv(N2, N2 * 2))
        # coordinates-only helper (model/model.py:189-190, transforms.py:298-299); the 1->1 kernel is
        # a parameter of the reference model, its values never influence any output
        self.down_conv = MinkowskiConvolution(in_channels=1, out_channels=1, kernel_size=3, stride=2, dimension=3)

    def forward(self, x, Q, coords=None, k=None):
        # condition_ablation reaches only the q_up_i blocks here, whose predict=False branch never reads it (blocks.py:179-181)
        full_pred = coords is not None
        aligned = _align(Q, x.map)
        if aligned is not None:
            Q = aligned
        if self.cond_conv:
            Q_plus = self.cond_conv(x)
            if Q.map is x.map:
                Q = SparseTensor(Q.F + Q_plus.F, coordinate_map=Q.map)
            else:
                feats = sp.gather_rows(Q_plus.F, Q_plus.map.lookup(Q.C), out=Q.F.clone(), accumulate=True)
                Q = SparseTensor(feats, coordinate_map=Q.map)

        x = self.pre_conv(x)
        Q = self.q_pre_conv(Q)

        predictions = []
        ups = (self.up_1, self.up_2, self.up_3)
        q_ups = (self.q_up_1, self.q_up_2, self.q_up_3)
        q_predicts = (self.q_predict_1, self.q_predict_2, self.q_predict_3)
        scales = (self.scale_1, self.scale_2, self.scale_3)
        for i in range(3):
            # beta/gamma are only ever read at x's coordinates (blocks.py:37): evaluate the last conv of
            # q_predict there (identical values; it is the whole map when Q already lives on x's map)
            prefetch_up_maps(x.map)          # side stream: the up block's coordinate set and kernel maps
            beta_gamma = q_predicts[i](Q, last_out_map=x.map)
            x = scales[i](x, beta_gamma)
            x, pred, up_map = ups[i](x, k=k[i], full_predictions=full_pred)
            Q = q_ups[i](Q, up_map)
            predictions.append(pred)

        x = self.post_conv(x)

        if coords is not None:
            points_1 = coords.map.down()
            points_2 = points_1.down()
            points = [SparseTensor(coordinate_map=points_2), SparseTensor(coordinate_map=points_1), coords]
            return x, points, predictions
        return x
