"""Oracle: the codec itself — g_a, hyperprior entropy model, g_s, container.

CPU restatement of /root/reference/model/model.py:51-208 (+ container :214-315),
model/transforms.py:8-304, model/blocks.py:10-53,78-251 and
model/entropy_models.py:253-414 (and :104-250, the two-hyperprior variant) for ``configs/Ours.yaml`` style configs,
driven by a flat ``state_dict`` whose keys follow the reference module tree
(SURVEY.md Appendix A).  Test infrastructure only — see oracle/__init__.py.
"""
import struct

import numpy as np
import torch

from . import coords as oc
from . import nn as on
from .entropy import EntropyBottleneck, GaussianConditional
from .nn import Params, SparseTensor


# ----------------------------------------------------------------------------------------
# blocks.py
# ----------------------------------------------------------------------------------------
def scaled_block(p, x, condition):
    """blocks.py:29-53 (ScaledBlock.forward; self.gdn is never called)."""
    x_res = x
    h = p.sub("conv_1").conv(x, "0")
    h = on.relu(h)
    h = p.sub("conv_1").conv(h, "2")
    beta, gamma = condition.features_at_coordinates(h.C).chunk(2, dim=1)       # [n, N] each, or [n, 1] (condition_ablation): broadcast
    h2 = SparseTensor(h.C, h.F * beta + gamma, h.stride)
    h2._cache = h._cache
    h = p.sub("conv_2").conv(h2, "0")
    h = on.relu(h)
    h = p.sub("conv_2").conv(h, "2")
    h = on.relu(h)
    out = SparseTensor(h.C, h.F + x_res.features_at_coordinates(h.C), h.stride)
    out._cache = h._cache
    return out


def topk_mask(pred, k):
    """blocks.py:130-150 — per-batch top-k on channel 0.

    torch.topk leaves ties unspecified; the oracle (and the HIP path) break
    exact ties by ascending canonical coordinate key.
    """
    logits = pred.F[:, 0].detach().numpy()
    keys = oc.pack(pred.C)
    b = pred.C[:, 0]
    mask = np.zeros(logits.shape[0], dtype=bool)
    for bi, batch in enumerate(np.unique(b)):
        rows = np.nonzero(b == batch)[0]
        kk = int(k[int(batch)]) if len(k) > int(batch) else int(k[bi])
        kk = min(kk, rows.size)
        order = np.lexsort((keys[rows], -logits[rows].astype(np.float64)))
        mask[rows[order[:kk]]] = True
    return mask


def prune_by_coords(x, keep_coords):
    """blocks.py:101-128 — isin on packed keys, order-preserving."""
    mask = np.isin(oc.pack(x.C), oc.pack(keep_coords))
    return on.prune(x, mask)


def up_block_predict(p, x, k, dense=True, condition_ablation=None, full_predictions=False):
    """blocks.py:152-177, predict=True; dense=False (:168-175) predicts on the raw candidates and refines the kept rows.
    ``full_predictions``: the caller returns the whole occupancy tensor (model.py:88, the training / eval forward) — the product then
    evaluates all its columns as one wide layer; otherwise (compress / decompress) only column 0 is ever read (blocks.py:142)."""
    def conv_2(t):
        h = on.relu(p.sub("conv_2").conv(t, "0"))
        return p.sub("conv_2").conv(h, "2")
    x = p.convT(x, "conv", 3)
    if dense and condition_ablation is None:
        x = conv_2(x)
    h = p.sub("occ_predict").conv(x, "0")
    h = on.relu(h)
    # only channel 0 is ever read (blocks.py:142); in "kernel" order it is evaluated the way the product evaluates it: alone
    # (a narrow head) in compress / decompress, as a column of the wide layer where the forward pass returns the whole tensor
    pred = p.sub("occ_predict").conv(h, "2", out_channels=1 if (on.ORDER == "kernel" and not full_predictions) else None)
    mask = topk_mask(pred, k)
    up_coords = pred.C[mask]
    x = prune_by_coords(x, up_coords)
    if not dense and condition_ablation is None:
        x = conv_2(x)
    return x, pred, up_coords


def up_block_follow(p, Q, up_coords):
    """blocks.py:179-181, predict=False (q_up_i): genConvT then prune."""
    Q = p.convT(Q, "conv", 3)
    return prune_by_coords(Q, up_coords)


def condition_encoder(p, Q, condition_ablation=None):
    """blocks.py:235-251 (conv_layers skipped: blocks.py:241; condition_ablation: the q-map itself is beta | gamma, :246-247)."""
    Q = on.relu(p.sub("pre_conv").conv(Q, "0"))
    bgs = []
    for i in range(3):
        Q = p.sub("down_layers").conv(Q, str(i), 3, 2)
        if condition_ablation is not None:
            bgs.append(Q)
            continue
        pl = p.sub("predict_layers").sub(str(i))
        h = on.relu(pl.conv(Q, "0"))
        h = on.relu(pl.conv(h, "2", ksize=1))
        bgs.append(pl.conv(h, "4"))
    return Q, bgs


# ----------------------------------------------------------------------------------------
# transforms.py
# ----------------------------------------------------------------------------------------
def analysis(p, x, Q, cfg=None):
    """transforms.py:75-128.  ``cfg``: the g_a section of the model config (source_condition, condition_ablation);
    None = configs/Ours.yaml."""
    cfg = cfg or {}
    k = [oc.count_per_batch(x.C)]
    if cfg.get("source_condition", True):
        h = on.relu(p.sub("cond_conv").conv(x, "0"))
        Q_plus = p.sub("cond_conv").conv(h, "2")
        Q = SparseTensor(Q.C, Q.F + Q_plus.features_at_coordinates(Q.C), 1)
    Q, bgs = condition_encoder(p.sub("condition_encoder"), Q, cfg.get("condition_ablation"))
    x = on.relu(p.sub("pre_conv").conv(x, "0"))
    x = p.conv(x, "down_1", 3, 2)
    x = scaled_block(p.sub("scale_1"), x, bgs[0])
    k.append(oc.count_per_batch(x.C))
    x = p.conv(x, "down_2", 3, 2)
    x = scaled_block(p.sub("scale_2"), x, bgs[1])
    k.append(oc.count_per_batch(x.C))
    x = p.conv(x, "down_3", 3, 2)
    x = scaled_block(p.sub("scale_3"), x, bgs[2])
    x = p.conv(x, "post_conv")
    Q8 = SparseTensor(x.C, Q.features_at_coordinates(x.C), x.stride)
    k.reverse()
    return x, Q8, k


def synthesis(p, x, Q, k, coords=None, cfg=None):
    """transforms.py:242-304.  ``cfg``: the g_s section of the model config (source_condition, dense); None = Ours.yaml."""
    cfg = cfg or {}
    if cfg.get("source_condition", True):
        h = on.relu(p.sub("cond_conv").conv(x, "0"))
        Q_plus = p.sub("cond_conv").conv(h, "2")
        Q = SparseTensor(Q.C, Q.F + Q_plus.features_at_coordinates(Q.C), Q.stride)
    x = on.relu(p.sub("pre_conv").conv(x, "0"))
    qp = p.sub("q_pre_conv")
    Q = qp.conv(on.relu(qp.conv(on.relu(qp.conv(Q, "0")), "2", ksize=1)), "4")
    preds = []
    for i in (1, 2, 3):
        qq = p.sub(f"q_predict_{i}")
        bg = qq.conv(on.relu(qq.conv(on.relu(qq.conv(Q, "0")), "2")), "4")
        x = scaled_block(p.sub(f"scale_{i}"), x, bg)
        x, pred, up_coords = up_block_predict(p.sub(f"up_{i}"), x, k[i - 1], dense=cfg.get("dense", True),
                                              full_predictions=coords is not None)
        Q = up_block_follow(p.sub(f"q_up_{i}"), Q, up_coords)
        preds.append(pred)
    pc = p.sub("post_conv")
    x = pc.conv(on.relu(pc.conv(on.relu(pc.conv(x, "0")), "2")), "4")
    if coords is not None:
        p1 = oc.stride_map(coords, 1)
        p2 = oc.stride_map(p1, 2)
        return x, [p2, p1, coords], preds
    return x


# ----------------------------------------------------------------------------------------
# entropy_models.py
# ----------------------------------------------------------------------------------------
def h_a(p, y):
    h = on.leaky_relu(p.conv(y, "0"))
    h = p.conv(h, "2", 3, 2)
    h = on.leaky_relu(p.conv(h, "3"))
    h = p.conv(h, "5", 3, 2)
    return p.conv(h, "6")


def h_s(p, z):
    """entropy_models.py:284-294; the Sorted* shims (:12-102) only permute rows."""
    h = p.conv(z, "0")
    h = on.leaky_relu(p.convT(h, "1", 2))
    h = p.conv(h, "3")
    h = on.leaky_relu(p.convT(h, "4", 2))
    return p.conv(h, "6")


def h_q(p, z):
    """entropy_models.py:296-306 on a fresh z_hat (decode semantics)."""
    h = p.conv(z, "0")
    h = on.relu(p.convT(h, "1", 3))
    h = p.conv(h, "3")
    h = on.relu(p.convT(h, "4", 3))
    return p.conv(h, "6")


class Hyperprior:
    """``MeanScaleHyperprior`` (entropy_models.py:104-250): h_a, factorized z, h_s, Gaussian y — the entropy model the
    two-hyperprior variant of ColorModel (model/model.py:22-24) instantiates twice, for y and for the stride-8 q-map.
    The same layers as MeanScaleHyperprior_Map minus h_q."""

    def __init__(self, p, gc):
        self.p = p
        self.eb = EntropyBottleneck(p.sub("entropy_bottleneck"))
        self.gc = gc

    def _gaussian_params(self, z_hat, y_coords):
        gp = h_s(self.p.sub("h_s"), z_hat).features_at_coordinates(y_coords)
        scales, means = gp.chunk(2, dim=1)
        return scales.t().unsqueeze(0).contiguous(), means.t().unsqueeze(0).contiguous()

    def compress(self, y):
        """entropy_models.py:172-212 -> (strings [y, z], shape, intermediates)"""
        z = h_a(self.p.sub("h_a"), y)
        y, z = y.sorted(), z.sorted()
        shape = [z.F.shape[0]]
        z_strings = self.eb.compress(z.F.t().unsqueeze(0))
        z_hat = SparseTensor(z.C, self.eb.decompress(z_strings, shape[0])[0].t().contiguous(), 32)
        scales, means = self._gaussian_params(z_hat, y.C)
        indexes = self.gc.build_indexes(scales)
        y_strings = self.gc.compress(y.F.t().unsqueeze(0).contiguous(), indexes, means)
        return [y_strings, z_strings], shape, dict(y=y, z=z, z_hat=z_hat, scales=scales, means=means, indexes=indexes)

    def decompress(self, y_pts, z_pts, strings, shape):
        """entropy_models.py:215-250 -> y_hat on the canonically sorted stride-8 coordinates"""
        z_hat = SparseTensor(z_pts, self.eb.decompress(strings[1], shape[0])[0].t().contiguous(), 32)
        scales, means = self._gaussian_params(z_hat, y_pts)
        y_hat = self.gc.decompress(strings[0], self.gc.build_indexes(scales), means)
        return SparseTensor(y_pts, y_hat[0].t().contiguous(), 8)

    def forward_eval(self, y):
        """entropy_models.py:145-169 in eval mode -> y_hat, (L_y, L_z), z's coordinates"""
        z = h_a(self.p.sub("h_a"), y)
        z_hat_f, z_lik = self.eb.forward_eval(z.F.t().unsqueeze(0))
        z_hat = SparseTensor(z.C, z_hat_f[0].t().contiguous(), 32)
        scales, means = self._gaussian_params(z_hat, y.C)
        y_hat_f, y_lik = self.gc.forward_eval(y.F.t().unsqueeze(0), scales, means)
        return SparseTensor(y.C, y_hat_f[0].t().contiguous(), 8), (y_lik, z_lik), z.C


class Codec:
    """Functional restatement of ``ColorModel`` (model/model.py:15-208)."""

    def __init__(self, state_dict, config=None, leaves=False):
        """``config``: the model section of the yaml ({"g_a": {...}, "g_s": {...}, ...}); None = configs/Ours.yaml.  Only the
        switches that change the forward pass are read: source_condition, condition_ablation (g_a), dense (g_s), and the presence
        of an "entropy_model_map" section.  ``leaves``: use the tensors of ``state_dict`` as they are (autograd leaves of
        oracle/train.py) instead of detached copies."""
        self.cfg_a = dict((config or {}).get("g_a", {}))
        self.cfg_s = dict((config or {}).get("g_s", {}))
        self.sd = state_dict if leaves else {k: torch.as_tensor(v).detach().to(torch.float32).cpu() for k, v in state_dict.items()
                                             if torch.as_tensor(v).dtype.is_floating_point}
        self.p = Params(self.sd)
        self.eb = EntropyBottleneck(self.p.sub("entropy_model").sub("entropy_bottleneck"))
        self.gc = GaussianConditional()
        # model/model.py:22-27: an "entropy_model_map" section selects two MeanScaleHyperprior models (y and the stride-8 q-map)
        self.two = "entropy_model_map" in (config or {})
        if self.two:
            self.em_y = Hyperprior(self.p.sub("entropy_model"), self.gc)
            self.em_y.eb = self.eb
            self.em_q = Hyperprior(self.p.sub("entropy_model_map"), self.gc)
        self.updated = False

    def update(self):
        """model/model.py:30-36."""
        self.eb.update()
        if self.two:
            self.em_q.eb.update()
        self.gc.update()
        self.updated = True

    def aux_loss(self):
        """model/model.py:40-47"""
        return self.eb.aux_loss() + (self.em_q.eb.aux_loss() if self.two else 0.0)

    # -- model.py:95-147 -------------------------------------------------------------
    def compress(self, x, Q_coords, Q_feats, batch=None):
        """x: float [N,6]; q-map given as coords [N,4] (batch first) + feats [N,2].

        Returns (strings, shape, k, coordinates) like the in-memory API.  ``batch`` (int [N], optional):
        item index of every point — the spatial-blocks-as-batch-items mode of SURVEY.md §8e: the model
        already counts k and selects top-k per item (transforms.py:65-71, blocks.py:130-150); one
        (y, z) stream pair covers all items in canonical (b, x, y, z) order.
        """
        assert self.updated, "call update() first (evaluate.py:80-84)"
        x = np.asarray(x, dtype=np.float32)
        N = x.shape[0]
        bcol = np.zeros((N, 1), np.float32) if batch is None else np.asarray(batch, dtype=np.float32).reshape(N, 1)
        pts = np.concatenate([bcol, x[:, :3]], axis=1).astype(np.int32)
        feats = torch.from_numpy(np.concatenate([np.ones((N, 1), np.float32), x[:, 3:6]], axis=1))
        inp = SparseTensor(pts, feats, 1)
        Q = SparseTensor(oc.to_int_coords(Q_coords), torch.as_tensor(Q_feats, dtype=torch.float32), 1)
        y, Q8, k = analysis(self.p.sub("g_a"), inp, Q, self.cfg_a)
        if self.two:
            # model/model.py:132-136: strings = [[y, z] of the latents, [y, z] of the q-map], shape likewise
            y_strings, y_shape, self.last = self.em_y.compress(y)
            q_strings, q_shape, self.last_q = self.em_q.compress(Q8)
            return [y_strings, q_strings], [y_shape, q_shape], k, y.C.copy()
        em = self.p.sub("entropy_model")
        z = h_a(em.sub("h_a"), y)
        y = y.sorted()
        z = z.sorted()
        shape = [z.F.shape[0]]
        z_strings = self.eb.compress(z.F.t().unsqueeze(0))
        z_hat = self.eb.decompress(z_strings, shape[0])
        z_hat = SparseTensor(z.C, z_hat[0].t().contiguous(), 32)
        params = h_s(em.sub("h_s"), z_hat)
        gp = params.features_at_coordinates(y.C)
        scales, means = gp.chunk(2, dim=1)
        scales = scales.t().unsqueeze(0).contiguous()
        means = means.t().unsqueeze(0).contiguous()
        indexes = self.gc.build_indexes(scales)
        y_strings = self.gc.compress(y.F.t().unsqueeze(0).contiguous(), indexes, means)
        self.last = dict(y=y, z=z, z_hat=z_hat, scales=scales, means=means, indexes=indexes)
        return [y_strings, z_strings], shape, k, y.C.copy()

    # -- model.py:152-208 ------------------------------------------------------------
    def decompress(self, coordinates, strings, shape, k):
        assert self.updated
        c8 = oc.to_int_coords(coordinates)
        c16 = oc.stride_map(c8, 8)
        c32 = oc.stride_map(c16, 16)
        y_pts = c8[oc.sort_order(c8)]
        z_pts = c32[oc.sort_order(c32)]
        if self.two:
            # model/model.py:197-201: both models decode on the same [stride-8, stride-32] point lists
            y_hat = self.em_y.decompress(y_pts, z_pts, strings[0], shape[0])
            Q_hat = self.em_q.decompress(y_pts, z_pts, strings[1], shape[1])
            x_hat = synthesis(self.p.sub("g_s"), y_hat, Q_hat, k, cfg=self.cfg_s)
            feats = torch.clamp(torch.round(x_hat.F * 255), 0.0, 255.0) / 255
            self.last_dec = dict(y_hat=y_hat, Q_hat=Q_hat, x_hat=x_hat)
            self.last_batch = x_hat.C[:, 0].copy()
            return np.concatenate([x_hat.C[:, 1:4].astype(np.float32), feats.numpy()], axis=1)
        em = self.p.sub("entropy_model")
        z_hat = self.eb.decompress(strings[1], shape[0])
        z_hat = SparseTensor(z_pts, z_hat[0].t().contiguous(), 32)
        Q_hat = h_q(em.sub("h_q"), z_hat)
        params = h_s(em.sub("h_s"), z_hat)
        gp = params.features_at_coordinates(y_pts)
        scales, means = gp.chunk(2, dim=1)
        scales = scales.t().unsqueeze(0).contiguous()
        means = means.t().unsqueeze(0).contiguous()
        indexes = self.gc.build_indexes(scales)
        y_hat = self.gc.decompress(strings[0], indexes, means)
        y_hat = SparseTensor(y_pts, y_hat[0].t().contiguous(), 8)
        x_hat = synthesis(self.p.sub("g_s"), y_hat, Q_hat, k, cfg=self.cfg_s)
        feats = torch.clamp(torch.round(x_hat.F * 255), 0.0, 255.0) / 255
        self.last_dec = dict(y_hat=y_hat, Q_hat=Q_hat, x_hat=x_hat)
        self.last_batch = x_hat.C[:, 0].copy()
        return np.concatenate([x_hat.C[:, 1:4].astype(np.float32), feats.numpy()], axis=1)

    # -- model.py:51-93 (eval mode) --------------------------------------------------
    def forward_eval(self, coords, colors, Q_coords, Q_feats):
        coords = oc.to_int_coords(coords)
        N = coords.shape[0]
        feats = torch.cat([torch.ones((N, 1)), torch.as_tensor(colors, dtype=torch.float32)], dim=1)
        x = SparseTensor(coords, feats, 1)
        Q = SparseTensor(oc.to_int_coords(Q_coords), torch.as_tensor(Q_feats, dtype=torch.float32), 1)
        y, Q8, k = analysis(self.p.sub("g_a"), x, Q, self.cfg_a)
        if self.two:
            # model/model.py:75-78: likelihoods = {"y": [L_y, L_Q], "z": [L_zy, L_zQ]}
            y_hat, (y_lik, zy_lik), z_rows = self.em_y.forward_eval(y)
            Q_hat, (q_lik, zq_lik), zq_rows = self.em_q.forward_eval(Q8)
            x_hat, points, preds = synthesis(self.p.sub("g_s"), y_hat, Q_hat, k, coords=coords, cfg=self.cfg_s)
            return {"prediction": x_hat, "points": points, "occ_predictions": preds,
                    "likelihoods": {"y": [y_lik, q_lik], "z": [zy_lik, zq_lik]}, "k": k,
                    "rows": {"y": y.C, "z": z_rows, "q": Q8.C, "zq": zq_rows}, "y_hat": y_hat, "Q_hat": Q_hat}
        em = self.p.sub("entropy_model")
        z = h_a(em.sub("h_a"), y)
        z_hat_f, z_lik = self.eb.forward_eval(z.F.t().unsqueeze(0))
        z_hat = SparseTensor(z.C, z_hat_f[0].t().contiguous(), 32)
        params = h_s(em.sub("h_s"), z_hat)
        Q_hat = h_q(em.sub("h_q"), z_hat)
        gp = params.features_at_coordinates(y.C)
        scales, means = gp.chunk(2, dim=1)
        y_hat_f, y_lik = self.gc.forward_eval(y.F.t().unsqueeze(0), scales.t().unsqueeze(0), means.t().unsqueeze(0))
        y_hat = SparseTensor(y.C, y_hat_f[0].t().contiguous(), 8)
        x_hat, points, preds = synthesis(self.p.sub("g_s"), y_hat, Q_hat, k, coords=coords, cfg=self.cfg_s)
        return {"prediction": x_hat, "points": points, "occ_predictions": preds,
                "likelihoods": {"y": y_lik, "z": z_lik}, "k": k,
                "rows": {"y": y.C, "z": z.C}}          # the coordinates the likelihood columns belong to, in column order


# ----------------------------------------------------------------------------------------
# container (model/model.py:214-315): 7 x int32 header then payloads
# ----------------------------------------------------------------------------------------
def pack_container(shape, points_bitstream, strings, k):
    """Header [N_z, len_gpcc, len_y, len_z, k0, k1, k2] as big-endian int32
    (the `bitstream` package writes MSB first), 28 bytes,
    then gpcc bytes, y bytes, z bytes (model.py:243-256)."""
    ks = [int(kk[0]) if isinstance(kk, (list, tuple)) else int(kk) for kk in k]
    hdr = struct.pack(">7i", int(shape[0]), len(points_bitstream), len(strings[0][0]), len(strings[1][0]), *ks)
    return hdr + bytes(points_bitstream) + strings[0][0] + strings[1][0]


def unpack_container(data):
    nz, lg, ly, lz, k0, k1, k2 = struct.unpack(">7I", data[:28])
    o = 28
    gp = data[o:o + lg]; o += lg
    ys = data[o:o + ly]; o += ly
    zs = data[o:o + lz]
    return gp, [[ys], [zs]], [int(nz)], [[int(k0)], [int(k1)], [int(k2)]]


def count_bits(strings):
    """utils.py:30-51."""
    total = 0
    for s in strings:
        total += count_bits(s) if isinstance(s, list) else len(s) * 8
    return total
