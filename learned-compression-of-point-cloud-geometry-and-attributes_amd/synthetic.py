"""Synthetic inputs and seeded weights for tests and benchmarks (no dataset / checkpoint exists:
README.md:218 "Trained weights: COMING SOON").

* ``OURS_CONFIG``: the ``model:`` section of /root/reference/configs/Ours.yaml:6-23.
* ``sphere_shell``: voxelised sphere shells — config 1 (32^3, N=4,904) and config 2 (1024^3,
  radius 260, N=850,824 ~ longdress_vox10_1300's 857,966 points).
* ``seeded_init``: variance-preserving seeded initialisation.  ME's default U(+-1/sqrt(27 C_in))
  shrinks the signal ~x0.15 per layer at 10-15 active neighbours, collapsing the latents to the
  biases after ~40 layers, so fixtures use N(0, gain^2 / (C_in * n_active)) instead and spread
  the hyper-decoder's scale outputs over several of the 64 table levels.
Pure numpy / torch-CPU; no dependency on the HIP library.
"""
import math

import numpy as np
import torch

OURS_CONFIG = {
    "entropy_model": {"type": "MeanScaleHyperprior_map", "C_bottleneck": 128, "C_hyper_bottleneck": 128, "C_Q": 2},
    "g_a": {"C_in": 4, "N1": 64, "N2": 128, "N3": 128, "source_condition": True},
    "g_s": {"C_out": 3, "N1": 128, "N2": 128, "N3": 64, "source_condition": True},
}

# the ``model:`` section of /root/reference/configs/Ablation_NoCondition_Convolution.yaml:6-25 (Ablation_200_logarithmic.yaml's
# model section equals Ours.yaml's): no source-conditioned q-map correction, the down-sampled q-map itself as beta | gamma
ABLATION_NOCONDITION_CONFIG = {
    "entropy_model": {"type": "MeanScaleHyperprior_map", "C_bottleneck": 128, "C_hyper_bottleneck": 128, "C_Q": 2},
    "g_a": {"C_in": 4, "N1": 64, "N2": 128, "N3": 128, "source_condition": False, "condition_ablation": "condition_ablation"},
    "g_s": {"C_out": 3, "N1": 128, "N2": 128, "N3": 64, "source_condition": False, "condition_ablation": "condition_ablation"},
}

# the two-hyperprior variant model/model.py:22-24 selects when the config has an "entropy_model_map" section (no shipped yaml has one;
# the widths of the q-map's model are this build's choice: 2 channels in, 8 in the hyper bottleneck -> h_s widths 8, 8, 3, 4)
TWO_HYPERPRIOR_CONFIG = {
    "entropy_model": {"type": "MeanScaleHyperprior", "C_bottleneck": 128, "C_hyper_bottleneck": 128},
    "entropy_model_map": {"type": "MeanScaleHyperprior", "C_bottleneck": 2, "C_hyper_bottleneck": 8},
    "g_a": {"C_in": 4, "N1": 64, "N2": 128, "N3": 128, "source_condition": True},
    "g_s": {"C_out": 3, "N1": 128, "N2": 128, "N3": 64, "source_condition": True},
}

CONFIG1 = dict(grid=32, radius=15.0, half_width=0.875)       # N = 4,904
CONFIG2 = dict(grid=1024, radius=260.0, half_width=0.5)      # N = 850,824


def sphere_shell(grid, radius, half_width, seed=0, noise=0.0, center=None):
    """float32 [N, 6]: voxel coordinates (as floats) + rgb in {k/255}.  Rows in raster (x,y,z) order."""
    c = (grid - 1) / 2.0 if center is None else center
    lo = int(max(0, math.floor(c - radius - half_width - 1)))
    hi = int(min(grid - 1, math.ceil(c + radius + half_width + 1)))
    pts = []
    ax = np.arange(lo, hi + 1, dtype=np.float64)
    yy, zz = np.meshgrid(ax, ax, indexing="ij")
    for x in ax:                                   # slab by slab keeps memory small at 1024^3
        d = np.sqrt((x - c) ** 2 + (yy - c) ** 2 + (zz - c) ** 2)
        m = np.abs(d - radius) < half_width
        if m.any():
            p = np.stack([np.full(m.sum(), x), yy[m], zz[m]], axis=1)
            pts.append(p)
    p = np.concatenate(pts, axis=0)
    phase = np.array([0.0, 2 * np.pi / 3, 4 * np.pi / 3])
    rgb = 0.5 + 0.5 * np.sin(2 * np.pi * p[:, :1] / grid * np.array([1.0, 2.0, 3.0]) + 2 * np.pi * p[:, 1:2] / grid
                             + 2 * np.pi * p[:, 2:3] / grid * 0.5 + phase)
    if noise > 0:
        rng = np.random.default_rng(seed)
        rgb = rgb + rng.normal(0.0, noise, rgb.shape)
    rgb = np.round(np.clip(rgb, 0.0, 1.0) * 255.0) / 255.0
    return np.concatenate([p, rgb], axis=1).astype(np.float32)


def uniform_qmap(points_xyz, q_g=0.5, q_a=0.5):
    """(coords [N,4] float with batch 0, feats [N,2] = [q_g, q_a]) — utils.py:436-445 layout."""
    n = points_xyz.shape[0]
    coords = np.concatenate([np.zeros((n, 1), np.float32), points_xyz.astype(np.float32)], axis=1)
    feats = np.concatenate([np.full((n, 1), q_g, np.float32), np.full((n, 1), q_a, np.float32)], axis=1)
    return coords, feats


# FiLM-head gain of the seeded initialisation.  0.1 (the default, and the headline workload's) keeps (beta, gamma) within a
# few percent of (1, 0), so the quality map barely moves the rate: the four operating points of BASELINE config 3 then span
# 0.2 % in bpp.  1.0 makes the heads respond: on a 17 k-point frame the (q_g, q_a) grid of plot.py:31-32 codes at 8.48 /
# 8.52 / 8.70 / 9.52 bpp — a monotone rate axis (the distortion axis stays that of random weights).  Config-3 tests and
# tools/rd_sweep.py use the responsive variant; the oracle loads the same state_dict.
FILM_GAIN_DEFAULT = 0.1
FILM_GAIN_Q_RESPONSIVE = 1.0


@torch.no_grad()
def seeded_init(model, seed=0, film_gain=FILM_GAIN_DEFAULT):
    """Deterministic variance-preserving init of a ColorModel (CPU generator, device independent).

    kernel ~ N(0, gain^2 / (C_in * n_active)) with gain sqrt(2) in front of a ReLU and n_active the
    typical number of occupied kernel offsets for the layer's position in the network; FiLM heads
    start near (beta, gamma) = (1, 0); the residual branch of every ScaledBlock is damped.
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    from .sparse import ACT_NONE, ConvChain, _ConvBase
    relu_followed = set()
    for _, chain in model.named_modules():
        if isinstance(chain, ConvChain):
            for conv, act in chain.plan():
                if act != ACT_NONE:
                    relu_followed.add(id(conv))
    for name, m in model.named_modules():
        if not isinstance(m, _ConvBase):
            continue
        K = m.kernel_size ** 3
        if m.transposed:
            n_active = 1.0 if m.kernel_size == 2 else 2.2      # parents per child
        elif m.stride == 2:
            n_active = 9.0
        elif K == 1:
            n_active = 1.0
        elif ".up_" in name and (".conv_2." in name or ".occ_predict." in name):
            n_active = 21.0                                     # dilated candidate sets are dense
        else:
            n_active = 12.0
        gain = math.sqrt(2.0) if id(m) in relu_followed else 1.0
        std = gain / math.sqrt(m.in_channels * n_active)
        m.kernel.copy_((torch.randn(m.kernel.shape, generator=g) * std).to(m.kernel.device))
        if m.bias is not None:
            m.bias.copy_((torch.randn(m.bias.shape, generator=g) * 0.05).to(m.bias.device))
    # FiLM heads: beta ~ 1, gamma ~ 0
    heads = [model.g_a.condition_encoder.predict_layers[i][4] for i in range(3)]
    heads += [model.g_s.q_predict_1[4], model.g_s.q_predict_2[4], model.g_s.q_predict_3[4]]
    for h in heads:
        n = h.out_channels // 2
        h.kernel.mul_(film_gain)
        h.bias[:, :n].add_(1.0)
    # residual branches: keep ScaledBlocks close to identity + perturbation
    for blk in (model.g_a.scale_1, model.g_a.scale_2, model.g_a.scale_3,
                model.g_s.scale_1, model.g_s.scale_2, model.g_s.scale_3):
        blk.conv_2[2].kernel.mul_(0.5)
    em = model.entropy_model
    # latents a few integer bins wide; scales spread over several table levels
    model.g_a.post_conv.kernel.mul_(0.4)
    last = em.h_s[6]
    C = em.C_bottleneck
    last.kernel[:, :, :C].mul_(0.15)      # scales: ~[0.3, 1.5]
    last.bias[:, :C].add_(0.8)
    last.kernel[:, :, C:].mul_(0.1)       # means: small against the latent's spread
    # colours: keep x_hat mostly inside [0, 1] so the 8-bit rounding (model/model.py:206) is exercised
    out = model.g_s.post_conv[4]
    out.kernel.mul_(0.02)
    out.bias.fill_(0.5)
    eb = em.entropy_bottleneck
    for i in range(5):
        b = getattr(eb, f"_bias{i}")
        b.copy_((torch.rand(b.shape, generator=g) - 0.5).to(b.device))
    emq = getattr(model, "entropy_model_map", None)
    if emq is not None:
        # the second hyperprior codes the stride-8 q-map ROUNDED TO INTEGERS (model/model.py:77): spread it over several bins
        model.g_a.condition_encoder.down_layers[2].kernel.mul_(3.0)
        last = emq.h_s[6]
        C = emq.C_bottleneck
        last.kernel[:, :, :C].mul_(0.15)
        last.bias[:, :C].add_(0.8)
        last.kernel[:, :, C:].mul_(0.1)
        for i in range(5):
            b = getattr(emq.entropy_bottleneck, f"_bias{i}")
            b.copy_((torch.rand(b.shape, generator=g) - 0.5).to(b.device))
    return model


def make_model(seed=0, device="cpu", config=None, film_gain=FILM_GAIN_DEFAULT):
    from .model import ColorModel
    model = ColorModel(config or OURS_CONFIG)
    seeded_init(model, seed, film_gain)
    model = model.to(device).eval()
    return model
