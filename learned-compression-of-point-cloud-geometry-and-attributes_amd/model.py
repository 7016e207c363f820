"""ColorModel — the codec facade (drop-in for /root/reference/model/model.py:15-208).

Same constructor config, same ``forward`` / ``compress`` / ``decompress`` / ``update`` /
``aux_loss`` signatures and return values as the reference, running on libpcc_hip.so.

File mode (``path=...``): the reference shells out to the MPEG G-PCC ``tmc3`` binary for the
stride-8 latent coordinates (model/model.py:318-395); that binary is external and unavailable
(SURVEY.md N15, §8f rank 2).  The 28-byte header and payload order of the container are kept
(model/model.py:241-256); the coordinate payload is written by the build's own lossless octree
coder (octree.py + csrc/octree.hip, "PCO1"), which is NOT G-PCC-compatible.
"""
import struct

import torch
import torch.nn as nn

from . import octree
from . import sparse as sp
from .entropy_models import MeanScaleHyperprior, MeanScaleHyperprior_Map, _canonical_map
from .sparse import CoordMap, SparseTensor
from .transforms import AnalysisTransform, SparseSynthesisTransform


class ColorModel(nn.Module):
    LATENT_STRIDE = 8      # three stride-2 stages of g_a (model/transforms.py:75-128)

    def __init__(self, config):
        super().__init__()
        self.g_a = AnalysisTransform(config["g_a"])
        self.g_s = SparseSynthesisTransform(config["g_s"])
        if "entropy_model_map" in config:
            # model/model.py:22-24: one hyperprior for y, a second one codes the stride-8 q-map (no shipped config selects it)
            self.entropy_model = MeanScaleHyperprior(config["entropy_model"])
            self.entropy_model_map = MeanScaleHyperprior(config["entropy_model_map"])
        else:
            self.entropy_model = MeanScaleHyperprior_Map(config["entropy_model"])
            self.entropy_model_map = None

    # -- model/model.py:30-47 ------------------------------------------------------------------
    def update(self, force=True):
        updated = self.entropy_model.update(force=force)
        if self.entropy_model_map is not None:
            updated |= self.entropy_model_map.update(force=force)
        return updated

    def aux_loss(self):
        if self.entropy_model_map is None:
            return self.entropy_model.aux_loss()
        return self.entropy_model.aux_loss() + self.entropy_model_map.aux_loss()

    @property
    def device(self):
        return self.g_s.down_conv.kernel.device

    # -- model/model.py:51-93 (eval: rounding; train: noise quantisation, autograd through the HIP kernels) ----
    def forward(self, x, Q, Lambda=None):
        coords = SparseTensor(coordinate_map=x.map)
        ones = torch.ones((x.map.n, 1), dtype=torch.float32, device=x.device)
        x = SparseTensor(torch.cat([ones, x.F], dim=1), coordinate_map=x.map)
        y, Q, k = self.g_a(x, Q)
        if self.entropy_model_map is None:
            y_hat, Q_hat, likelihoods = self.entropy_model(y)
            likelihoods = {"y": likelihoods[0], "z": likelihoods[1]}
        else:
            y_hat, y_likelihoods = self.entropy_model(y)
            Q_hat, Q_likelihoods = self.entropy_model_map(Q)
            likelihoods = {"y": [y_likelihoods[0], Q_likelihoods[0]], "z": [y_likelihoods[1], Q_likelihoods[1]]}
        x_hat, points, predictions = self.g_s(y_hat, Q_hat, coords=coords, k=k)
        return {"prediction": x_hat, "points": points, "occ_predictions": predictions, "q_map": Lambda,
                "likelihoods": likelihoods}

    # -- model/model.py:95-147 -------------------------------------------------------------------------
    @torch.no_grad()
    def compress(self, x, Q, path=None, batch=None):
        """``batch`` (int [N] device tensor, optional; not in the reference's signature): item index per point
        for the spatial-blocks-as-batch-items mode of SURVEY.md §8e — k and top-k are per item
        (transforms.py:65-71, blocks.py:130-150), one (y, z) stream pair covers all items; ``k`` then holds
        one count per item and stage, and ``coordinates`` carries the item index in column 0."""
        N = x.shape[0]
        dev = x.device
        if N == 0 or x.dim() != 2 or x.shape[1] < 6:
            raise ValueError(f"compress: expected a [N, 6] tensor (xyz voxel coordinates + rgb) with N > 0, got {tuple(x.shape)}")
        if batch is None:
            bcol = torch.zeros((N, 1), device=dev, dtype=torch.int32)
            nbatch = 1
        else:
            if path:
                raise ValueError("file mode holds one cloud (the 28-byte header has one k per stage)")
            bcol = batch.to(device=dev, dtype=torch.int32).reshape(N, 1)
            nbatch = int(bcol.max().item()) + 1
        coords = torch.cat([bcol, x[:, :3].to(torch.int32)], dim=1)
        feats = torch.cat([torch.ones((N, 1), device=dev, dtype=torch.float32), x[:, 3:6].float()], dim=1)
        in_map = CoordMap(coords.contiguous(), 1, nbatch=nbatch)
        in_map.count_duplicates()
        inp = SparseTensor(feats, coordinate_map=in_map)
        if Q.map._nbatch is None:
            Q.map._nbatch = nbatch
        y, Q8, k = self.g_a(inp, Q)
        if self.entropy_model_map is None:
            points, strings, shape = self.entropy_model.compress(y)
        else:
            if path:
                raise ValueError("file mode: the container's header has fields for ONE (y, z) stream pair (model/model.py:243-250); "
                                 "the two-hyperprior variant returns two pairs — use the in-memory API")
            points, y_strings, y_shape = self.entropy_model.compress(y)
            _, Q_strings, Q_shape = self.entropy_model_map.compress(Q8)
            strings, shape = [y_strings, Q_strings], [y_shape, Q_shape]
        coordinates = y.C
        # (the range coder has waited for the stream: this read costs a few microseconds)
        dups = in_map.duplicates()
        if dups:
            raise ValueError(f"compress: {dups} of the {N} points repeat the voxel coordinates of an earlier point.  The reference's "
                             "ME.SparseTensor (model/model.py:121) would keep an unspecified one of each group; drop them first "
                             "(e.g. torch.unique on the coordinates, or pcc_amd.metrics._drop_duplicated_points: first occurrence wins)")
        if path:
            self.save_bitstream(path=path, points=coordinates, strings=strings, shape=shape, k=k)
            return None
        return strings, shape, k, coordinates

    # -- model/model.py:152-208 ------------------------------------------------------------------------
    @torch.no_grad()
    def decompress(self, path=None, coordinates=None, strings=None, shape=None, k=None, return_batch=False):
        """``return_batch`` (not in the reference's signature): also return the item index of every decoded
        point — for streams produced by ``compress(..., batch=...)``."""
        device = self.device
        if path and self.entropy_model_map is not None:
            raise ValueError("file mode holds one (y, z) stream pair; the two-hyperprior variant needs two (see compress)")
        if path:
            coordinates, strings, shape, k = self.load_bitstream(path)
            coordinates = coordinates.to(device)
            batch = torch.zeros((coordinates.shape[0], 1), device=device, dtype=coordinates.dtype)
            coordinates = torch.cat([batch, coordinates], dim=1)
        nbatch = len(k[0]) if isinstance(k[0], (list, tuple)) else 1          # one count per item and stage
        if self.entropy_model_map is None:
            self.entropy_model.start_z_decode(strings, shape, device)           # host work that needs no coordinates: first
        c8 = CoordMap(sp._as_int_coords(coordinates.to(device)), 8, nbatch=nbatch)
        c32 = c8.down().down()      # coordinates only (g_s.down_conv applied twice, model.py:188-190)
        if self.entropy_model_map is None:
            y_hat, Q_hat = self.entropy_model.decompress([c8, c32], strings, shape)
        else:
            points = [_canonical_map(c8, 8), _canonical_map(c32, 32)]          # model/model.py:197-201: both models decode on the same lists
            y_hat = self.entropy_model.decompress(points, strings[0], shape[0])
            Q_hat = self.entropy_model_map.decompress(points, strings[1], shape[1])
        return self.reconstruct(y_hat, Q_hat, k, return_batch)

    @torch.no_grad()
    def reconstruct(self, y_hat, Q_hat, k, return_batch=False):
        """The second half of decompress (model/model.py:196-208): g_s on entropy-decoded latents, 8-bit colour rounding.
        Public so that a decoder can be checked on latents it did not decode itself (tests/_parity.py)."""
        x_hat = self.g_s(y_hat, Q_hat, k=k)
        feats = torch.clamp(torch.round(x_hat.F * 255), 0.0, 255.0) / 255
        points = torch.cat([x_hat.C[:, 1:4].to(feats.dtype), feats], dim=1)
        return (points, x_hat.C[:, 0].clone()) if return_batch else points

    # -- container (model/model.py:214-315) --------------------------------------------------------------
    def gpcc_encode(self, points, directory=None):
        """model/model.py:318-364: latent coordinates [N,4] -> bytes.  ``directory`` (the reference's
        scratch location for tmc3's temporary files) is accepted and unused: nothing touches disk."""
        return octree.encode_coordinates(points, self.LATENT_STRIDE)

    def gpcc_decode(self, bin, directory=None):
        """model/model.py:366-395: bytes -> float64 [N,3] like the reference's PLY read-back (on the
        model's device, in the stream's Morton order; decompress() re-sorts canonically anyway)."""
        return octree.decode_coordinates(bin, self.device)[:, 1:4].to(torch.float64)

    def save_bitstream(self, path, points, strings, shape, k):
        pts = self.gpcc_encode(points, path)
        ks = [int(kk[0]) if isinstance(kk, (list, tuple)) else int(kk) for kk in k]
        # 7 x int32, MSB first like the `bitstream` package: 28 bytes
        header = struct.pack(">7i", int(shape[0]), len(pts), len(strings[0][0]), len(strings[1][0]), *ks)
        with open(path, "wb") as f:
            f.write(header + pts + strings[0][0] + strings[1][0])

    def load_bitstream(self, path):
        with open(path, "rb") as f:
            data = f.read()
        nz, lp, ly, lz, k0, k1, k2 = struct.unpack(">7I", data[:28])
        o = 28
        pts = data[o:o + lp]; o += lp
        ys = data[o:o + ly]; o += ly
        zs = data[o:o + lz]
        return self.gpcc_decode(pts, path), [[ys], [zs]], [int(nz)], [[int(k0)], [int(k1)], [int(k2)]]
