"""Oracle: sparse tensor + sparse convolution arithmetic (torch-CPU fp32).

Restates the MinkowskiEngine operators used on the path (SURVEY.md §2.2 N3-N9,
Appendix B.1): ``out[j] = bias + sum_k in[nbr(j,k)] @ W[k]`` with the kernel
tensor ``[K, C_in, C_out]`` (``[C_in, C_out]`` for kernel_size 1) and bias
``[1, C_out]``, accumulated per kernel offset in fixed order k = 0..K-1
(gather -> sgemm -> index_add_).

Two summation orders (``set_order``):
  * ``"blas"`` (default): the above — an independent statement of the operator, whose fp32 sums differ from the
    product's in the last bits (MKL's blocking against the MFMA chain); comparisons against it carry tolerances.
  * ``"kernel"``: every output element is ONE fused multiply-add chain in the order the product's kernels document
    (oracle/chain.c), including which layer shapes the product evaluates as a thin kernel, an MFMA tile or a narrow head
    (scores per input row, then a plain sum over the offsets).  Against this mode latents, streams and decoded voxel
    sets are compared for EQUALITY (tests/test_exact_parity.py).

Test infrastructure only — see oracle/__init__.py.
"""
from dataclasses import dataclass, field

import numpy as np
import torch

from . import coords as oc


@dataclass
class SparseTensor:
    """Minimal stand-in for ME.SparseTensor: ``C`` int32 [N,4], ``F`` fp32 [N,C]."""
    C: np.ndarray
    F: torch.Tensor
    stride: int = 1
    _cache: dict = field(default_factory=dict, repr=False)

    def __post_init__(self):
        self.C = oc.to_int_coords(self.C)
        self.F = torch.as_tensor(self.F, dtype=torch.float32)
        assert self.C.shape[0] == self.F.shape[0]

    def features_at_coordinates(self, query):
        """On-grid lookup, zero-fill when absent (blocks.py:37,50)."""
        q = oc.to_int_coords(query)
        idx = oc.lookup(self.C, q)
        out = torch.zeros((q.shape[0], self.F.shape[1]), dtype=torch.float32)
        hit = idx >= 0
        out[torch.from_numpy(hit)] = self.F[torch.from_numpy(idx[hit])]
        return out

    def sorted(self):
        """utils.sort_tensor (utils.py:155-180)."""
        order = oc.sort_order(self.C)
        return SparseTensor(self.C[order], self.F[torch.from_numpy(order)], self.stride)


ORDER = "blas"
NARROW_HEAD_MAX_COUT = 4          # the product's rule (pcc_amd/sparse.py: conv_forward): cout <= 4 on inputs of a multiple of 32 channels


def set_order(order):
    """"blas" or "kernel" (module docstring); returns the previous order"""
    global ORDER
    assert order in ("blas", "kernel"), order
    was, ORDER = ORDER, order
    return was


def _apply_conv_kernel_order(F_in, W, bias, nbr, n_out):
    """The product's summation order, layer shape by layer shape (csrc/conv.hip; dispatch in pcc_amd/sparse.py:conv_forward):
    cin % 32 != 0 -> thin kernel, channels ascending; cin % 32 == 0 -> MFMA visit order; kernel_size > 1 with cout <= 4 on
    such inputs -> narrow head: scores[i, k, c] = in[i] . W[k][:, c] (an MFMA chain per input row), out = sum_k scores."""
    from . import chain
    K, cin, cout = W.shape
    F_np, W_np = F_in.detach().numpy(), W.detach().numpy()
    if nbr is not None and K > 1 and cin % 32 == 0 and cout <= NARROW_HEAD_MAX_COUT:
        w_r = np.ascontiguousarray(np.transpose(W_np, (1, 0, 2)).reshape(1, cin, K * cout))
        scores = chain.conv_chain(F_np, w_r, None, F_np.shape[0], True)
        out = chain.gather_sum(scores, nbr, cout)
    else:
        out = chain.conv_chain(F_np, W_np, nbr, n_out, cin % 32 == 0)
    out = torch.from_numpy(out)
    if bias is not None:
        out = out + bias.reshape(1, -1)          # the epilogue adds the bias to the finished accumulator (csrc/conv.hip:861)
    return out


def _apply_conv(F_in, W, bias, nbr, n_out):
    K = nbr.shape[1]
    if W.dim() == 2:
        W = W[None]
    assert W.shape[0] == K, (W.shape, K)
    if ORDER == "kernel":
        return _apply_conv_kernel_order(F_in, W, bias, nbr, n_out)
    out = torch.zeros((n_out, W.shape[2]), dtype=torch.float32)
    for k in range(K):
        col = nbr[:, k]
        rows = np.nonzero(col >= 0)[0]
        if rows.size == 0:
            continue
        src = torch.from_numpy(col[rows])
        out.index_add_(0, torch.from_numpy(rows), F_in[src] @ W[k])
    if bias is not None:
        out += bias.reshape(1, -1)
    return out


def conv(x, W, bias=None, ksize=3, stride=1):
    """ME.MinkowskiConvolution (transforms.py:35-57 etc.)."""
    if ksize == 1:
        assert stride == 1
        Wk = W if W.dim() == 2 else W[0]
        if ORDER == "kernel":
            y = SparseTensor(x.C, _apply_conv_kernel_order(x.F, Wk[None], bias, None, x.C.shape[0]), x.stride)
            y._cache = x._cache
            return y
        out = x.F @ Wk
        if bias is not None:
            out = out + bias.reshape(1, -1)
        return SparseTensor(x.C, out, x.stride)
    if stride == 1:
        key = ("nbr", ksize, 1)
        if key not in x._cache:
            x._cache[key] = oc.kernel_map(x.C, x.C, ksize, x.stride)
        nbr = x._cache[key]
        y = SparseTensor(x.C, _apply_conv(x.F, W, bias, nbr, x.C.shape[0]), x.stride)
        y._cache = x._cache  # same coordinate map -> share kernel maps (ME caches them too)
        return y
    assert stride == 2
    key = ("down", ksize)
    if key not in x._cache:
        out_c = oc.stride_map(x.C, x.stride)
        x._cache[key] = (out_c, oc.kernel_map(x.C, out_c, ksize, x.stride))
    out_c, nbr = x._cache[key]
    return SparseTensor(out_c, _apply_conv(x.F, W, bias, nbr, out_c.shape[0]), x.stride * 2)


def conv_transpose_generative(x, W, bias=None, ksize=3):
    """ME.MinkowskiGenerativeConvolutionTranspose, stride 2 (blocks.py:84;
    entropy_models.py:286,290) — also the non-generative ConvTranspose of h_q on a
    fresh coordinate manager (entropy_models.py:298,302)."""
    key = ("up", ksize)
    if key not in x._cache:
        out_c = oc.children(x.C, x.stride, ksize)
        x._cache[key] = (out_c, oc.kernel_map(x.C, out_c, ksize, x.stride // 2, transposed=True))
    out_c, nbr = x._cache[key]
    return SparseTensor(out_c, _apply_conv(x.F, W, bias, nbr, out_c.shape[0]), x.stride // 2)


# Activation gates of the implementation under test, forced onto the oracle (tests/test_train_model.py).  A ReLU is
# discontinuous in its gradient: a pre-activation within the two implementations' ~1e-7 difference of zero is gated open on
# one side and closed on the other, and that one gate changes the gradients of its whole neighbourhood — a legitimate
# difference that would otherwise have to be covered by a loose gradient tolerance.  With FORCED_GATES = {layer name:
# (coordinates [n, 4], gate bool [n, C])} the activation of that layer multiplies by the GIVEN gate (rows matched by
# coordinate; rows the other side did not evaluate keep their own gate), so both sides differentiate the same piecewise-
# linear function and gradients can be held to the tight tolerance; every element whose own sign disagrees with the forced
# gate is logged to GATE_FLIPS as (layer, |pre-activation|, largest |pre-activation| of the layer).
FORCED_GATES = None
GATE_FLIPS = None


def _activate(x, neg_slope):
    tag = getattr(x, "tag", None)
    own = x.F > 0
    gate = own
    if FORCED_GATES is not None and tag in FORCED_GATES:
        f_coords, f_gate = FORCED_GATES[tag]
        idx = oc.lookup(oc.to_int_coords(f_coords), x.C)
        hit = torch.from_numpy(idx >= 0)
        gate = own.clone()
        gate[hit] = torch.as_tensor(f_gate, dtype=torch.bool)[torch.from_numpy(idx[idx >= 0])]
        if GATE_FLIPS is not None:
            flipped = gate != own
            if bool(flipped.any()):
                top = float(x.F.detach().abs().max())
                GATE_FLIPS.extend((tag, float(v), top) for v in x.F.detach()[flipped].abs().tolist())
    out = x.F * gate if neg_slope == 0.0 else x.F * torch.where(gate, 1.0, neg_slope)
    y = SparseTensor(x.C, out, x.stride)
    y._cache = x._cache
    return y


def relu(x):
    if FORCED_GATES is not None:
        return _activate(x, 0.0)
    y = SparseTensor(x.C, torch.relu(x.F), x.stride)
    y._cache = x._cache
    return y


def leaky_relu(x, slope=0.01):
    if FORCED_GATES is not None:
        return _activate(x, slope)
    y = SparseTensor(x.C, torch.nn.functional.leaky_relu(x.F, slope), x.stride)
    y._cache = x._cache
    return y


def prune(x, mask):
    """ME.MinkowskiPruning: order-preserving row compaction (blocks.py:90,126)."""
    m = np.asarray(mask, dtype=bool)
    return SparseTensor(x.C[m], x.F[torch.from_numpy(m)], x.stride)


class Params:
    """View on a flat state_dict with a dotted prefix."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def sub(self, name):
        return Params(self.sd, f"{self.prefix}{name}.")

    def get(self, name, default=None):
        v = self.sd.get(self.prefix + name, default)
        return None if v is None else torch.as_tensor(v, dtype=torch.float32)

    def conv(self, x, name, ksize=3, stride=1, out_channels=None):
        """``out_channels``: evaluate only the first columns of the kernel (where the reference reads only channel 0,
        blocks.py:142).  A column's value does not depend on the others in "blas" order up to BLAS blocking; in "kernel"
        order the narrower layer is the shape the product evaluates (a narrow head), so the slice is part of the order."""
        p = self.sub(name)
        W, b = p.get("kernel"), p.get("bias")
        if out_channels is not None:
            W = W[..., :out_channels].contiguous()
            b = None if b is None else b.reshape(-1)[:out_channels].contiguous()
        y = conv(x, W, b, ksize, stride)
        y.tag = self.prefix + name            # the layer's state_dict name: keys FORCED_GATES
        return y

    def convT(self, x, name, ksize=3):
        p = self.sub(name)
        y = conv_transpose_generative(x, p.get("kernel"), p.get("bias"), ksize)
        y.tag = self.prefix + name
        return y
