"""Differentiable sparse convolution (training path).

The reference trains through MinkowskiEngine's autograd functions (every ``ME.Minkowski*Convolution*``
in model/, driven by train.py:194-206).  Here one ``torch.autograd.Function`` wraps the HIP kernels:

    forward   out = bias + sum_k X[nbr[:, k]] @ W[k]                      pcc_conv_fwd
    backward  dX  = the same kernel over the transposed map with W[k]^T   pcc_kernel_map_transpose + pcc_conv_fwd
              dW[k] = sum over pairs of X[i]^T dY[j]                      pcc_conv_wgrad
              db  = column sums of dY

Activations, FiLM and residuals — fused into the convolution's epilogue on the inference path — are
ordinary torch ops here so that autograd differentiates them.
"""
import os

import torch

from . import _lib
from ._lib import check, ptr

_THIN_CIN = (1, 2, 3, 4, 6, 8, 12, 16, 24)

# bf16 compute for the training path (BASELINE config 5: "bf16"): convolutions whose input width is a multiple of
# 64 cast their input (forward: the features; backward-data: the output gradient) and weights to bf16 and run on
# v_mfma_f32_32x32x16_bf16 with fp32 accumulation and fp32 outputs; everything else — narrow and thin layers,
# the entropy models, the losses, the master weights — stays fp32; weight gradients of layers with both widths
# multiples of 64 take bf16 operands too (fp32 sums).  Off by default.
BF16 = os.environ.get("PCC_TRAIN_BF16", "0") == "1"


def set_bf16(enabled):
    global BF16
    BF16 = bool(enabled)


# Split-bf16 arithmetic for the fp32 training step (opt-in, PCC_TRAIN_X3=1 / set_x3): forward and backward-data
# convolutions with fp32 data run their products as six bf16 MFMA terms of an exact three-way split (csrc/conv.hip, X3):
# fp32-class values at 3/8 of the fp32 MFMA time.  Weight gradients stay on the fp32 kernel.  Ignored where BF16 applies.
X3 = os.environ.get("PCC_TRAIN_X3", "0") == "1"


def set_x3(enabled):
    global X3
    X3 = bool(enabled)


def _x3_ok(rows, cin, cout, n_out, K, has_nbr):
    return (X3 and cin % 32 == 0 and ((cout + 31) // 32 * 32) % 64 == 0 and rows * cin * 4 < 0xFFFFF000
            and (not has_nbr or n_out * K * 4 < 0xFFFFF000))


def _packed(w):
    """[K, cin, cout] -> MFMA packing (or None for thin cin)"""
    K, cin, cout = w.shape
    if cin % 32:
        return None
    L = _lib.lib()
    wp = torch.empty(L.pcc_conv_packed_elems(K, cin, cout), dtype=torch.float32, device=w.device)
    check(L.pcc_conv_pack_weights(ptr(w), K, cin, cout, ptr(wp), _lib.stream()))
    return wp


def _bf16_ok(rows, cin):
    return BF16 and cin % 64 == 0 and rows * cin * 2 < 0xFFFFF000


def _launch_conv(feats, w, bias, nbr, order, gmask, n_out):
    """feats: fp32, or an already cast bf16 copy (bf16 mode: one cast per tensor, shared by its consumers)"""
    L = _lib.lib()
    K, cin, cout = w.shape
    out = torch.empty((n_out, cout), dtype=torch.float32, device=feats.device)
    if feats.dtype == torch.bfloat16 or _bf16_ok(feats.shape[0], cin):
        wp = torch.empty(L.pcc_conv_packed_elems_bf16(K, cin, cout), dtype=torch.bfloat16, device=feats.device)
        check(L.pcc_conv_pack_weights_bf16(ptr(w), K, cin, cout, ptr(wp), _lib.stream()))
        x = feats if feats.dtype == torch.bfloat16 else feats.to(torch.bfloat16)
        check(L.pcc_conv_fwd_bf16(ptr(x), feats.shape[0], cin, ptr(wp), ptr(bias), ptr(nbr), ptr(order), ptr(gmask), K, ptr(out),
                                  n_out, cout, 0, None, None, _lib.stream()))
        return out
    if _x3_ok(feats.shape[0], cin, cout, n_out, K, nbr is not None):
        wp = torch.empty(L.pcc_conv_packed_elems_x3(K, cin, cout), dtype=torch.bfloat16, device=feats.device)
        check(L.pcc_conv_pack_weights_x3(ptr(w), K, cin, cout, ptr(wp), _lib.stream()))
        check(L.pcc_conv_fwd_x3(ptr(feats), feats.shape[0], cin, ptr(wp), ptr(bias), ptr(nbr), ptr(order), ptr(gmask), K, ptr(out),
                                n_out, cout, 0, None, None, _lib.stream()))
        return out
    check(L.pcc_conv_fwd(ptr(feats), feats.shape[0], cin, ptr(w), ptr(_packed(w)), ptr(bias), ptr(nbr), ptr(order), ptr(gmask), K,
                         ptr(out), n_out, cout, 0, None, None, _lib.stream()))
    return out


def _forward_map(in_map, out_map, ksize, transposed, cin):
    """(nbr, order, gmask) of the forward map; execution order when the MFMA kernels will run on it"""
    if ksize == 1:
        return None, None, None
    if cin % 32 == 0:
        nbr, order, gmask, _ = in_map.ordered_kernel_map(out_map, ksize, transposed)
        return nbr, order, gmask
    nbr, _, _ = in_map.kernel_map(out_map, ksize, transposed)
    return nbr, None, None


def _transposed_map(in_map, out_map, ksize, transposed):
    """kernel map of the backward-data convolution: for input row i and offset k the output row that read i"""
    from .sparse import ORDER_BLOCK_LOG2
    key = ("tmap", id(out_map), ksize, transposed, ORDER_BLOCK_LOG2)
    hit = in_map._cache.get(key)
    if hit is not None and (hit[0] is out_map or (hit[0] is None and out_map is in_map)):
        return hit[1:]
    L = _lib.lib()
    nbr, _, _ = in_map.kernel_map(out_map, ksize, transposed)
    K = nbr.shape[1]
    n_in, n_out = in_map.n, out_map.n
    dev = nbr.device
    nbr_t = torch.empty((n_in, K), dtype=torch.int32, device=dev)
    mask_t = torch.empty(n_in, dtype=torch.int32, device=dev)
    check(L.pcc_kernel_map_transpose(ptr(nbr), n_out, K, n_in, ptr(nbr_t), ptr(mask_t), _lib.stream()))
    order = torch.empty(n_in, dtype=torch.int32, device=dev)
    gmask = torch.empty((n_in + 31) // 32, dtype=torch.int32, device=dev)
    nbytes = L.pcc_order_scratch_bytes(n_in)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(L.pcc_order_rows_by_mask(ptr(mask_t), ptr(in_map.coords), n_in, ORDER_BLOCK_LOG2, in_map.stride, ptr(order),
                                   ptr(gmask), ptr(scratch), nbytes, _lib.stream()))
    res = (nbr_t, order, gmask)
    in_map._cache[key] = (None if out_map is in_map else out_map,) + res
    return res


class SparseConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, kernel, bias, in_map, out_map, ksize, transposed, out_channels):
        feats = feats.contiguous()
        w = kernel.detach()
        if w.dim() == 2:
            w = w.unsqueeze(0)
        b = None if bias is None else bias.detach().reshape(-1)
        if out_channels is not None:
            w = w[:, :, :out_channels]
            b = None if b is None else b[:out_channels].contiguous()
        w = w.contiguous()
        nbr, order, gmask = _forward_map(in_map, out_map, ksize, transposed, feats.shape[1])
        if _bf16_ok(feats.shape[0], feats.shape[1]):
            feats = feats.to(torch.bfloat16)          # the one cast of this tensor: forward now, weight gradient later
        out = _launch_conv(feats, w, b, nbr, order, gmask, out_map.n)
        ctx.save_for_backward(feats, w)
        ctx.meta = (in_map, out_map, ksize, transposed, out_channels, tuple(kernel.shape), bias is not None)
        return out

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        feats, w = ctx.saved_tensors
        in_map, out_map, ksize, transposed, out_channels, kshape, has_bias = ctx.meta
        dy = dy.contiguous()
        K, cin, cout = w.shape
        n_in, n_out = feats.shape[0], dy.shape[0]
        dev = dy.device
        d_feats = d_kernel = d_bias = None
        bf = feats.dtype == torch.bfloat16                 # forward ran in bf16: the saved input is the bf16 copy
        want_db = has_bias and ctx.needs_input_grad[2]
        # dY is read once for its bf16 copy (weight gradient, backward-data) and its column sums (bias gradient)
        dy_bf = db = None
        if (want_db or (bf and ctx.needs_input_grad[1])) and cout % 4 == 0 and cout <= 1024:
            if bf and ctx.needs_input_grad[1]:
                dy_bf = torch.empty((n_out, cout), dtype=torch.bfloat16, device=dev)
            if want_db:
                db = torch.empty(cout, dtype=torch.float32, device=dev)
            ne_cs = L.pcc_cast_colsum_scratch_elems(cout)
            cs_scratch = torch.empty(ne_cs if want_db else 0, dtype=torch.float32, device=dev)
            check(L.pcc_cast_colsum(ptr(dy), n_out, cout, ptr(dy_bf), ptr(db), ptr(cs_scratch) if want_db else None, ne_cs,
                                    _lib.stream()))

        if ctx.needs_input_grad[1]:
            # Thin shapes (q-map branches, input layer, narrow heads) are zero-padded to 32 channels and take the MFMA
            # kernel too: the scalar kernel walks its rows serially and needs 60 ms for 2 -> 128 on 3.4 M rows where
            # the padded MFMA launch takes 7 (16x the multiplications, all of them in the matrix pipe).
            unit = 64 if bf else 32
            cin_p, cout_p = (cin + unit - 1) // unit * unit, (cout + unit - 1) // unit * unit
            x_w = feats
            g_w = (dy_bf if dy_bf is not None else dy.to(torch.bfloat16)) if bf else dy
            if cin_p != cin:
                x_w = torch.cat([x_w, torch.zeros((n_in, cin_p - cin), dtype=x_w.dtype, device=dev)], dim=1)
            if cout_p != cout:
                g_w = torch.cat([g_w, torch.zeros((n_out, cout_p - cout), dtype=g_w.dtype, device=dev)], dim=1)
            if ksize == 1:
                nbr = torch.arange(n_out, dtype=torch.int32, device=dev).unsqueeze(1).contiguous()
                order = gmask = None
            else:
                nbr, order, gmask, _ = in_map.position_ordered_table(out_map, ksize, transposed)      # the wgrad kernels index by position
            dw = torch.empty((K, cin_p, cout_p), dtype=torch.float32, device=dev)
            ne = L.pcc_conv_wgrad_scratch_elems(K, cin_p, cout_p)
            scratch = torch.empty(ne, dtype=torch.float32, device=dev)
            if bf:
                check(L.pcc_conv_wgrad_bf16(ptr(x_w), n_in, cin_p, ptr(g_w), n_out, cout_p, ptr(nbr), ptr(order), ptr(gmask), K, ptr(dw),
                                            ptr(scratch), ne, _lib.stream()))
            else:
                check(L.pcc_conv_wgrad(ptr(x_w), n_in, cin_p, ptr(g_w), n_out, cout_p, ptr(nbr), ptr(order), ptr(gmask), K, ptr(dw),
                                       ptr(scratch), ne, _lib.stream()))
            if cin_p != cin or cout_p != cout:
                dw = dw[:, :cin, :cout].contiguous()
            if out_channels is not None:
                full = torch.zeros((K, cin, kshape[-1]), dtype=torch.float32, device=dev)
                full[:, :, :out_channels] = dw
                dw = full
            d_kernel = dw.reshape(kshape)

        if want_db:
            if db is None:
                db = dy.sum(dim=0)
            if out_channels is not None:
                full = torch.zeros(kshape[-1], dtype=torch.float32, device=dev)
                full[:out_channels] = db
                db = full
            d_bias = db.reshape(1, -1)

        if ctx.needs_input_grad[0]:
            wt = w.transpose(1, 2)                                  # [K, cout, cin]
            g = g_w if (ctx.needs_input_grad[1] and feats.dtype == torch.bfloat16 and cout % 64 == 0 and _bf16_ok(n_out, cout)) else dy
            if cout % 32 and cout not in _THIN_CIN:                 # input widths of the thin forward kernel: _THIN_CIN
                pad = next(c for c in _THIN_CIN if c >= cout) - cout
                g = torch.cat([dy, torch.zeros((n_out, pad), dtype=torch.float32, device=dev)], dim=1).contiguous()
                wt = torch.cat([wt, torch.zeros((K, pad, cin), dtype=torch.float32, device=dev)], dim=1)
            wt = wt.contiguous()
            if ksize == 1:
                d_feats = _launch_conv(g, wt, None, None, None, None, n_in)
            else:
                nbr_t, order_t, gmask_t = _transposed_map(in_map, out_map, ksize, transposed)
                if g.shape[1] % 32 == 0:
                    d_feats = _launch_conv(g, wt, None, nbr_t, order_t, gmask_t, n_in)
                else:
                    d_feats = _launch_conv(g, wt, None, nbr_t, None, None, n_in)
        return d_feats, d_kernel, d_bias, None, None, None, None, None


class EpilogueFn(torch.autograd.Function):
    """out = act(c * beta + gamma) + residual as ONE kernel forward and ONE backward (csrc/epilogue.hip) — the terms the
    inference path fuses into the convolution's epilogue; same operation order, hence the same values and gradients, as
    the chain of torch ops this replaces"""

    @staticmethod
    def forward(ctx, c, film, residual, act):
        c = c.contiguous()
        film = None if film is None else film.contiguous()
        residual = None if residual is None else residual.contiguous()
        n, ch = c.shape
        out = torch.empty_like(c)
        check(_lib.lib().pcc_epilogue_fwd(ptr(c), ptr(film), ptr(residual), n, ch, act, ptr(out), _lib.stream()))
        ctx.save_for_backward(c, film)
        ctx.act = act
        ctx.has_res = residual is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        c, film = ctx.saved_tensors
        dout = dout.contiguous()
        n, ch = c.shape
        dc = torch.empty_like(c)
        dfilm = None if film is None else torch.empty_like(film)
        check(_lib.lib().pcc_epilogue_bwd(ptr(dout), ptr(c), ptr(film), n, ch, ctx.act, ptr(dc), ptr(dfilm), _lib.stream()))
        return dc, dfilm, (dout if ctx.has_res else None), None


def epilogue_train(c, film, residual, act):
    """differentiable epilogue; falls back to torch ops for shapes the kernel does not take (channels % 4 != 0)"""
    ch = c.shape[1]
    if ch % 4 == 0 and (film is None or film.shape[1] == 2 * ch):
        return EpilogueFn.apply(c, film, residual, act)
    if film is not None:
        c = c * film[:, :ch] + film[:, ch:]
    if act == 1:
        c = torch.relu(c)
    elif act == 2:
        c = torch.nn.functional.leaky_relu(c, 0.01)
    return c if residual is None else c + residual


def conv_train(x_feats, in_map, out_map, layer, ksize, transposed, out_channels=None):
    """differentiable out = bias + conv(x) on the HIP kernels"""
    return SparseConvFn.apply(x_feats, layer.kernel, layer.bias, in_map, out_map, ksize, transposed, out_channels)
