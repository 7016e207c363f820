"""bf16-input convolution (pcc_conv_fwd_bf16; BASELINE config 5's precision for training): against the CPU
oracle's convolution evaluated on the SAME bf16-rounded operands in fp32 — products of bf16 values are exact
in fp32, so only the order of the fp32 additions differs."""
import numpy as np
import pytest
import torch

from oracle import coords as oc
from oracle import nn as on

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def shell(grid=36, radius=13.0, thick=0.9):
    g = np.stack(np.meshgrid(*[np.arange(grid)] * 3, indexing="ij"), -1).reshape(-1, 3)
    keep = np.abs(np.linalg.norm(g - (grid - 1) / 2, axis=1) - radius) < thick
    return np.concatenate([np.zeros((int(keep.sum()), 1), np.int32), g[keep].astype(np.int32)], axis=1)


@pytest.mark.parametrize("cin,cout,ksize", [(64, 64, 3), (128, 128, 3), (128, 256, 3), (192, 64, 3), (256, 128, 3), (64, 32, 3),
                                             (128, 3, 3), (128, 128, 1)])
def test_bf16_conv_matches_fp32_on_rounded_operands(pcc, cin, cout, ksize):
    from pcc_amd import _lib
    from pcc_amd._lib import check, ptr
    L = pcc.lib()
    torch.manual_seed(cin + cout)
    c = shell()
    n = c.shape[0]
    K = ksize ** 3
    F = torch.randn(n, cin)
    W = torch.randn(K, cin, cout) / np.sqrt(cin * 10)
    b = torch.randn(cout) * 0.1
    Fb, Wb = F.to(torch.bfloat16), W.to(torch.bfloat16)
    m = pcc.CoordMap(torch.from_numpy(c).to(DEV), 1)
    if ksize == 1:
        nbr = order = gmask = None
        want = Fb.float() @ Wb.float()[0] + b
    else:
        nbr, order, gmask, _ = m.ordered_kernel_map(m, ksize)
        want = on._apply_conv(Fb.float(), Wb.float(), b.reshape(1, -1), oc.kernel_map(c, c, ksize, 1), n)
    wp = torch.empty(L.pcc_conv_packed_elems_bf16(K, cin, cout), dtype=torch.bfloat16, device=DEV)
    # packing rounds the fp32 weights to bf16 itself: feed it the already rounded values (same result)
    check(L.pcc_conv_pack_weights_bf16(ptr(Wb.float().to(DEV).contiguous()), K, cin, cout, ptr(wp), _lib.stream()))
    out = torch.empty((n, cout), dtype=torch.float32, device=DEV)
    x = Fb.to(DEV).contiguous()
    check(L.pcc_conv_fwd_bf16(ptr(x), n, cin, ptr(wp), ptr(b.to(DEV)), ptr(nbr), ptr(order), ptr(gmask), K, ptr(out), n, cout, 0,
                              None, None, _lib.stream()))
    got = out.cpu()
    assert torch.allclose(got, want, rtol=1e-4, atol=2e-5 * float(want.abs().max())), float((got - want).abs().max())
    # and it is close to the unrounded fp32 convolution at bf16 precision
    full = (F @ W[0] + b) if ksize == 1 else on._apply_conv(F, W, b.reshape(1, -1), oc.kernel_map(c, c, ksize, 1), n)
    assert float((got - full).abs().max()) < 3e-2 * float(full.abs().max())


@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 128), (128, 64), (64, 128), (192, 256), (256, 64)])
def test_bf16_weight_gradient_matches_fp32_on_rounded_operands(pcc, cin, cout):
    from pcc_amd import _lib
    from pcc_amd._lib import check, ptr
    L = pcc.lib()
    torch.manual_seed(cin * 3 + cout)
    c = shell()
    n = c.shape[0]
    X, G = torch.randn(n, cin), torch.randn(n, cout)
    Xb, Gb = X.to(torch.bfloat16), G.to(torch.bfloat16)
    m = pcc.CoordMap(torch.from_numpy(c).to(DEV), 1)
    nbr, order, gmask, _ = m.position_ordered_table(m, 3)             # the weight-gradient kernels index their table by position
    dw = torch.empty((27, cin, cout), dtype=torch.float32, device=DEV)
    ne = L.pcc_conv_wgrad_scratch_elems(27, cin, cout)
    scratch = torch.empty(ne, dtype=torch.float32, device=DEV)
    xd, gd = Xb.to(DEV).contiguous(), Gb.to(DEV).contiguous()          # keep the device copies alive across the launch
    check(L.pcc_conv_wgrad_bf16(ptr(xd), n, cin, ptr(gd), n, cout, ptr(nbr), ptr(order), ptr(gmask), 27, ptr(dw), ptr(scratch), ne,
                                _lib.stream()))
    nb = torch.from_numpy(oc.kernel_map(c, c, 3, 1)).long()
    want = torch.zeros(27, cin, cout)
    Xf, Gf = Xb.float(), Gb.float()
    for k in range(27):
        ok = nb[:, k] >= 0
        want[k] = Xf[nb[ok, k]].t() @ Gf[ok]
    got = dw.cpu()
    assert torch.allclose(got, want, rtol=1e-4, atol=2e-5 * float(want.abs().max())), float((got - want).abs().max())
