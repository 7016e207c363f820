"""A small kernel map and its execution order built by ONE launch (csrc/select.hip small_map_kernel, pcc_small_kernel_map)
against the separate launches (pcc_kernel_map + pcc_order_rows_by_mask): every output BIT FOR BIT — neighbour table, row
masks, order, group masks per 32 positions — for stride-1, strided and transposed maps, kernel
sizes 2 and 3, from one row to the 256-row limit (and past it: the one-workgroup ordering of maps up to 16,384 rows against the separate launches); and a frame coded to the same bytes either way."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.as_tensor(a).to(DEV).contiguous()


def _coords(pcc, n, seed=0, scale=1):
    rng = np.random.default_rng(seed)
    p = pcc.synthetic.sphere_shell(64, 27.0, 0.9)[:, :3]
    p = p[np.argsort(((p - p[0]) ** 2).sum(axis=1))][:n]
    c = np.concatenate([np.zeros((p.shape[0], 1)), p * scale], axis=1).astype(np.int32)
    return c[rng.permutation(c.shape[0])]


def _maps(pcc, kind, n, seed):
    """(input map, output map, kernel size, transposed) as the layers build them"""
    if kind == "same":
        m = pcc.CoordMap(dev(_coords(pcc, n, seed)), 1)
        return m, m, 3, False
    if kind == "down":
        m = pcc.CoordMap(dev(_coords(pcc, n, seed)), 1)
        return m, m.down(), 3, False
    m = pcc.CoordMap(dev(_coords(pcc, n, seed, scale=2)), 2)
    ks = int(kind[-1])
    return m, m.up(ks), ks, True


def _both(pcc, kind, n, seed):
    from pcc_amd import sparse as sp
    outs = []
    for cap in (1 << 20, 0):
        was = sp.set_small_map_max(cap)
        try:
            m, o, ks, tr = _maps(pcc, kind, n, seed)
            nbr_o, order, gmask, _ = m.ordered_kernel_map(o, ks, tr)
            nbr, row_mask, pairs = m.kernel_map(o, ks, tr)
            assert nbr_o.data_ptr() == nbr.data_ptr()                   # one table, by output row
            outs.append([t.cpu().numpy() for t in (nbr, row_mask, order, gmask)] + [int(pairs.item())])
        finally:
            sp.set_small_map_max(was)
    return outs


@pytest.mark.parametrize("kind", ["same", "down", "up3", "up2"])
@pytest.mark.parametrize("n", [1, 5, 33, 200, 256])
def test_one_launch_equals_separate_launches(pcc, kind, n):
    if kind in ("up3", "up2"):
        n = min(n, 30 if kind == "up3" else 12)    # the child sets are 8-27 x larger: stay under the limit
    a, b = _both(pcc, kind, n, seed=n)
    if a[0].shape[0] > 256:
        pytest.skip("output set above the one-launch limit")
    names = ("nbr", "row_mask", "order", "group_mask32")
    for name, x, y in zip(names, a, b):
        assert x.shape == y.shape and np.array_equal(x, y), name
    assert a[-1] == b[-1]


@pytest.mark.parametrize("n", [600, 1136, 4904, 9000])
def test_one_workgroup_ordering_equals_separate_launches(pcc, n):
    """maps above the one-launch limit: counts + keys + sort in one workgroup (PCC_ORDER_SMALL=0 in a child process: the
    separate launches)"""
    import os, subprocess, sys, hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, %r)
import pcc_amd as pcc
p = pcc.synthetic.sphere_shell(64, 27.0, 0.9)[:, :3]
p = p[np.argsort(((p - p[0]) ** 2).sum(axis=1))][:%d]
c = np.concatenate([np.zeros((p.shape[0], 1)), p], axis=1).astype(np.int32)
c = torch.from_numpy(c[np.random.default_rng(3).permutation(c.shape[0])]).to("cuda:0")
m = pcc.CoordMap(c, 1)
h = hashlib.sha256()
for t in m.ordered_kernel_map(m, 3)[:3]:
    h.update(t.cpu().numpy().tobytes())
print("digest", h.hexdigest())
""" % (root, n)
    digests = []
    for v in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, PCC_ORDER_SMALL=v), capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0 and "digest" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        digests.append(r.stdout.split("digest")[-1].strip())
    assert digests[0] == digests[1]


@pytest.mark.parametrize("n", [1, 700, 20000, 32768])
def test_one_workgroup_topk_equals_separate_launches(pcc, n):
    """pcc_topk_mask for one batch item and at most 32,768 rows runs in one workgroup (csrc/select.hip topk_small_kernel);
    PCC_TOPK_SMALL=0 in a child process = the 26 separate launches.  Ties (equal logits, -0.0), k = 0, 1, n/3, n, n + 5."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, %r)
import pcc_amd as pcc
from pcc_amd import sparse as sp
n = %d
rng = np.random.default_rng(n)
flat = rng.choice(64 ** 3, size=n, replace=False)
c = np.stack([np.zeros(n, np.int64), flat // 4096 - 20, (flat // 64) %% 64 - 3, flat %% 64], axis=1).astype(np.int32)
logits = rng.normal(size=(n, 3)).astype(np.float32)
logits[rng.integers(0, n, n // 3 + 1), 0] = 0.25
logits[rng.integers(0, n, n // 20 + 1), 0] = -0.0
h = hashlib.sha256()
for k in (0, 1, n // 3, n, n + 5):
    m = sp.topk_mask(torch.from_numpy(logits).to("cuda:0"), torch.from_numpy(c).to("cuda:0"), [k], 1).cpu().numpy()
    assert int(m.sum()) == min(k, n), (k, int(m.sum()))
    h.update(m.tobytes())
print("digest", h.hexdigest())
""" % (root, n)
    digests = []
    for v in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, PCC_TOPK_SMALL=v), capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0 and "digest" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        digests.append(r.stdout.split("digest")[-1].strip())
    assert digests[0] == digests[1]


@pytest.mark.parametrize("n", [1, 40, 300, 1000, 8000])
def test_one_workgroup_unique_equals_separate_launches(pcc, n):
    """coordinate sets of at most 8,192 candidates (strided parents, children of kernel 2 and 3) are built by one workgroup
    (csrc/coords.hip unique_small_kernel); PCC_UNIQUE_SMALL=0 in a child process = the seven separate launches.  Same rows in
    the same order, and the same hash table (probed through a kernel map)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, %r)
import pcc_amd as pcc
n = %d
p = pcc.synthetic.sphere_shell(64, 27.0, 0.9)[:, :3]
p = p[np.argsort(((p - p[0]) ** 2).sum(axis=1))][:n]
c = np.concatenate([np.zeros((p.shape[0], 1)), p], axis=1).astype(np.int32)
c = torch.from_numpy(c[np.random.default_rng(5).permutation(c.shape[0])]).to("cuda:0")
h = hashlib.sha256()
m = pcc.CoordMap(c, 1)
d = m.down()
h.update(d.coords.cpu().numpy().tobytes())
h.update(m.kernel_map(d, 3)[0].cpu().numpy().tobytes())
for ks in (2, 3):
    if d.n * ks ** 3 > 8192 and n > 300:
        continue
    u = d.up(ks)
    h.update(u.coords.cpu().numpy().tobytes())
    h.update(d.kernel_map(u, ks, True)[0].cpu().numpy().tobytes())
    h.update(u.kernel_map(u, 3)[0].cpu().numpy().tobytes())
print("digest", h.hexdigest())
""" % (root, n)
    digests = []
    for v in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, PCC_UNIQUE_SMALL=v), capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0 and "digest" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        digests.append(r.stdout.split("digest")[-1].strip())
    assert digests[0] == digests[1]


def test_runtime_switch_of_the_one_workgroup_paths(pcc):
    """pcc_small_paths: the mask reads back, and a frame codes to the same bytes with every path off and on in one process
    (what bench.py's `small_frame` record does)"""
    from pcc_amd import sparse as sp
    was = sp.set_small_paths(-1)
    assert 0 <= was <= 7
    assert sp.set_small_paths(5) == was and sp.set_small_paths(-1) == 5 and sp.set_small_paths(was) == 5
    syn = pcc.synthetic
    model = syn.make_model(0, DEV)
    model.update()
    pts = syn.sphere_shell(**syn.CONFIG1)
    qc, qf = syn.uniform_qmap(pts[:, :3], 0.5, 0.5)

    def run():
        Q = pcc.SparseTensor(coordinates=dev(qc), features=dev(qf), device=DEV)
        strings, shape, k, coords = model.compress(dev(pts), Q)
        return strings, shape, k, model.decompress(coordinates=coords, strings=strings, shape=shape, k=k)

    state = (sp.set_conv_small_max(0), sp.set_small_map_max(0), sp.set_small_paths(0))
    try:
        s0, sh0, k0, r0 = run()
    finally:
        sp.set_conv_small_max(state[0]); sp.set_small_map_max(state[1]); sp.set_small_paths(state[2])
    s1, sh1, k1, r1 = run()
    assert s0 == s1 and sh0 == sh1 and k0 == k1 and torch.equal(r0, r1)


def test_limit_and_switch(pcc):
    from pcc_amd import sparse as sp
    cap = sp._small_map_max()
    assert cap in (0, 256)
    was = sp.set_small_map_max(0)
    assert sp._small_map_max() == 0
    sp.set_small_map_max(was)
    assert sp._small_map_max() == was


def test_frame_codes_to_the_same_bytes(pcc):
    from pcc_amd import sparse as sp
    syn = pcc.synthetic
    model = syn.make_model(0, DEV)
    model.update()
    pts = syn.sphere_shell(**syn.CONFIG1)
    qc, qf = syn.uniform_qmap(pts[:, :3], 0.5, 0.5)

    def run():
        Q = pcc.SparseTensor(coordinates=dev(qc), features=dev(qf), device=DEV)
        strings, shape, k, coords = model.compress(dev(pts), Q)
        return strings, shape, k, model.decompress(coordinates=coords, strings=strings, shape=shape, k=k)

    was = sp.set_small_map_max(0)
    try:
        s0, sh0, k0, r0 = run()
        sp.set_small_map_max(1 << 20)
        s1, sh1, k1, r1 = run()
    finally:
        sp.set_small_map_max(was)
    assert s0 == s1 and sh0 == sh1 and k0 == k1 and torch.equal(r0, r1)
