"""Factorized entropy bottleneck and Gaussian conditional on libpcc_hip.so.

Host-side mirror of ``compressai.entropy_models.EntropyBottleneck`` / ``GaussianConditional``
as the reference uses them (model/entropy_models.py:269-270,313,330,352-353,371-372,393,407-408):
same constructor arguments, parameter / buffer names (so reference state_dicts load), and the
``compress`` / ``decompress`` / ``forward`` / ``update`` / ``loss`` methods.  Quantisation, index
build and likelihoods run as HIP kernels on [N, C] feature matrices; CDF tables are built on the
host at ``update()`` (one-off) and the rANS coder is the host C++ one in the same library
(SURVEY.md N12-N14).  Training mode — additive U(-.5, .5) noise instead of rounding, likelihoods differentiable through torch
ops, compressai's LowerBound gradients — is ``forward(..., training=True)`` / ``module.train()``.
"""
import math

import threading

import numpy as np
import scipy.stats
import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr

SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256.0, 64


def get_scale_table(minimum=SCALES_MIN, maximum=SCALES_MAX, levels=SCALES_LEVELS):
    return torch.exp(torch.linspace(math.log(minimum), math.log(maximum), levels))


class _LowerBoundFn(torch.autograd.Function):
    """max(x, bound) whose gradient also flows where x < bound if it pushes x up (compressai.ops.LowerBound)."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, grad):
        x, bound = ctx.saved_tensors
        keep = (x >= bound) | (grad < 0)
        return keep.to(grad.dtype) * grad, None


class _LowerBound(nn.Module):
    """compressai.ops.LowerBound (same state_dict entry)."""

    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.tensor([float(bound)]))

    def forward(self, x):
        return _LowerBoundFn.apply(x, self.bound.to(x.device))


# Training-mode quantisation adds U(-0.5, 0.5) noise (compressai EntropyModel.quantize, mode "noise").  Tests
# replace the source to feed the CPU oracle the same draws: a callable (shape, coords) -> CPU float tensor,
# where coords are the int32 [N, 4] coordinates of the tensor's points (NOISE_ROWS, set by the caller of the
# entropy model) — row order differs between implementations, coordinates do not.
NOISE_SOURCE = None
NOISE_ROWS = None


def _uniform_noise(like):
    if NOISE_SOURCE is not None:
        rows = None if NOISE_ROWS is None else NOISE_ROWS.detach().cpu().numpy()
        return NOISE_SOURCE(tuple(like.shape), rows).to(like.device, like.dtype)
    return torch.empty_like(like).uniform_(-0.5, 0.5)


def _pmf_to_cdf(pmf, tail, pmf_length, max_length):
    """compressai EntropyModel._pmf_to_cdf through the library's pmf_to_quantized_cdf."""
    L = _lib.lib()
    cdf = np.zeros((len(pmf_length), max_length + 2), dtype=np.int32)
    for i in range(len(pmf_length)):
        n = int(pmf_length[i])
        prob = np.ascontiguousarray(np.concatenate([pmf[i, :n], tail[i].reshape(-1)]).astype(np.float32))
        row = np.zeros(n + 2, dtype=np.int32)
        check(L.pcc_pmf_to_quantized_cdf(ptr(prob), n + 1, 16, ptr(row)))
        cdf[i, : n + 2] = row
    return cdf


# Pinned staging buffers for the symbol / index planes (the only per-frame host traffic of the
# path): page-locked copies run at PCIe rate instead of through a pageable bounce buffer.
_PINNED = {}


_PIN_BUSY = {}      # (thread, key) -> event recorded behind the last asynchronous upload FROM that staging buffer


def _pinned(key, numel, dtype):
    key = (threading.get_ident(), key)         # one staging set per worker thread (streamed sequences: one thread per frame in flight)
    ev = _PIN_BUSY.pop(key, None)
    if ev is not None:
        ev.synchronize()                       # the previous upload from this buffer (long done by now) before it is rewritten
    buf = _PINNED.get(key)
    if buf is None or buf.numel() < numel or buf.dtype != dtype:
        buf = torch.empty(max(numel, 1), dtype=dtype, pin_memory=True)
        _PINNED[key] = buf
    return buf[:numel]


def _to_host(t, key):
    """Device tensor -> numpy view of a pinned host buffer (valid until the next call with ``key``)."""
    t = t.contiguous()
    buf = _pinned(key, t.numel(), t.dtype)
    buf.copy_(t.reshape(-1), non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return buf.numpy().reshape(t.shape)


def _to_host_async(t, key):
    """Like _to_host without the wait: -> (numpy view, event); the view is valid after ``event.synchronize()``."""
    t = t.contiguous()
    buf = _pinned(key, t.numel(), t.dtype)
    buf.copy_(t.reshape(-1), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(t.device))
    return buf.numpy().reshape(t.shape), ev


def _upload_guard(key, device):
    """the staging buffer ``key`` of this thread has an upload in flight on the current stream: its next user waits for THAT (an event),
    not the uploader for the whole stream — a stream wait here stopped the coding thread until the GPU had drained everything queued
    (the kernel maps in front of h_s, the prefetched maps in front of g_s) and the chip then idled while the thread caught up"""
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    _PIN_BUSY[(threading.get_ident(), key)] = ev


def _to_device(arr, device, key):
    """numpy array -> device tensor through a pinned staging buffer."""
    src = torch.from_numpy(np.ascontiguousarray(arr))
    buf = _pinned(key, src.numel(), src.dtype)
    buf.copy_(src.reshape(-1))
    out = buf.to(device, non_blocking=True).reshape(src.shape)
    _upload_guard(key, out.device)
    return out


class _HostWorker:
    """A persistent thread per coding thread for the serial range-coder calls that run beside GPU work (the C calls release the
    interpreter lock).  Creating a thread per call cost the coding thread ~0.3 ms at the start of every decode (thread start + the new
    thread's Python prelude under the lock) — on the critical path, with the GPU idle."""

    def __init__(self):
        import queue
        self.jobs = queue.SimpleQueue()
        self.thread = threading.Thread(target=self._run, name="pcc-rans", daemon=True)
        self.thread.start()

    def _run(self):
        while True:
            fn, done = self.jobs.get()
            try:
                fn()
            except BaseException as e:      # re-raised on the coding thread by wait()
                done.err = e
            done.set()

    def submit(self, fn):
        done = threading.Event()
        done.err = None
        self.jobs.put((fn, done))
        return done


_WORKERS = {}


def _host_worker():
    key = threading.get_ident()
    w = _WORKERS.get(key)
    if w is None:
        w = _WORKERS[key] = _HostWorker()
    return w


def _wait(done):
    done.wait()
    if done.err is not None:
        raise done.err


_CHANNEL_INDEX_PLANES = {}


def _channel_index_plane(c, n):
    """np.repeat(np.arange(c), n) as int32 (the factorized model's indexes: channel-major planes), cached: 150 k elements per decode"""
    key = (c, n)
    hit = _CHANNEL_INDEX_PLANES.get(key)
    if hit is None:
        if len(_CHANNEL_INDEX_PLANES) > 64:
            _CHANNEL_INDEX_PLANES.clear()
        hit = _CHANNEL_INDEX_PLANES[key] = np.repeat(np.arange(c, dtype=np.int32), n)
    return hit


def _rans_encode(symbols, indexes, cdf, cdf_length, offset):
    """symbols / indexes: host int32 arrays (flattened channel-major)."""
    L = _lib.lib()
    symbols = np.ascontiguousarray(symbols, dtype=np.int32).reshape(-1)
    indexes = np.ascontiguousarray(indexes, dtype=np.int32).reshape(-1)
    n = symbols.size
    # typical streams need well under 1 byte per symbol; fall back to the worst-case bound if not
    for cap in (n + 4096, 4 * (3 * n + 4)):
        out = np.empty(cap, dtype=np.uint8)
        nbytes = L.pcc_rans_encode_with_indexes(ptr(symbols), ptr(indexes), n, ptr(cdf), cdf.shape[1],
                                                ptr(cdf_length), ptr(offset), ptr(out), cap)
        if nbytes >= 0:
            return out[:nbytes].tobytes()
    check(nbytes)


def _rans_encode_packed(symbols, indexes, cdf, cdf_length, offset):
    """symbols int16 / indexes uint8 host arrays (the planes pcc_gc_encode_prep_packed wrote): identical bytes"""
    L = _lib.lib()
    n = symbols.size
    for cap in (n + 4096, 4 * (3 * n + 4)):
        out = np.empty(cap, dtype=np.uint8)
        nbytes = L.pcc_rans_encode_with_indexes_i16u8(ptr(symbols), ptr(indexes), n, ptr(cdf), cdf.shape[1],
                                                      ptr(cdf_length), ptr(offset), ptr(out), cap)
        if nbytes >= 0:
            return out[:nbytes].tobytes()
    check(nbytes)


def _rans_decode_packed(data, indexes, cdf, cdf_length, offset, out):
    """indexes uint8 -> symbols into ``out`` (int16, pinned); returns False when a symbol did not fit int16"""
    L = _lib.lib()
    buf = np.frombuffer(data, dtype=np.uint8)
    narrowed = np.zeros(1, dtype=np.int32)
    check(L.pcc_rans_decode_with_indexes_u8i16(ptr(buf), len(data), ptr(indexes), indexes.size, ptr(cdf), cdf.shape[1],
                                               ptr(cdf_length), ptr(offset), ptr(out), ptr(narrowed)))
    return int(narrowed[0]) == 0


def _rans_decode(data, indexes, cdf, cdf_length, offset):
    L = _lib.lib()
    indexes = np.ascontiguousarray(indexes, dtype=np.int32).reshape(-1)
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(indexes.size, dtype=np.int32)
    check(L.pcc_rans_decode_with_indexes(ptr(buf), len(data), ptr(indexes), indexes.size, ptr(cdf), cdf.shape[1],
                                         ptr(cdf_length), ptr(offset), ptr(out)))
    return out


class _EntropyModelBase(nn.Module):
    def __init__(self, likelihood_bound=1e-9, entropy_coder_precision=16):
        super().__init__()
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.likelihood_lower_bound = _LowerBound(likelihood_bound)
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())
        self._host_tables = None

    def _set_tables(self, cdf, cdf_length, offset):
        dev = self._offset.device
        self._quantized_cdf = torch.from_numpy(cdf).to(dev)
        self._cdf_length = torch.from_numpy(cdf_length).to(dev)
        self._offset = torch.from_numpy(offset).to(dev)
        self._host_tables = (np.ascontiguousarray(cdf), np.ascontiguousarray(cdf_length), np.ascontiguousarray(offset))

    def tables(self):
        if self._host_tables is None:
            if self._offset.numel() == 0:
                raise RuntimeError("entropy tables missing: call update() before compress/decompress "
                                   "(reference: evaluate.py:80-84)")
            self._host_tables = (np.ascontiguousarray(self._quantized_cdf.cpu().numpy().astype(np.int32)),
                                 np.ascontiguousarray(self._cdf_length.cpu().numpy().astype(np.int32)),
                                 np.ascontiguousarray(self._offset.cpu().numpy().astype(np.int32)))
        return self._host_tables

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # table buffers change size between a fresh module and an updated checkpoint
        for name in ("_offset", "_quantized_cdf", "_cdf_length", "scale_table"):
            key = prefix + name
            if key in state_dict and hasattr(self, name):
                setattr(self, name, state_dict[key].clone().to(getattr(self, name).device))
        self._host_tables = None
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class EntropyBottleneck(_EntropyModelBase):
    """compressai EntropyBottleneck(channels, tail_mass=1e-9, init_scale=10, filters=(3,3,3,3))."""

    def __init__(self, channels, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3)):
        super().__init__()
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        assert self.filters == (3, 3, 3, 3), "the HIP likelihood kernel is specialised for filters (3,3,3,3)"
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        f = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / f[i + 1]))
            self.register_parameter(f"_matrix{i}", nn.Parameter(torch.full((channels, f[i + 1], f[i]), float(init))))
            self.register_parameter(f"_bias{i}", nn.Parameter(torch.empty(channels, f[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self.register_parameter(f"_factor{i}", nn.Parameter(torch.zeros(channels, f[i + 1], 1)))
        self.quantiles = nn.Parameter(torch.tensor([-self.init_scale, 0.0, self.init_scale]).repeat(channels, 1, 1))
        target = np.log(2 / self.tail_mass - 1)
        self.register_buffer("target", torch.tensor([-target, 0.0, target], dtype=torch.float32))

    # -- host math (used by update() and loss()) ---------------------------------------------
    def _density_params(self, device=None):
        get = lambda n: getattr(self, n) if device is None else getattr(self, n).detach().float().to(device)
        return ([get(f"_matrix{i}") for i in range(5)], [get(f"_bias{i}") for i in range(5)],
                [get(f"_factor{i}") for i in range(4)])

    @staticmethod
    def _logits_cumulative(params, v):
        mats, biases, factors = params
        for i in range(5):
            v = torch.matmul(torch.nn.functional.softplus(mats[i]), v) + biases[i]
            if i < 4:
                v = v + torch.tanh(factors[i]) * torch.tanh(v)
        return v

    def loss(self):
        """aux loss (reference: model/model.py:40-47, train.py:209)."""
        # compressai evaluates the density with stop_gradient=True here: only ``quantiles`` receive a gradient
        mats, biases, factors = self._density_params()
        frozen = ([m.detach() for m in mats], [b.detach() for b in biases], [f.detach() for f in factors])
        logits = self._logits_cumulative(frozen, self.quantiles)
        return torch.abs(logits - self.target).sum()

    def medians(self):
        q = self.quantiles
        key = (q._version, q.data_ptr())
        hit = self.__dict__.get("_medians_cache")
        if hit is None or hit[0] != key:
            hit = (key, q[:, 0, 1].detach().contiguous())          # (a strided slice: one small copy kernel per parameter version)
            self.__dict__["_medians_cache"] = hit
        return hit[1]

    @torch.no_grad()
    def update(self, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        q = self.quantiles.detach().float().cpu()
        cpu = self._density_params("cpu")   # host copy of the density parameters
        med = q[:, 0, 1]
        minima = torch.clamp(torch.ceil(med - q[:, 0, 0]).int(), min=0)
        maxima = torch.clamp(torch.ceil(q[:, 0, 2] - med).int(), min=0)
        pmf_start = med - minima
        pmf_length = maxima + minima + 1
        max_length = int(pmf_length.max())
        samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
        lo = self._logits_cumulative(cpu, samples - 0.5)
        up = self._logits_cumulative(cpu, samples + 0.5)
        s = -torch.sign(lo + up)
        pmf = torch.abs(torch.sigmoid(s * up) - torch.sigmoid(s * lo))[:, 0, :]
        tail = torch.sigmoid(lo[:, 0, :1]) + torch.sigmoid(-up[:, 0, -1:])
        cdf = _pmf_to_cdf(pmf.numpy(), tail.numpy(), pmf_length.numpy(), max_length)
        self._set_tables(cdf, (pmf_length + 2).numpy().astype(np.int32), (-minima).numpy().astype(np.int32))
        return True

    # -- device side -----------------------------------------------------------------------------
    def _kernel_params(self):
        """[C, 58] = softplus(matrix_i) | bias_i | tanh(factor_i) flattened (see pcc_eb_likelihood)."""
        parts = []
        C = self.channels
        for i in range(5):
            parts.append(torch.nn.functional.softplus(getattr(self, f"_matrix{i}").detach()).reshape(C, -1))
            parts.append(getattr(self, f"_bias{i}").detach().reshape(C, -1))
            if i < 4:
                parts.append(torch.tanh(getattr(self, f"_factor{i}").detach()).reshape(C, -1))
        out = torch.cat(parts, dim=1).contiguous().float()
        assert out.shape[1] == 58
        return out

    def quantize_features(self, z_feats, want_symbols=True, want_zhat=True):
        """z_feats [N, C] -> (symbols int32 [C, N] | None, z_hat [N, C] | None)."""
        n, c = z_feats.shape
        dev = z_feats.device
        sym = torch.empty((c, n), dtype=torch.int32, device=dev) if want_symbols else None
        zhat = torch.empty((n, c), dtype=torch.float32, device=dev) if want_zhat else None
        check(_lib.lib().pcc_eb_quantize(ptr(z_feats.contiguous()), n, c, ptr(self.medians()), ptr(sym), ptr(zhat),
                                         _lib.stream()))
        return sym, zhat

    def likelihood_features(self, zhat_feats):
        """z_hat [N, C] -> likelihood plane [C, N]."""
        n, c = zhat_feats.shape
        lik = torch.empty((c, n), dtype=torch.float32, device=zhat_feats.device)
        check(_lib.lib().pcc_eb_likelihood(ptr(zhat_feats.contiguous()), n, c, ptr(self._kernel_params()), ptr(lik),
                                           _lib.stream()))
        return lik

    def _likelihood_train(self, v):
        """differentiable likelihood of v [C, 1, N] (compressai EntropyBottleneck._likelihood); elementwise
        torch ops: the training path needs gradients into the density parameters"""
        params = self._density_params()
        lower = self._logits_cumulative(params, v - 0.5)
        upper = self._logits_cumulative(params, v + 0.5)
        sign = -torch.sign(lower + upper).detach()
        return torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))

    def forward(self, x, training=None):
        """x: (1, C, N) as in the reference (``z.F.t().unsqueeze(0)``) -> (x_hat, likelihoods), both (1, C, N).
        Training mode: additive uniform noise instead of rounding, likelihoods with gradients."""
        if self.training if training is None else training:
            v = x.permute(1, 0, 2)                                   # [C, 1, N]
            v = v + _uniform_noise(v)
            lik = self.likelihood_lower_bound(self._likelihood_train(v))
            return v.permute(1, 0, 2), lik.permute(1, 0, 2)
        feats = x[0].t().contiguous()
        _, zhat = self.quantize_features(feats, want_symbols=False)
        lik = self.likelihood_features(zhat)
        return zhat.t().unsqueeze(0), lik.unsqueeze(0)

    def compress_features(self, z_feats, perm=None):
        """-> ([bytes], z_hat [N, C]); symbol order = channel-major over rows (optionally permuted)."""
        sym, zhat = self.quantize_features(z_feats)
        if perm is not None:
            sym = sym.index_select(1, perm.long())
        finish, _ = self._encode_begin(sym)
        return finish(), zhat

    def _encode_begin(self, sym):
        cdf, cdf_len, off = self.tables()
        c, n = sym.shape
        host, ev = _to_host_async(sym, "eb_sym")

        def finish():
            ev.synchronize()
            return [_rans_encode(host, np.repeat(np.arange(c, dtype=np.int32), n), cdf, cdf_len, off)]

        return finish, ev

    def compress_features_begin(self, z_feats, perm=None):
        """Two-phase form of compress_features: quantisation and the copy to the host are enqueued now; the returned
        function waits for the copy and runs the (serial, host) range coder — in between the caller can enqueue the
        GPU work that only needs z_hat (h_s, the preparation of the y symbols).  -> (finish() -> [bytes], z_hat)"""
        sym, zhat = self.quantize_features(z_feats)
        if perm is not None:
            sym = sym.index_select(1, perm.long())
        finish, _ = self._encode_begin(sym)
        return finish, zhat

    def compress(self, x):
        strings, _ = self.compress_features(x[0].t().contiguous())
        return strings

    def decompress_features(self, strings, n, device):
        """-> z_hat [N, C] on ``device``."""
        cdf, cdf_len, off = self.tables()
        c = self.channels
        idx = np.repeat(np.arange(c, dtype=np.int32), n)
        sym = torch.from_numpy(_rans_decode(strings[0], idx, cdf, cdf_len, off).reshape(c, n)).to(device)
        zhat = torch.empty((n, c), dtype=torch.float32, device=device)
        check(_lib.lib().pcc_eb_dequantize(ptr(sym), n, c, ptr(self.medians().to(device)), ptr(zhat), _lib.stream()))
        return zhat

    def decompress_features_async(self, strings, n, device):
        """Start the (host, serial) range decode of z on a worker thread — it needs nothing from the GPU — and return a
        function that joins it and dequantises: in between the caller builds the coordinate sets and kernel maps that
        depend on coordinates only."""
        cdf, cdf_len, off = self.tables()
        c = self.channels
        box = {}
        indexes = _channel_index_plane(c, n)
        # everything the call needs is prepared HERE: the worker's job is the C call alone (it releases the interpreter lock; Python
        # steps on the worker would each queue for the lock this thread holds while it builds the kernel maps)
        buf = np.frombuffer(strings[0], dtype=np.uint8)
        sym_host = _pinned("eb_sym32", c * n, torch.int32)          # decoded straight into the page-locked plane the upload reads
        out = sym_host.numpy()
        fn = _lib.lib().pcc_rans_decode_with_indexes
        args = (ptr(buf), len(strings[0]), ptr(indexes), indexes.size, ptr(cdf), cdf.shape[1], ptr(cdf_len), ptr(off), ptr(out))

        def work():
            box["rc"] = fn(*args)

        done = _host_worker().submit(work)

        def finish():
            _wait(done)
            check(box["rc"])
            sym = sym_host.to(device, non_blocking=True).reshape(c, n)
            _upload_guard("eb_sym32", device)
            zhat = torch.empty((n, c), dtype=torch.float32, device=device)
            check(_lib.lib().pcc_eb_dequantize(ptr(sym), n, c, ptr(self.medians().to(device)), ptr(zhat), _lib.stream()))
            return zhat

        return finish

    def decompress(self, strings, size):
        n = int(size[0])
        return self.decompress_features(strings, n, self.quantiles.device).t().unsqueeze(0)


class GaussianConditional(_EntropyModelBase):
    """compressai GaussianConditional(scale_table, scale_bound=0.11, tail_mass=1e-9)."""

    def __init__(self, scale_table=None, scale_bound=0.11, tail_mass=1e-9):
        super().__init__()
        self.tail_mass = float(tail_mass)
        self.lower_bound_scale = _LowerBound(scale_bound)
        self.register_buffer("scale_table",
                             torch.tensor([float(s) for s in scale_table]) if scale_table is not None else torch.Tensor())

    @torch.no_grad()
    def update_scale_table(self, scale_table, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        self.scale_table = torch.tensor([float(s) for s in scale_table], device=self.scale_table.device)
        self.update()
        return True

    @torch.no_grad()
    def update(self):
        table = self.scale_table.detach().float().cpu()
        mult = -scipy.stats.norm.ppf(self.tail_mass / 2)
        center = torch.ceil(table * mult).int()
        pmf_length = 2 * center + 1
        max_length = int(pmf_length.max())
        samples = torch.abs(torch.arange(max_length).int() - center[:, None]).float()
        s = table[:, None]
        Phi = lambda v: 0.5 * torch.erfc(-(2 ** -0.5) * v)
        upper = Phi((0.5 - samples) / s)
        lower = Phi((-0.5 - samples) / s)
        pmf = upper - lower
        tail = 2 * lower[:, :1]
        cdf = _pmf_to_cdf(pmf.numpy(), tail.numpy(), pmf_length.numpy(), max_length)
        self._set_tables(cdf, (pmf_length + 2).numpy().astype(np.int32), (-center).numpy().astype(np.int32))

    # -- device side, [N, C] features; params [N, 2C] = (scales | means) aligned with y ------------
    def encode_prep(self, y_feats, params):
        n, c = y_feats.shape
        dev = y_feats.device
        sym = torch.empty((c, n), dtype=torch.int32, device=dev)
        idx = torch.empty((c, n), dtype=torch.int32, device=dev)
        table = self.scale_table.to(dev).contiguous()
        check(_lib.lib().pcc_gc_encode_prep(ptr(y_feats.contiguous()), ptr(params.contiguous()), n, c, ptr(table),
                                            table.numel(), ptr(sym), ptr(idx), _lib.stream()))
        return sym, idx

    def indexes_for(self, params, c):
        n = params.shape[0]
        dev = params.device
        idx = torch.empty((c, n), dtype=torch.int32, device=dev)
        table = self.scale_table.to(dev).contiguous()
        check(_lib.lib().pcc_gc_encode_prep(None, ptr(params.contiguous()), n, c, ptr(table), table.numel(), None,
                                            ptr(idx), _lib.stream()))
        return idx

    def compress_features(self, y_feats, params, perm=None):
        sym, idx = self.encode_prep(y_feats, params)
        if perm is not None:
            p = perm.long()
            sym, idx = sym.index_select(1, p), idx.index_select(1, p)
        return self._encode_begin(sym, idx)()

    def _encode_begin(self, sym, idx):
        cdf, cdf_len, off = self.tables()
        both, ev = _to_host_async(torch.stack([sym, idx]), "gc_sym_idx")

        def finish():
            ev.synchronize()
            return [_rans_encode(both[0], both[1], cdf, cdf_len, off)]

        return finish

    def compress_features_begin(self, y_feats, params, perm=None):
        """Two-phase form of compress_features (see EntropyBottleneck.compress_features_begin) -> finish() -> [bytes].
        One kernel writes the int16 symbol plane and the uint8 index plane in stream order (rows permuted on the way)
        into one buffer, one copy takes it to the host: 3 bytes per symbol instead of 8.  Symbols beyond int16 (legal,
        escape-coded, not seen with sane scales) fall back to the int32 planes."""
        n, c = y_feats.shape
        dev = y_feats.device
        cn = c * n
        flag_at = (3 * cn + 3) // 4 * 4
        packed = torch.empty(flag_at + 4, dtype=torch.uint8, device=dev)
        table = self.scale_table.to(dev).contiguous()
        base = packed.data_ptr()
        check(_lib.lib().pcc_gc_encode_prep_packed(ptr(y_feats.contiguous()), ptr(params.contiguous()), n, c, ptr(table), table.numel(),
                                                   ptr(None if perm is None else perm.contiguous()), base, base + 2 * cn, base + flag_at,
                                                   _lib.stream()))
        host, ev = _to_host_async(packed, "gc_packed")
        cdf, cdf_len, off = self.tables()

        def finish():
            ev.synchronize()
            if int(host[flag_at:flag_at + 4].view(np.int32)[0]):
                return self.compress_features(y_feats, params, perm)             # int32 planes
            return [_rans_encode_packed(host[:2 * cn].view(np.int16), host[2 * cn:3 * cn], cdf, cdf_len, off)]

        return finish

    def decompress_features(self, strings, params, c):
        """params rows must be in the bitstream's row order.  -> y_hat [N, C]."""
        return self.decompress_features_async(strings, params, c)()

    def decompress_features_async(self, strings, params, c):
        """Start decoding y: indexes are built on the GPU and copied to the host, then the serial rANS
        decode runs on a worker thread (the C call releases the GIL).  Returns a function that joins the
        thread, uploads the symbols and returns y_hat [N, C]; in between the caller can enqueue GPU
        work that does not depend on y (h_q, kernel maps of the first synthesis stage)."""
        import threading
        n = params.shape[0]
        dev = params.device
        cn = c * n
        idx_dev = torch.empty(cn, dtype=torch.uint8, device=dev)
        table = self.scale_table.to(dev).contiguous()
        check(_lib.lib().pcc_gc_encode_prep_packed(None, ptr(params.contiguous()), n, c, ptr(table), table.numel(), None, None,
                                                   ptr(idx_dev), None, _lib.stream()))
        idx_host, idx_ev = _to_host_async(idx_dev, "gc_idx8")      # the worker waits for the copy; this thread goes on enqueueing h_q
        sym_host = _pinned("gc_sym16", cn, torch.int16)
        cdf, cdf_len, off = self.tables()
        box = {}

        # (arguments prepared here, the worker's job is the event wait and the C call: see the factorized model's decode)
        buf = np.frombuffer(strings[0], dtype=np.uint8)
        narrowed = np.zeros(1, dtype=np.int32)
        sym_np = sym_host.numpy()
        fn = _lib.lib().pcc_rans_decode_with_indexes_u8i16
        args = (ptr(buf), len(strings[0]), ptr(idx_host), idx_host.size, ptr(cdf), cdf.shape[1], ptr(cdf_len), ptr(off), ptr(sym_np), ptr(narrowed))

        def work():
            idx_ev.synchronize()
            check(fn(*args))
            box["fits"] = int(narrowed[0]) == 0
            if not box["fits"]:                                              # a symbol beyond int16: int32 planes
                box["sym"] = _rans_decode(strings[0], idx_host.astype(np.int32), cdf, cdf_len, off)

        done = _host_worker().submit(work)

        def finish():
            _wait(done)
            if "err" in box:
                raise box["err"]
            yhat = torch.empty((n, c), dtype=torch.float32, device=dev)
            if box["fits"]:
                sym = sym_host.to(dev, non_blocking=True)
                _upload_guard("gc_sym16", dev)                                   # the pinned plane is reused by the next frame
                check(_lib.lib().pcc_gc_dequantize_i16(ptr(sym), ptr(params.contiguous()), n, c, ptr(yhat), _lib.stream()))
            else:
                sym = _to_device(box["sym"].reshape(c, n), dev, "gc_sym_up")
                check(_lib.lib().pcc_gc_dequantize(ptr(sym), ptr(params.contiguous()), n, c, ptr(yhat), _lib.stream()))
            return yhat

        return finish

    def forward_features(self, y_feats, params):
        n, c = y_feats.shape
        dev = y_feats.device
        yhat = torch.empty((n, c), dtype=torch.float32, device=dev)
        lik = torch.empty((c, n), dtype=torch.float32, device=dev)
        check(_lib.lib().pcc_gc_forward(ptr(y_feats.contiguous()), ptr(params.contiguous()), n, c, ptr(yhat), ptr(lik),
                                        _lib.stream()))
        return yhat, lik

    # reference-shaped wrappers ((1, C, N) tensors) ------------------------------------------------
    def build_indexes(self, scales):
        c, n = scales.shape[1], scales.shape[2]
        params = torch.cat([scales[0].t(), torch.zeros((n, c), device=scales.device)], dim=1).contiguous()
        return self.indexes_for(params, c).unsqueeze(0)

    def forward(self, inputs, scales, means=None, training=None):
        if self.training if training is None else training:
            # compressai GaussianConditional.forward, mode "noise": y + U(-.5,.5); likelihood of the noisy value
            outputs = inputs + _uniform_noise(inputs)
            values = outputs if means is None else outputs - means
            s = self.lower_bound_scale(scales)
            values = torch.abs(values)
            Phi = lambda t: 0.5 * torch.erfc(-(2 ** -0.5) * t)
            lik = Phi((0.5 - values) / s) - Phi((-0.5 - values) / s)
            return outputs, self.likelihood_lower_bound(lik)
        means = torch.zeros_like(inputs) if means is None else means
        params = torch.cat([scales[0].t(), means[0].t()], dim=1).contiguous()
        yhat, lik = self.forward_features(inputs[0].t().contiguous(), params)
        return yhat.t().unsqueeze(0), lik.unsqueeze(0)
