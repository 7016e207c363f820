"""Sparse tensors and sparse-convolution layers on top of libpcc_hip.so.

Host-side mirror of the MinkowskiEngine surface the reference uses:
``SparseTensor`` (``.C``, ``.F``, ``.tensor_stride``, ``.device``, ``.coordinate_manager``,
``.features_at_coordinates``), ``MinkowskiConvolution``, ``MinkowskiGenerativeConvolutionTranspose``,
``MinkowskiConvolutionTranspose``, ``MinkowskiPruning``, ``MinkowskiReLU`` / ``MinkowskiLeakyReLU``.
Layer parameters keep ME's names and shapes (``kernel`` [K, C_in, C_out] or [C_in, C_out] for
kernel_size 1, ``bias`` [1, C_out]) so reference state_dicts load unchanged.

Everything numeric runs in the HIP library; torch provides device memory, streams and the
parameter containers only.
"""
import math
import weakref

import os

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2


def _require_cuda(t):
    if not t.is_cuda:
        raise RuntimeError("libpcc_hip operators need tensors on an MI355X device (got a CPU tensor); "
                           "there is no CPU fallback")


# Optional log of the HBM-bound (non-matrix) operators for bench.py's `roofline_hbm` record: a list that receives
# (operator class, rows, algorithmic bytes or a callable returning them, start event, end event, on the main stream?) per call.
# Events are recorded on the stream the kernels are launched on (torch's current stream: the frame's own or the map-prefetch
# side stream).  Off (None) in the timed steps of a benchmark: two events are ~9 us of host time per call.
COORD_PROFILER = None


def _cp_begin():
    if COORD_PROFILER is None:
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


def _cp_end(ev0, name, rows, alg_bytes):
    if ev0 is None or COORD_PROFILER is None:
        return
    ev1 = torch.cuda.Event(enable_timing=True)
    ev1.record()
    main = torch.cuda.current_stream() == torch.cuda.default_stream()       # else: the map-prefetch side stream
    COORD_PROFILER.append((name, int(rows), alg_bytes, ev0, ev1, main))


def _as_int_coords(coords):
    """ME floors non-int32 coordinates (reference: model/model.py:184-188, utils.py:438)."""
    if coords.dtype.is_floating_point:
        coords = torch.floor(coords)
    return coords.to(torch.int32).contiguous()


_COUNT_BUFS = {}


def _host_count():
    """A page-locked int64 the device writes a row count into directly (zero-copy).  The count kernel (the scan of the
    block sums) runs BEFORE the kernels that move the rows, so the host — which polls the word instead of synchronising
    the stream — learns the count while those are still running and goes on allocating and enqueueing.  One word per
    host thread (worker threads code frames concurrently); armed with -1 before every use."""
    import threading
    key = threading.get_ident()
    hit = _COUNT_BUFS.get(key)
    if hit is None:
        buf = torch.zeros(1, dtype=torch.int64, pin_memory=True)
        hit = _COUNT_BUFS[key] = (buf, buf.numpy())
    hit[1][0] = -1
    return hit[0]


_SMALL_MAP_MAX = None


def _small_map_max():
    """rows up to which a kernel map and its execution order are built by one launch (0 = never; PCC_SMALL_MAP=0)"""
    global _SMALL_MAP_MAX
    if _SMALL_MAP_MAX is None:
        _SMALL_MAP_MAX = int(_lib.lib().pcc_small_map_max())
    return _SMALL_MAP_MAX


def set_small_map_max(rows):
    """tests / A-B: cap (or switch off with 0) the one-launch small maps; returns the previous cap"""
    global _SMALL_MAP_MAX
    was = _small_map_max()
    _SMALL_MAP_MAX = min(int(rows), int(_lib.lib().pcc_small_map_max())) if rows else 0
    return was


def set_small_paths(mask):
    """one-workgroup forms of the small per-map chains (bit 0 execution order, 1 top-k, 2 coordinate sets; include/pcc_hip.h):
    sets the mask, returns the previous one; negative = read only"""
    return int(_lib.lib().pcc_small_paths(int(mask)))


class CoordinateRangeError(ValueError):
    """a coordinate outside the voxel key's range (|c| <= 130000, 0 <= batch index <= 1022: include/pcc_hip.h)"""


COUNT_ARMED, COUNT_ERR_RANGE = -1, -2            # PCC_COUNT_ERR_RANGE of include/pcc_hip.h
_POLL_YIELD = os.environ.get("PCC_COUNT_POLL_YIELD", "1") == "1"      # 0: spin without yielding the interpreter lock (A/B)


def _read_count(buf, device):
    """The row count the scan kernel wrote into this thread's page-locked word.  A short poll (the kernel that writes it
    runs ahead of the row movers, so the word usually lands while the host is still here), yielding the interpreter lock
    between reads so that the other coding threads of a streamed run are not starved; then a real wait on the stream,
    which releases the lock for its whole duration."""
    import time
    arr = _COUNT_BUFS[__import__("threading").get_ident()][1]
    deadline = time.perf_counter() + 2e-3
    while arr[0] == COUNT_ARMED:
        if time.perf_counter() > deadline:        # not visible yet (or the stream is behind): fall back to a real wait
            torch.cuda.current_stream(device).synchronize()
            break
        if _POLL_YIELD:
            time.sleep(0)
    n = int(arr[0])
    if n == COUNT_ERR_RANGE:
        raise CoordinateRangeError("libpcc_hip: a voxel coordinate is outside the supported range (|c| <= 130000, batch index "
                                   "<= 1022): the 18-bit fields of the voxel key would alias — translate the cloud towards the "
                                   "origin or re-voxelise it")
    if n < 0:
        raise RuntimeError("libpcc_hip: the row count was never written (kernel failure?)")
    return n


class PairCount:
    """The pair count of a kernel map (sum of the row masks' popcounts = the `pairs` of 2 * pairs * C_in * C_out), computed on
    first use: the codec never reads it — only FLOP accounting does (bench.py's profiler, tools/) — so building a map
    costs no reduction kernel and no memset for it.  Reads like the device scalar it used to be: .item(), int(), float()."""

    def __init__(self, row_mask):
        self._mask, self._val = row_mask, None

    def item(self):
        if self._val is None:
            out = torch.empty(1, dtype=torch.int64, device=self._mask.device)
            check(_lib.lib().pcc_pair_count(ptr(self._mask), self._mask.shape[0], ptr(out), _lib.stream()))
            self._val = int(out.item())
        return self._val

    __int__ = item

    def __float__(self):
        return float(self.item())


def _same_map(kept, out_map, self_map):
    """is a cached kernel map the one for `out_map`?  Entries are keyed by id(out_map) and remember the map itself — weakly:
    a strong reference makes a cycle wherever an output map is an ancestor of the input map (h_s back onto y's coordinates:
    y.map -> "down" -> ... -> z.map -> this entry -> y.map), and a frame's maps, with the device memory of their tables and
    kernel maps, then wait for the cyclic collector instead of going back to the allocator when the frame is done.  A dead
    reference is a miss, so an id reused by a new map can never hit an old entry."""
    if kept is None:
        return out_map is self_map
    return kept() is out_map


class CoordMap:
    """A coordinate set of one tensor stride with its hashed-voxel table and cached kernel maps.

    Plays the role of ME's coordinate manager + coordinate-map key (N1-N3 of SURVEY.md §2.2).
    """

    def __init__(self, coords, stride=1, table=None, nbatch=None):
        _require_cuda(coords)
        assert coords.dtype == torch.int32 and coords.dim() == 2 and coords.shape[1] == 4
        self.coords = coords.contiguous()
        self.stride = int(stride)
        self._table = table          # (keys uint64-as-int64, vals int32, cap)
        self._nbatch = nbatch
        self._cache = {}
        self._dup_word = None        # count_duplicates(): device int32 the table build adds the duplicate rows to

    # -- basic properties ------------------------------------------------------------------
    @property
    def n(self):
        return self.coords.shape[0]

    @property
    def device(self):
        return self.coords.device

    @property
    def nbatch(self):
        if self._nbatch is None:
            self._nbatch = int(self.coords[:, 0].max().item()) + 1 if self.n else 1
        return self._nbatch

    def table(self):
        if self._table is None:
            L = _lib.lib()
            cap = L.pcc_hash_capacity(self.n)
            keys = torch.empty(cap, dtype=torch.int64, device=self.device)
            vals = torch.empty(cap, dtype=torch.int32, device=self.device)
            ev = _cp_begin()
            check(L.pcc_hash_build(ptr(self.coords), self.n, ptr(keys), ptr(vals), cap, self.stride, ptr(self._dup_word), _lib.stream()))
            _cp_end(ev, "hash_build", self.n, 28 * self.n + 12 * cap)      # table cleared (12 B / slot), coordinates read, key + value written
            self._table = (keys, vals, cap)
        return self._table

    def count_duplicates(self):
        """Ask the table build (which must not have happened yet) to count rows whose coordinates an earlier row already holds; read
        the count with duplicates() once the stream has been waited for anyway.  For coordinate lists that come from outside
        (ColorModel.compress): ME's SparseTensor constructor would have dropped such rows (an unspecified one of each group)."""
        assert self._table is None
        self._dup_word = torch.empty(1, dtype=torch.int32, device=self.device)       # (pcc_hash_build clears it)

    def duplicates(self):
        return 0 if self._dup_word is None or self._table is None else int(self._dup_word.item())

    def lookup(self, query):
        """Row index of every query coordinate (int32 [M,4]) or -1."""
        keys, vals, cap = self.table()
        q = _as_int_coords(query)
        out = torch.empty(q.shape[0], dtype=torch.int32, device=self.device)
        ev = _cp_begin()
        check(_lib.lib().pcc_hash_lookup(ptr(keys), ptr(vals), cap, self.stride, ptr(q), q.shape[0], ptr(out), _lib.stream()))
        _cp_end(ev, "hash_lookup", q.shape[0], 32 * q.shape[0])           # coordinate, key probe, value, index written
        return out

    # -- derived coordinate sets -------------------------------------------------------------
    def _unique(self, fn_name, m, *args):
        L = _lib.lib()
        cap = L.pcc_hash_capacity(m)
        dev = self.device
        keys = torch.empty(cap, dtype=torch.int64, device=dev)
        vals = torch.empty(cap, dtype=torch.int32, device=dev)
        scratch = torch.empty(L.pcc_scan_scratch_elems(m), dtype=torch.int32, device=dev)
        out = torch.empty((max(m, 1), 4), dtype=torch.int32, device=dev)
        count = _host_count()
        ev = _cp_begin()
        check(getattr(L, fn_name)(ptr(self.coords), self.n, *args, ptr(keys), ptr(vals), cap, ptr(scratch),
                                  ptr(out), ptr(count), _lib.stream()))
        n_out = _read_count(count, dev)      # the one host sync of a coordinate-set construction
        # table cleared, 16 B per source row, 12 B per candidate (key claim + flag), 16 B + 12 B per unique row written
        _cp_end(ev, "unique_" + fn_name[4:], m, 12 * cap + 16 * self.n + 12 * m + 28 * n_out)
        return out[:n_out], (keys, vals, cap)

    def down(self):
        """Coordinate set of a stride-2 convolution output (ME stride map)."""
        if "down" not in self._cache:
            coords, table = self._unique("pcc_stride_map", self.n, self.stride)
            self._cache["down"] = CoordMap(coords, self.stride * 2, table, self._nbatch)
        return self._cache["down"]

    def up(self, ksize):
        """Coordinate set of a generative transposed convolution output (kernel 2 or 3, stride 2)."""
        key = ("up", ksize)
        if key not in self._cache:
            coords, table = self._unique("pcc_children", self.n * ksize ** 3, self.stride, ksize)
            self._cache[key] = CoordMap(coords, self.stride // 2, table, self._nbatch)
        return self._cache[key]

    def kernel_map(self, out_map, ksize, transposed=False):
        """(nbr int32 [N_out, K], row_mask int32 [N_out], PairCount) for input=self, output=out_map."""
        key = ("kmap", id(out_map), ksize, transposed)
        hit = self._cache.get(key)
        if hit is not None and _same_map(hit[0], out_map, self):
            return hit[1:]
        keys, vals, cap = self.table()
        K = ksize ** 3
        n_out = out_map.n
        nbr = torch.empty((n_out, K), dtype=torch.int32, device=self.device)
        row_mask = torch.empty(n_out, dtype=torch.int32, device=self.device)
        step = self.stride // 2 if transposed else self.stride
        ev = _cp_begin()
        check(_lib.lib().pcc_kernel_map(ptr(out_map.coords), n_out, ptr(keys), ptr(vals), cap, ksize, step,
                                        -1 if transposed else 1, ptr(nbr), ptr(row_mask), None, _lib.stream()))
        pairs = PairCount(row_mask)
        # per row: coordinates 16 B, K key probes of 8 B (a transposed map probes only on-grid parents: ~K / 8), a value per
        # hit, K indices and the mask written
        probes = K * n_out if not transposed else max(n_out, K * n_out // 8)
        _cp_end(ev, "kernel_map", n_out, lambda: 20 * n_out + 8 * probes + 4 * int(pairs) + 4 * K * n_out)
        # the entry remembers out_map weakly (_same_map): id() alone could be reused, a strong reference makes cycles
        self._cache[key] = (None if out_map is self else weakref.ref(out_map), nbr, row_mask, pairs)
        return nbr, row_mask, pairs

    def ordered_kernel_map(self, out_map, ksize, transposed=False):
        """Kernel map with its MFMA execution order: (nbr [N_out, K] by output row, order [N_out], group_mask32, pair_count).

        Output rows are executed sorted by neighbour mask (optionally inside spatial blocks, ORDER_BLOCK_LOG2) so that 32-row
        MFMA tiles skip the offsets none of their rows has.  The table stays in output-row order — the kernels read row
        order[position] of it — so ordering a map writes 4 bytes per row, not a second copy of the table."""
        key = ("okmap", id(out_map), ksize, transposed, ORDER_BLOCK_LOG2)
        hit = self._cache.get(key)
        if hit is not None and _same_map(hit[0], out_map, self):
            return hit[1:]
        if (0 < out_map.n <= _small_map_max() and ksize in (2, 3) and ORDER_BLOCK_LOG2 < 0
                and ("kmap", id(out_map), ksize, transposed) not in self._cache):
            return self._small_ordered_kernel_map(out_map, ksize, transposed, key)
        nbr, row_mask, pairs = self.kernel_map(out_map, ksize, transposed)
        L = _lib.lib()
        n_out = nbr.shape[0]
        dev = self.device
        order = torch.empty(n_out, dtype=torch.int32, device=dev)
        gmask = torch.empty((n_out + 31) // 32, dtype=torch.int32, device=dev)
        nbytes = L.pcc_order_scratch_bytes(n_out)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ev = _cp_begin()
        check(L.pcc_order_rows_by_mask(ptr(row_mask), ptr(out_map.coords), n_out, ORDER_BLOCK_LOG2, out_map.stride,
                                       ptr(order), ptr(gmask), ptr(scratch), nbytes, _lib.stream()))
        # offset counts 4 B, keys 4 + 4 B, four radix passes of (4 B counted + 8 B read + 8 B written), group masks 8 B per row
        _cp_end(ev, "execution_order", n_out, 100 * n_out)
        self._cache[key] = (None if out_map is self else weakref.ref(out_map), nbr, order, gmask, pairs)
        return nbr, order, gmask, pairs

    def _small_ordered_kernel_map(self, out_map, ksize, transposed, key):
        """kernel_map + ordered_kernel_map of a small map in one launch (csrc/select.hip small_map_kernel); fills both caches"""
        L = _lib.lib()
        keys, vals, cap = self.table()
        n_out, K, dev = out_map.n, ksize ** 3, self.device
        nbr = torch.empty((n_out, K), dtype=torch.int32, device=dev)
        row_mask = torch.empty(n_out, dtype=torch.int32, device=dev)
        order = torch.empty(n_out, dtype=torch.int32, device=dev)
        gmask = torch.empty((n_out + 31) // 32, dtype=torch.int32, device=dev)
        nbytes = L.pcc_order_scratch_bytes(n_out)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        step = self.stride // 2 if transposed else self.stride
        ev = _cp_begin()
        check(L.pcc_small_kernel_map(ptr(out_map.coords), n_out, ptr(keys), ptr(vals), cap, ksize, step, -1 if transposed else 1,
                                     ptr(nbr), ptr(row_mask), ptr(order), ptr(gmask), ptr(scratch), nbytes, _lib.stream()))
        pairs = PairCount(row_mask)
        _cp_end(ev, "small_map+order", n_out, (120 + 12 * K) * n_out)
        keep = None if out_map is self else weakref.ref(out_map)
        self._cache[("kmap", id(out_map), ksize, transposed)] = (keep, nbr, row_mask, pairs)
        self._cache[key] = (keep, nbr, order, gmask, pairs)
        return nbr, order, gmask, pairs

    def position_ordered_table(self, out_map, ksize, transposed=False):
        """the kernel map's table permuted into execution order (nbr[order]) for the weight-gradient kernels of the training
        path, which index it by execution position (csrc/conv_bwd.hip): (nbr_sorted, order, group_mask32, pair_count)"""
        key = ("okmap_sorted", id(out_map), ksize, transposed, ORDER_BLOCK_LOG2)
        hit = self._cache.get(key)
        if hit is not None and _same_map(hit[0], out_map, self):
            return hit[1:]
        nbr, order, gmask, pairs = self.ordered_kernel_map(out_map, ksize, transposed)
        nbr_sorted = torch.empty_like(nbr)
        check(_lib.lib().pcc_permute_map_rows(ptr(nbr), ptr(order), nbr.shape[0], nbr.shape[1], ptr(nbr_sorted), _lib.stream()))
        self._cache[key] = (None if out_map is self else weakref.ref(out_map), nbr_sorted, order, gmask, pairs)
        return nbr_sorted, order, gmask, pairs

    def mfma_kernel_map(self, out_map, ksize, transposed=False):
        """Build (and cache) the map form the wide inference convolutions will ask for — what the prefetchers call"""
        return self.ordered_kernel_map(out_map, ksize, transposed)

    def count_per_batch(self):
        """AnalysisTransform.count_per_batch (model/transforms.py:65-71)."""
        if self._nbatch == 1:
            return [self.n]
        nb = self.nbatch
        counts = torch.empty(nb, dtype=torch.int32, device=self.device)
        check(_lib.lib().pcc_count_per_batch(ptr(self.coords), self.n, nb, ptr(counts), _lib.stream()))
        return [int(v) for v in counts.tolist() if v > 0]

    def sort_permutation(self):
        """perm with coords[perm] in canonical (b,x,y,z) order (utils.py:155-204)."""
        L = _lib.lib()
        nbytes = L.pcc_sort_scratch_bytes(self.n)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        perm = torch.empty(self.n, dtype=torch.int32, device=self.device)
        ev = _cp_begin()
        check(L.pcc_sort_coords(ptr(self.coords), self.n, ptr(perm), ptr(scratch), nbytes, _lib.stream()))
        _cp_end(ev, "canonical_sort", self.n, (24 + 8 * 32) * self.n)        # keys made, eight passes of (8 counted + 12 read + 12 written)
        return perm


class SparseTensor:
    """Features [N, C] on a CoordMap.  Mirrors the ME.SparseTensor surface used by the reference."""

    def __init__(self, features=None, coordinates=None, tensor_stride=1, coordinate_manager=None,
                 device=None, coordinate_map=None, nbatch=None):
        if coordinate_map is None:
            if coordinates is None:
                raise ValueError("coordinates or coordinate_map required")
            if device is not None:
                coordinates = coordinates.to(device)
            if isinstance(tensor_stride, (list, tuple)):
                tensor_stride = tensor_stride[0]
            coordinate_map = CoordMap(_as_int_coords(coordinates), tensor_stride, nbatch=nbatch)
        self.map = coordinate_map
        if features is None:
            features = torch.ones((self.map.n, 1), dtype=torch.float32, device=self.map.device)
        features = features.to(device=self.map.device, dtype=torch.float32)
        if features.shape[0] != self.map.n:
            raise ValueError(f"{features.shape[0]} feature rows for {self.map.n} coordinates")
        self.F = features.contiguous()

    @classmethod
    def _wrap(cls, feats, coordinate_map):
        """an operator's own output (fp32, contiguous, on the map's device, one row per coordinate): no checks, no conversions —
        the constructor's cost is paid a hundred times per frame"""
        t = object.__new__(cls)
        t.map = coordinate_map
        t.F = feats
        return t

    @property
    def C(self):
        return self.map.coords

    @property
    def tensor_stride(self):
        return [self.map.stride] * 3

    @property
    def coordinate_manager(self):
        return self.map

    @property
    def device(self):
        return self.F.device

    @property
    def shape(self):
        return self.F.shape

    def features_at_coordinates(self, query):
        """Exact lookup, zero-fill if absent (every query on the path is on-grid: SURVEY.md N8)."""
        idx = self.map.lookup(query)
        return gather_rows(self.F, idx)


# ---------------------------------------------------------------------------------------------
# functional operators
# ---------------------------------------------------------------------------------------------
def _tracked(*tensors):
    """True when autograd must see the operation (training path): the row movers then run as autograd functions
    over the same HIP kernels (_GatherRowsFn, _ScatterRowsFn, _CompactFeatsFn)"""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class _GatherRowsFn(torch.autograd.Function):
    """rows = src[idx] (zeros where idx < 0) on the HIP row movers, both directions (torch's index_select
    backward takes 14 ms for 812 k x 3 on this stack; the atomic scatter-add 0.1 ms)"""

    @staticmethod
    def forward(ctx, src, idx):
        src = src.contiguous()
        n, c = idx.shape[0], src.shape[1]
        out = torch.empty((n, c), dtype=torch.float32, device=src.device)
        check(_lib.lib().pcc_gather_rows(ptr(src), c, ptr(idx), n, ptr(out), 0, _lib.stream()))
        ctx.save_for_backward(idx)
        ctx.rows = src.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = g.contiguous()
        d = torch.zeros((ctx.rows, g.shape[1]), dtype=torch.float32, device=g.device)
        check(_lib.lib().pcc_scatter_add_rows(ptr(g), g.shape[1], ptr(idx), idx.shape[0], ptr(d), _lib.stream()))
        return d, None


def gather_rows(src, idx, out=None, accumulate=False):
    """out[r] (+)= src[idx[r]] (zero rows where idx < 0).  Returns the result: callers must use the return
    value (on the training path a new tensor is returned instead of writing into ``out``)."""
    if _tracked(src, out):
        sel = _GatherRowsFn.apply(src, idx.contiguous())
        return out + sel if (accumulate and out is not None) else sel
    n = idx.shape[0]
    c = src.shape[1]
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=src.device)
    ev = _cp_begin()
    check(_lib.lib().pcc_gather_rows(ptr(src), c, ptr(idx), n, ptr(out), 1 if accumulate else 0, _lib.stream()))
    _cp_end(ev, "gather_rows", n, (4 + (12 if accumulate else 8) * c) * n)
    return out


class _ScatterRowsFn(torch.autograd.Function):
    """out[idx[r]] += src[r] (rows with idx < 0 dropped) and its backward d_src[r] = g[idx[r]], both on the HIP row movers"""

    @staticmethod
    def forward(ctx, src, idx, n_out):
        src = src.contiguous()
        out = torch.zeros((n_out, src.shape[1]), dtype=torch.float32, device=src.device)
        check(_lib.lib().pcc_scatter_add_rows(ptr(src), src.shape[1], ptr(idx), idx.shape[0], ptr(out), _lib.stream()))
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = g.contiguous()
        d = torch.empty((idx.shape[0], g.shape[1]), dtype=torch.float32, device=g.device)
        check(_lib.lib().pcc_gather_rows(ptr(g), g.shape[1], ptr(idx), idx.shape[0], ptr(d), 0, _lib.stream()))
        return d, None, None


class _CompactFeatsFn(torch.autograd.Function):
    """Differentiable half of compact_rows: kept rows of ``feats`` in order; backward hands a kept row's gradient back to
    its source row (new_index = position of a kept row, -1 for a dropped one) and zeros to the dropped rows"""

    @staticmethod
    def forward(ctx, feats, mask, coords):
        out_c, out_f, new_index, _ = compact_rows(mask, coords, feats.detach().contiguous(), True)
        ctx.save_for_backward(new_index)
        if out_c is None:
            out_c = new_index.new_empty((0, 4))
        ctx.mark_non_differentiable(out_c, new_index)
        return out_f, out_c, new_index

    @staticmethod
    def backward(ctx, g, _gc, _gi):
        (new_index,) = ctx.saved_tensors
        g = g.contiguous()
        d = torch.empty((new_index.shape[0], g.shape[1]), dtype=torch.float32, device=g.device)
        check(_lib.lib().pcc_gather_rows(ptr(g), g.shape[1], ptr(new_index), new_index.shape[0], ptr(d), 0, _lib.stream()))
        return d, None, None


def scatter_rows(src, idx, n_out):
    if _tracked(src):
        return _ScatterRowsFn.apply(src, idx.contiguous(), n_out)
    out = torch.zeros((n_out, src.shape[1]), dtype=torch.float32, device=src.device)
    check(_lib.lib().pcc_scatter_rows(ptr(src), src.shape[1], ptr(idx), idx.shape[0], ptr(out), _lib.stream()))
    return out


def compact_rows(mask, coords=None, feats=None, want_index=False, expected=None):
    """Order-preserving compaction (ME.MinkowskiPruning).  Returns (coords, feats, new_index, n).

    ``expected``: the number of set mask bytes when the caller knows it (a top-k selection of one item keeps exactly min(k, n) rows):
    the count is then not read back (PCC_CHECK_COUNTS=1 reads and compares it) — the coding thread does not stop enqueueing."""
    if _tracked(feats):
        feats_k, coords_k, new_index = _CompactFeatsFn.apply(feats, mask, coords)
        return (coords_k if coords is not None else None), feats_k, (new_index if want_index else None), feats_k.shape[0]
    L = _lib.lib()
    n = mask.shape[0]
    dev = mask.device
    scratch = torch.empty(L.pcc_scan_scratch_elems(n), dtype=torch.int32, device=dev)
    known = expected is not None and n > 0 and not CHECK_EXPECTED_COUNTS
    count = None if known else _host_count()
    out_c = torch.empty((n, 4), dtype=torch.int32, device=dev) if coords is not None else None
    c = feats.shape[1] if feats is not None else 0
    out_f = torch.empty((n, c), dtype=torch.float32, device=dev) if feats is not None else None
    new_index = torch.empty(n, dtype=torch.int32, device=dev) if want_index else None
    ev = _cp_begin()
    check(L.pcc_compact_rows(ptr(mask), n, ptr(coords), ptr(out_c), ptr(feats), c, ptr(out_f), ptr(new_index),
                             ptr(scratch), ptr(count), _lib.stream()))
    if known:
        m = int(expected)
    else:
        m = _read_count(count, dev)
        if expected is not None and m != int(expected):
            raise RuntimeError(f"compact_rows: {m} rows kept where the caller expected {int(expected)}")
    # mask + scan per row, coordinates and features of the kept rows read and written
    _cp_end(ev, "prune", n, 13 * n + (2 * (16 if coords is not None else 0) + 8 * c) * m)
    return (out_c[:m] if out_c is not None else None, out_f[:m] if out_f is not None else None, new_index, m)


def topk_mask(logits, coords, k_per_batch, nbatch):
    """Per-batch top-k on column 0 of ``logits`` (model/blocks.py:130-150)."""
    L = _lib.lib()
    n = logits.shape[0]
    dev = logits.device
    # the counts reach the device without a stream synchronisation: torch.tensor(list, device=...) copies from pageable
    # memory, i.e. it blocks the host until everything queued before it — the candidate-level convolutions — has run
    ks = [int(v) for v in k_per_batch]
    assert len(ks) == nbatch, (len(ks), nbatch)
    if nbatch == 1:
        k = torch.full((1,), ks[0], dtype=torch.int32, device=dev)
    else:
        k = torch.tensor(ks, dtype=torch.int32).pin_memory().to(dev, non_blocking=True)
    state = torch.empty(L.pcc_topk_state_elems(nbatch), dtype=torch.int32, device=dev)
    mask = torch.empty(n, dtype=torch.uint8, device=dev)
    ev = _cp_begin()
    check(L.pcc_topk_mask(ptr(logits), logits.stride(0), ptr(coords), n, nbatch, ptr(k), ptr(mask), ptr(state),
                          _lib.stream()))
    _cp_end(ev, "top_k", n, 37 * n)               # four passes over the fp32 logit, then logit + coordinates read, mask written
    return mask


def conv_forward(x_feats, in_map, out_map, layer, ksize, transposed=False, act=ACT_NONE, film=None,
                 residual=None, out_channels=None):
    """out = act(film(bias + sum_k in[nbr] @ W[k])) + residual — one fused launch."""
    L = _lib.lib()
    w, wp, bias = layer.weights(out_channels)
    n_in, cin = x_feats.shape
    cout = w.shape[-1]
    if ksize > 1 and cin % 32 == 0 and cout <= NARROW_HEAD_MAX_COUT and film is None and residual is None:
        return _narrow_head_forward(x_feats, in_map, out_map, layer, ksize, transposed, act, out_channels)
    order = gmask = None
    bf16 = INFER_BF16 and cin % 64 == 0 and n_in * cin * 2 < 0xFFFFF000
    if (THIN_IM2COL and ksize > 1 and cin in (1, 2, 4, 8) and cout % 32 == 0 and ksize ** 3 * cin <= 256
            and out_map.n * ((ksize ** 3 * cin + 31) // 32 * 32) * 4 < 0xFFFFF000):
        return _thin_im2col_forward(x_feats, in_map, out_map, layer, ksize, transposed, act, film, residual, out_channels)
    if ksize == 1:
        nbr = pairs = None
        K = 1
    elif cin % 32 == 0:
        nbr, order, gmask, pairs = in_map.ordered_kernel_map(out_map, ksize, transposed)
        K = ksize ** 3
    else:
        nbr, _, pairs = in_map.kernel_map(out_map, ksize, transposed)
        K = ksize ** 3
    n_out = out_map.n
    out = torch.empty((n_out, cout), dtype=torch.float32, device=x_feats.device)
    x3 = (INFER_X3 and not bf16 and cin % 32 == 0 and ((cout + 31) // 32 * 32) % 64 == 0 and n_in * cin * 4 < 0xFFFFF000
          and (nbr is None or n_out * K * 4 < 0xFFFFF000))
    prof = PROFILER
    timed = prof is not None and n_out >= PROFILER_MIN_ROWS
    if not timed and not bf16 and not x3:
        # the default inference launch, without the bookkeeping of the other modes (a hundred of these per small frame)
        check(L.pcc_conv_fwd(x_feats.data_ptr(), n_in, cin, w.data_ptr(), None if wp is None else wp.data_ptr(),
                             None if bias is None else bias.data_ptr(), None if nbr is None else nbr.data_ptr(),
                             None if order is None else order.data_ptr(), None if gmask is None else gmask.data_ptr(), K,
                             out.data_ptr(), n_out, cout, act, None if film is None else film.data_ptr(),
                             None if residual is None else residual.data_ptr(), _lib.stream()))
        if prof is not None:                          # counted (FLOPs, launches), not bracketed by events
            prof.append((("conv", "", nbr is not None), cin, cout, pairs if pairs is not None else n_out, n_out, None, None, gmask))
        return out
    log, ev0, ev1 = prof, None, None
    if not timed:
        prof = None
    if prof is not None:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    if bf16:
        xb = x_feats.to(torch.bfloat16)
        check(L.pcc_conv_fwd_bf16(ptr(xb), n_in, cin, ptr(layer.weights_bf16(out_channels)), ptr(bias), ptr(nbr),
                                  ptr(order), ptr(gmask), K, ptr(out), n_out, cout, act, ptr(film), ptr(residual), _lib.stream()))
    elif x3:
        check(L.pcc_conv_fwd_x3(ptr(x_feats), n_in, cin, ptr(layer.weights_x3(out_channels)), ptr(bias), ptr(nbr),
                                ptr(order), ptr(gmask), K, ptr(out), n_out, cout, act, ptr(film), ptr(residual), _lib.stream()))
    else:
        check(L.pcc_conv_fwd(ptr(x_feats), n_in, cin, ptr(w), ptr(wp), ptr(bias), ptr(nbr), ptr(order), ptr(gmask),
                             K, ptr(out), n_out, cout, act, ptr(film), ptr(residual), _lib.stream()))
    if prof is not None:
        ev1.record()
    if log is not None:
        # the launch's name is worked out by the reader (profiled_name): string building has no place between two launches
        log.append((("conv", "[bf16]" if bf16 else "[x3]" if x3 else "", nbr is not None), cin, cout,
                    pairs if pairs is not None else n_out, n_out, ev0, ev1, gmask))
    return out


def profiled_name(entry):
    """kernel name of a PROFILER entry (conv_forward stores what the name is made from)"""
    name, cin, cout, _, n_out = entry[:5]
    if isinstance(name, tuple):
        _, tag, has_nbr = name
        return conv_kernel_name(cin, cout, n_out, has_nbr).replace("conv_mfma_buf_kernel", "conv_mfma_buf_kernel" + tag)
    return name


_MFMA_VISIT = (0, 4, 1, 5, 2, 6, 3, 7)     # physical position, within 8 channels, of the t-th channel the MFMA loop contracts


def _thin_im2col_forward(x_feats, in_map, out_map, layer, ksize, transposed, act, film, residual, out_channels):
    """cin <= 8, cout % 32 == 0 (2 -> 128, 4 -> 64: first layers of the q-map heads and of g_a): the row's K * cin inputs gathered
    into a dense matrix (pcc_im2col_thin) and one kernel_size-1 MFMA convolution over it — the same per-element sum, in the same
    order, as the scalar thin kernel (csrc/conv.hip), at the matrix cores' rate"""
    L = _lib.lib()
    w2p, bias, k2, cout = layer.im2col_weights(out_channels)
    nbr, _, pairs = in_map.kernel_map(out_map, ksize, transposed)
    n_out, K, cin = out_map.n, ksize ** 3, x_feats.shape[1]
    dev = x_feats.device
    x2 = torch.empty((n_out, k2), dtype=torch.float32, device=dev)
    out = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
    prof = PROFILER
    if prof is not None:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    ev = _cp_begin()
    check(L.pcc_im2col_thin(ptr(x_feats), cin, ptr(nbr), n_out, K, ptr(x2), k2, _lib.stream()))
    _cp_end(ev, "im2col_thin", n_out, lambda: 4 * K * n_out + 4 * cin * int(pairs) + 4 * k2 * n_out)
    check(L.pcc_conv_fwd(ptr(x2), n_out, k2, None, ptr(w2p), ptr(bias), None, None, None, 1, ptr(out), n_out, cout, act, ptr(film),
                         ptr(residual), _lib.stream()))
    if prof is not None:
        ev1.record()
        prof.append((f"thin_im2col<{cin}>", cin, cout, pairs, n_out, ev0, ev1, None))
    return out


def _narrow_head_forward(x_feats, in_map, out_map, layer, ksize, transposed, act, out_channels):
    """cout <= 4 on wide inputs (occupancy logit, q-map heads): per-input-row scores by one dense MFMA
    GEMM, then a scalar gather-sum per output row (see csrc/conv.hip, gather_sum_kernel)."""
    L = _lib.lib()
    w_r, wp_r, bias, K, cout = layer.narrow_weights(out_channels)
    n_in, cin = x_feats.shape
    n_out = out_map.n
    nbr, _, pairs = in_map.kernel_map(out_map, ksize, transposed)
    ld = K * cout
    scores = torch.empty((n_in, ld), dtype=torch.float32, device=x_feats.device)
    out = torch.empty((n_out, cout), dtype=torch.float32, device=x_feats.device)
    prof = PROFILER
    if prof is not None:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(L.pcc_conv_fwd(ptr(x_feats), n_in, cin, ptr(w_r), ptr(wp_r), None, None, None, None, 1, ptr(scores), n_in, ld,
                         ACT_NONE, None, None, _lib.stream()))
    ev = _cp_begin()
    check(L.pcc_gather_sum_fwd(ptr(scores), ld, ptr(nbr), K, cout, ptr(bias), ptr(out), n_out, act, _lib.stream()))
    _cp_end(ev, "gather_sum", n_out, lambda: 4 * K * n_out + 4 * cout * int(pairs) + 4 * cout * n_out)
    if prof is not None:
        ev1.record()
        prof.append((f"narrow_head<{cin}>", cin, cout, pairs, n_out, ev0, ev1, None))
    return out


# Thin-input / wide-output convolutions (cin <= 8, cout % 32 == 0) run as im2col + one kernel_size-1 MFMA convolution
# (_thin_im2col_forward); PCC_THIN_IM2COL=0 takes the scalar conv_thin_kernel instead (A/B runs; results are bit-identical).
THIN_IM2COL = os.environ.get("PCC_THIN_IM2COL", "1") == "1"


def set_conv_small_max(workgroups):
    """Threshold (in 32 x 32 output tiles) below which a map convolution runs on the small-launch kernel (csrc/conv.hip,
    conv_small_kernel; bit-identical results); 0 = never.  Returns the previous threshold; negative = read only."""
    return int(_lib.lib().pcc_conv_small_max(int(workgroups)))


# Output widths up to this use the narrow-head path (K * cout score columns must fit one 128-wide GEMM tile).
NARROW_HEAD_MAX_COUT = 4

# Spatial block size (log2, in voxels of the map's stride) inside which rows are ordered by neighbour
# mask; -1 = order by mask over the whole map (best MFMA tile occupancy, least gather locality).
# PCC_CHECK_COUNTS=1: read back (and compare) the row counts a caller passes as known instead of trusting them (A/B, tests)
CHECK_EXPECTED_COUNTS = os.environ.get("PCC_CHECK_COUNTS", "0") == "1"
ORDER_BLOCK_LOG2 = int(os.environ.get("PCC_ORDER_BLOCK_LOG2", "-1"))

# Opt-in reduced-precision inference (never the default, never the headline number): convolutions whose input width
# is a multiple of 64 take bf16 operands (features cast per layer, weights packed once) on v_mfma_f32_32x32x16_bf16
# with fp32 accumulation, epilogue and outputs; narrow / thin layers and the entropy models stay fp32.  Encoder and
# decoder must run in the same mode (the streams differ from the fp32 ones); PCC_INFER_BF16=1 / set_infer_bf16().
INFER_BF16 = os.environ.get("PCC_INFER_BF16", "0") == "1"


def set_infer_bf16(enabled):
    global INFER_BF16
    INFER_BF16 = bool(enabled)


# Opt-in split-bf16 arithmetic on fp32 data (never the default, never the headline number): the wide convolutions split
# every fp32 operand exactly into three bf16 numbers and run the six significant products on v_mfma_f32_32x32x16_bf16 with
# fp32 accumulation (csrc/conv.hip, X3) — fp32-class results (dropped terms < 3 x 2^-24 per product) at 3/8 of the fp32
# MFMA time.  Features, gathers, epilogues, entropy models and streams stay fp32; encoder and decoder must run in the same
# mode.  PCC_INFER_X3=1 / set_infer_x3().
INFER_X3 = os.environ.get("PCC_INFER_X3", "0") == "1"


def set_infer_x3(enabled):
    global INFER_X3
    INFER_X3 = bool(enabled)


# Optional activation probe of the training path (tests/test_train_model.py): a list that receives (layer module, output
# coordinates, gate = pre-activation > 0) for every activated convolution of a training-mode forward, so that the gradient
# comparison with the oracle can differentiate both sides through the same ReLU gates.
GATE_LOG = None


# Optional launch log for bench.py: a list that receives one tuple per convolution launch
# (kernel class, cin, cout, pairs (device scalar or int), n_out, start event, end event).  The
# events are recorded on the stream the kernel is launched on (torch's current stream).
PROFILER = None
# Launches of fewer output rows are logged without their two events (4.6 us of host time each — more than such a launch takes
# on the device, in the stretches of a frame where the host is what the GPU waits for); bench.py --breakdown sets it to 0.
PROFILER_MIN_ROWS = 65536


CONV_BM32_MAX = int(os.environ.get("PCC_CONV_BM32_MAX", "3000"))      # mirrors csrc/conv.hip's tile switch (launch names only)
CONV_BM = int(os.environ.get("PCC_CONV_BM", "0") or 0)                 # a forced tile height also switches the small-launch kernel off


def conv_kernel_name(cin, cout, n_out=0, has_nbr=True):
    """The kernel pcc_conv_fwd dispatches to, spelled like rocprofv3 prints it (mirrors csrc/conv.hip)."""
    if cin % 32 != 0:
        cpt = 8 if cout % 8 == 0 else 4 if cout % 4 == 0 else 2 if cout % 2 == 0 else 1
        return f"conv_thin_kernel<{cin}, {cpt}>"
    coutp = (cout + 31) // 32 * 32
    wgs_small = ((n_out + 31) // 32) * (coutp // 32)
    if has_nbr and CONV_BM == 0 and wgs_small <= set_conv_small_max(-1) * (1 if coutp % 128 == 0 else 2):
        cch = cin // 32
        sc, ns = (4, 3) if cch % 4 == 0 and wgs_small <= 256 else (2, 3) if cch % 2 == 0 else (1, 8)
        return f"conv_small_kernel<{sc}, {ns}>"
    if coutp % 128 == 0:
        wgs128 = ((n_out + 63) // 64) * (coutp // 128)
        bm, bn = (64, 64) if wgs128 < 768 else (32, 128) if wgs128 < CONV_BM32_MAX else (64, 128)
    elif coutp % 64 == 0:
        bm, bn = 128, 64
    else:
        bm, bn = 128, 32
    wm, wn = (4, 1) if bn == 32 else (1, 4) if bm == 32 else (2, 2)
    return f"conv_mfma_buf_kernel<{bm}, {bn}, {wm}, {wn}, {cin // 32}, {'true' if has_nbr else 'false'}>"



# ---------------------------------------------------------------------------------------------
# layers (parameter containers + forward through the fused operator)
# ---------------------------------------------------------------------------------------------
class _ConvBase(nn.Module):
    transposed = False
    generative = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=3):
        super().__init__()
        assert dimension == 3 and dilation == 1
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = int(kernel_size), int(stride)
        K = self.kernel_size ** 3
        shape = (in_channels, out_channels) if K == 1 else (K, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape, dtype=torch.float32))
        self.bias = nn.Parameter(torch.empty((1, out_channels), dtype=torch.float32)) if bias else None
        self.reset_parameters()
        self._packed = {}

    def reset_parameters(self):
        """ME default: U(-1/sqrt(n), 1/sqrt(n)), n = C_in*K (C_out*K when transposed)."""
        K = self.kernel_size ** 3
        n = (self.out_channels if self.transposed else self.in_channels) * K
        bound = 1.0 / math.sqrt(n)
        with torch.no_grad():
            self.kernel.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def weights(self, out_channels=None):
        """(raw kernel, MFMA-packed kernel or None, bias or None), optionally sliced to the first
        ``out_channels`` outputs (used where the reference reads only channel 0: blocks.py:142)."""
        params = self._parameters                    # (not self.kernel / self.bias: Module.__getattr__ is ~0.7 us a time)
        kernel, bias_p = params["kernel"], params.get("bias")
        key = (kernel._version, kernel.data_ptr(), out_channels,
               None if bias_p is None else (bias_p._version, bias_p.data_ptr()))
        hit = self._packed.get("w")
        if hit is not None and hit[0] == key:
            return hit[1]
        w = self.kernel.detach()
        _require_cuda(w)
        if w.dim() == 2:
            w = w.unsqueeze(0)
        b = None if self.bias is None else self.bias.detach().reshape(-1)
        if out_channels is not None:
            w = w[:, :, :out_channels]
            b = None if b is None else b[:out_channels]
        w = w.contiguous()
        b = None if b is None else b.contiguous()
        K, cin, cout = w.shape
        wp = None
        if cin % 32 == 0:
            L = _lib.lib()
            wp = torch.empty(L.pcc_conv_packed_elems(K, cin, cout), dtype=torch.float32, device=w.device)
            check(L.pcc_conv_pack_weights(ptr(w), K, cin, cout, ptr(wp), _lib.stream()))
        res = (w, wp, b)
        self._packed["w"] = (key, res)
        return res

    def weights_bf16(self, out_channels=None):
        """bf16 MFMA packing of the kernel (opt-in reduced-precision inference), cached like weights()"""
        key = ("bf16", self.kernel._version, self.kernel.data_ptr(), out_channels)
        hit = self._packed.get("b")
        if hit is not None and hit[0] == key:
            return hit[1]
        w, _, _ = self.weights(out_channels)
        K, cin, cout = w.shape
        L = _lib.lib()
        wpb = torch.empty(L.pcc_conv_packed_elems_bf16(K, cin, cout), dtype=torch.bfloat16, device=w.device)
        check(L.pcc_conv_pack_weights_bf16(ptr(w), K, cin, cout, ptr(wpb), _lib.stream()))
        self._packed["b"] = (key, wpb)
        return wpb

    def weights_x3(self, out_channels=None):
        """three-plane bf16 packing of the kernel (opt-in split-bf16 arithmetic), cached like weights()"""
        key = ("x3", self.kernel._version, self.kernel.data_ptr(), out_channels)
        hit = self._packed.get("x")
        if hit is not None and hit[0] == key:
            return hit[1]
        w, _, _ = self.weights(out_channels)
        K, cin, cout = w.shape
        L = _lib.lib()
        wpx = torch.empty(L.pcc_conv_packed_elems_x3(K, cin, cout), dtype=torch.bfloat16, device=w.device)
        check(L.pcc_conv_pack_weights_x3(ptr(w), K, cin, cout, ptr(wpx), _lib.stream()))
        self._packed["x"] = (key, wpx)
        return wpx

    def im2col_weights(self, out_channels=None):
        """Kernel re-laid-out for the thin-input / wide-output path: [1, k2, cout] with row 8 g + _MFMA_VISIT[t] = W[k, ci, :] of
        logical index 8 g + t = k * cin + ci (zero rows up to k2, a multiple of 32), MFMA-packed; bias; k2; cout."""
        key = ("im2col", self.kernel._version, self.kernel.data_ptr(), out_channels,
               None if self.bias is None else (self.bias._version, self.bias.data_ptr()))
        hit = self._packed.get("i")
        if hit is not None and hit[0] == key:
            return hit[1]
        w, _, b = self.weights(out_channels)
        K, cin, cout = w.shape
        kc = K * cin
        k2 = (kc + 31) // 32 * 32
        logical = torch.zeros((k2, cout), dtype=torch.float32, device=w.device)
        logical[:kc] = w.reshape(kc, cout)
        phys = torch.tensor([8 * (j // 8) + _MFMA_VISIT[j % 8] for j in range(k2)], dtype=torch.long, device=w.device)
        w2 = torch.empty_like(logical)
        w2[phys] = logical
        w2 = w2.unsqueeze(0).contiguous()
        L = _lib.lib()
        w2p = torch.empty(L.pcc_conv_packed_elems(1, k2, cout), dtype=torch.float32, device=w.device)
        check(L.pcc_conv_pack_weights(ptr(w2), 1, k2, cout, ptr(w2p), _lib.stream()))
        res = (w2p, b, k2, cout)
        self._packed["i"] = (key, res)
        return res

    def narrow_weights(self, out_channels=None):
        """Kernel re-laid-out for the narrow-head path: [1, cin, K*cout] (+ MFMA packing), bias, K, cout."""
        key = ("narrow", self.kernel._version, self.kernel.data_ptr(), out_channels,
               None if self.bias is None else (self.bias._version, self.bias.data_ptr()))
        hit = self._packed.get("n")
        if hit is not None and hit[0] == key:
            return hit[1]
        w, _, b = self.weights(out_channels)
        K, cin, cout = w.shape
        w_r = w.permute(1, 0, 2).reshape(1, cin, K * cout).contiguous()
        L = _lib.lib()
        wp_r = torch.empty(L.pcc_conv_packed_elems(1, cin, K * cout), dtype=torch.float32, device=w.device)
        check(L.pcc_conv_pack_weights(ptr(w_r), 1, cin, K * cout, ptr(wp_r), _lib.stream()))
        res = (w_r, wp_r, b, K, cout)
        self._packed["n"] = (key, res)
        return res

    def output_map(self, in_map):
        if self.transposed:
            assert self.stride == 2
            return in_map.up(self.kernel_size)
        if self.stride == 1:
            return in_map
        assert self.stride == 2
        return in_map.down()

    def forward(self, x, act=ACT_NONE, film=None, residual=None, out_map=None, out_channels=None):
        if out_map is None:
            out_map = self.output_map(x.map)
        if torch.is_grad_enabled() and (self.kernel.requires_grad or x.F.requires_grad):
            # training path: the convolution is an autograd node on the HIP kernels, its epilogue (FiLM, activation,
            # residual — in the order the fused inference epilogue applies them) a second one (csrc/epilogue.hip)
            from .autograd import conv_train, epilogue_train
            feats = conv_train(x.F, x.map, out_map, self, self.kernel_size, self.transposed, out_channels)
            if GATE_LOG is not None and act != ACT_NONE:
                ch = feats.shape[1]
                pre = feats.detach() if film is None else feats.detach() * film.detach()[:, :ch] + film.detach()[:, ch:]
                GATE_LOG.append((self, out_map.coords, pre > 0))
            if film is not None or act != ACT_NONE or residual is not None:
                feats = epilogue_train(feats, film, residual, act)          # one kernel forward, one backward
            return SparseTensor(feats, coordinate_map=out_map)
        feats = conv_forward(x.F, x.map, out_map, self, self.kernel_size, self.transposed, act, film, residual,
                             out_channels)
        return SparseTensor._wrap(feats, out_map)


class MinkowskiConvolution(_ConvBase):
    """ME.MinkowskiConvolution (model/transforms.py:35-57 etc.)."""


class MinkowskiGenerativeConvolutionTranspose(_ConvBase):
    """ME.MinkowskiGenerativeConvolutionTranspose (model/blocks.py:84, entropy_models.py:286,290)."""
    transposed = True
    generative = True


class MinkowskiConvolutionTranspose(_ConvBase):
    """ME.MinkowskiConvolutionTranspose (model/entropy_models.py:298,302).  Output coordinates are
    generated like the generative variant (decode semantics)."""
    transposed = True


class MinkowskiReLU(nn.Module):
    """Placeholder keeping the reference's nn.Sequential indices; the activation itself is fused
    into the preceding convolution's epilogue."""
    act = ACT_RELU

    def __init__(self, inplace=False):
        super().__init__()


class MinkowskiLeakyReLU(MinkowskiReLU):
    act = ACT_LRELU


class ConvChain(nn.Sequential):
    """nn.Sequential of conv / activation modules executed with fused activations.

    Module indices (hence state_dict keys such as ``conv_1.0.kernel``) match the reference's
    nn.Sequential definitions.
    """

    def plan(self):
        hit = self.__dict__.get("_plan")
        if hit is not None and hit[0] == len(self):
            return hit[1]
        steps = []
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            assert isinstance(m, _ConvBase), f"unexpected module {type(m)} at {i}"
            act = ACT_NONE
            if i + 1 < len(mods) and isinstance(mods[i + 1], MinkowskiReLU):
                act = mods[i + 1].act
                i += 1
            steps.append((m, act))
            i += 1
        self.__dict__["_plan"] = (len(self), steps)      # the module list of a chain is fixed at construction
        return steps

    def forward(self, x, last_film=None, last_residual=None, last_out_map=None, last_out_channels=None):
        steps = self.plan()
        last_j = len(steps) - 1
        for j, (m, act) in enumerate(steps):
            # (the layers' forward directly: no hooks are registered on them, and Module.__call__ is ~1.5 us per layer)
            if j == last_j:
                x = m.forward(x, act, last_film, last_residual, last_out_map, last_out_channels)
            else:
                x = m.forward(x, act)
        return x


class MinkowskiPruning(nn.Module):
    """ME.MinkowskiPruning (model/blocks.py:90,126)."""

    def forward(self, x, mask):
        coords, feats, _, _ = compact_rows(mask.to(torch.uint8), x.C, x.F)
        return SparseTensor(feats, coordinate_map=CoordMap(coords, x.map.stride, nbatch=x.map._nbatch))
