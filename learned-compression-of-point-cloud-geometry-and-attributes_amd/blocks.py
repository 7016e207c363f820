"""Building blocks of the transforms: ScaledBlock, GenerativeUpBlock, ConditionEncoder.

Same module tree and parameter names as /root/reference/model/blocks.py so
that reference checkpoints load with ``strict=True``; the forward passes are re-organised around
the fused HIP convolution (bias + FiLM + activation + residual in the epilogue, shared kernel
maps, outputs evaluated only where the reference reads them).
"""
import os

import torch
import torch.nn as nn

from . import sparse as sp
from .sparse import (ACT_NONE, ACT_RELU, ConvChain, CoordMap, MinkowskiConvolution,
                     MinkowskiGenerativeConvolutionTranspose, MinkowskiPruning, MinkowskiReLU, SparseTensor)


class _NonNegativeParametrizer(nn.Module):
    def __init__(self, minimum=0.0, reparam_offset=2 ** -18):
        super().__init__()
        pedestal = float(reparam_offset) ** 2
        self.register_buffer("pedestal", torch.tensor([pedestal]))
        self.lower_bound = _Bound((float(minimum) + pedestal) ** 0.5)


class _Bound(nn.Module):
    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.tensor([float(bound)]))


class MinkowskiGDN(nn.Module):
    """Parameter holder for the (inverse) GDN that the reference constructs in every ScaledBlock
    (model/blocks.py:27) but never executes (its forward at :29-53 does not call it).  Kept so the
    parameter count (31,469,942 <-> README.md:125) and state_dict keys match."""

    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_reparam = _NonNegativeParametrizer(minimum=beta_min)
        self.gamma_reparam = _NonNegativeParametrizer()
        ped = float(self.beta_reparam.pedestal)
        self.beta = nn.Parameter(torch.sqrt(torch.clamp(torch.ones(in_channels) + ped, min=ped)))
        self.gamma = nn.Parameter(torch.sqrt(torch.clamp(gamma_init * torch.eye(in_channels) + ped, min=ped)))


def _conv(cin, cout, k=3, s=1, bias=True):
    return MinkowskiConvolution(in_channels=cin, out_channels=cout, kernel_size=k, stride=s, bias=bias, dimension=3)


class ScaledBlock(nn.Module):
    """model/blocks.py:10-53: conv,ReLU,conv -> x*beta+gamma -> conv,ReLU,conv,ReLU -> + x."""

    def __init__(self, N, encode=True, scale=True):
        super().__init__()
        self.encode, self.scale = encode, scale
        self.conv_1 = ConvChain(_conv(N, N), MinkowskiReLU(), _conv(N, N))
        self.conv_2 = ConvChain(_conv(N, N), MinkowskiReLU(), _conv(N, N), MinkowskiReLU())
        self.gdn = MinkowskiGDN(N, inverse=(not encode))

    def forward(self, x, condition):
        """``condition``: SparseTensor of beta|gamma ([*, 2N]).  When it lives on x's coordinate map
        its rows are already aligned and the FiLM is fused; otherwise it is looked up at x's
        coordinates first (features_at_coordinates, blocks.py:37)."""
        if condition.map is x.map:
            film = condition.F
        else:
            film = condition.features_at_coordinates(x.C)
        N = x.F.shape[1]
        if film.shape[1] == 2 and N != 1:
            # condition_ablation (blocks.py:246-247 hands the 2-channel q-map itself to blocks.py:37-40): beta and gamma are
            # one column each and broadcast over the channels
            film = torch.cat([film[:, :1].expand(-1, N), film[:, 1:].expand(-1, N)], dim=1).contiguous()
        h = self.conv_1(x, last_film=film if self.scale else None)
        return self.conv_2(h, last_residual=x.F)


_SIDE_STREAMS = {}
_HELPERS = {}


class _PrefetchHelper:
    """One helper thread per coding thread.  Building an up block's coordinate set reads a row count back from the device; on the
    coding thread that read stopped the enqueueing of the main stream's own work (q_predict, ScaledBlock) for as long as the GPU
    needed to reach it — the chip then ran dry behind every prune.  The helper makes the same calls on the side stream and does the
    waiting; the coding thread goes on and meets it at _join_prefetch."""

    def __init__(self, device):
        import queue
        import threading
        self.jobs = queue.SimpleQueue()
        self.device = device
        self.thread = threading.Thread(target=self._run, name="pcc-map-prefetch", daemon=True)
        self.thread.start()

    def _run(self):
        torch.cuda.set_device(self.device)          # a new thread starts on device 0
        while True:
            fn, done = self.jobs.get()
            try:
                with torch.no_grad():               # (grad mode is per thread too)
                    fn()
            except BaseException as e:              # re-raised on the coding thread at the join
                done.err = e
            done.set()

    def submit(self, fn):
        import threading
        done = threading.Event()
        done.err = None
        self.jobs.put((fn, done))
        return done


def _helper(device):
    import threading
    key = (threading.get_ident(), device)
    h = _HELPERS.get(key)
    if h is None:
        h = _HELPERS[key] = _PrefetchHelper(device)
    return h


def prefetch_up_maps(x_map):
    """Build, on a side stream, the coordinate set and kernel maps the next GenerativeUpBlock will need for
    ``x_map`` (candidates = k3 children; parent->candidate map; candidate->candidate maps).  They are pure
    functions of the coordinates, cached on the CoordMaps, and made of hash probes and sorts — memory-
    latency work that runs beside the MFMA-bound convolutions of the same stage (q_predict, ScaledBlock)
    instead of in front of the up block.  The calls are made by a helper thread (_PrefetchHelper), which also does the
    waiting for the candidates' row count.  Inference path only.

    Allocator note: everything allocated here belongs to the side stream's pool; a freed block can only be
    handed out again by a later prefetch, which starts with ``side.wait_stream(main)`` — after every main-
    stream kernel that read the block was enqueued."""
    if os.environ.get("PCC_PREFETCH_MAPS", "1") == "0" or torch.is_grad_enabled():
        return
    key = ("okmap_prefetched",)
    if x_map._cache.get(key):
        return
    dev = x_map.device
    main = torch.cuda.current_stream(dev)
    skey = (dev, main.cuda_stream)            # one side stream per main stream (worker threads bring their own)
    side = _SIDE_STREAMS.get(skey)
    if side is None:
        side = _SIDE_STREAMS[skey] = torch.cuda.Stream(device=dev)
    x_map.table()              # shared with the main stream's own maps of x_map: build it there, before the fork
    side.wait_stream(main)

    first = _LevelSync()        # the candidates and the parent -> candidate map: what the up block's first layer needs

    def job():
        try:
            with torch.cuda.stream(side):
                cand = x_map.up(3)
                x_map.mfma_kernel_map(cand, 3, True)
                first.publish(side.record_event())          # the generative convolution can start; the candidates' own map follows
                cand.mfma_kernel_map(cand, 3)
                cand.kernel_map(cand, 3)
                x_map._cache[("prefetch_event",)] = side.record_event()
        except BaseException as e:
            first.fail(e)
            raise

    x_map._cache[("prefetch_first",)] = first

    if PREFETCH_THREAD and x_map.n >= PREFETCH_THREAD_MIN_ROWS:
        x_map._cache[("prefetch_job",)] = _helper(dev).submit(job)
    else:
        job()
    x_map._cache[key] = True


# PCC_PREFETCH_THREAD=0: the up blocks' prefetch calls are made (and their row count waited for) on the coding thread (A/B)
PREFETCH_THREAD = os.environ.get("PCC_PREFETCH_THREAD", "1") == "1"
PREFETCH_THREAD_MIN_ROWS = 8192          # below: the hand-over costs more than the wait (the 4.9 k-point frame: 5.4 -> 5.7 ms with it)


def prefetch_analysis_maps(x_map, levels=5, hyper_ups=2):
    """The encoder's counterpart: every coordinate set below ``x_map`` (stride-2 sets down to the hyper-latents') and every
    kernel map and execution order of g_a, h_a and h_s are pure functions of the input coordinates.  Built on the side stream
    while the main stream runs the first full-resolution layers (which the caller has already enqueued), instead of one by
    one in front of the layers that use them — ~25 launches and seven count reads per frame, most of them one-workgroup
    kernels of 30-100 us on sets the chip cannot be filled with.  The calls are made by the helper thread, which publishes every
    level as soon as its maps are in the caches; the analysis transform takes them up level by level (``join_analysis_level``).
    Inference path only."""
    if os.environ.get("PCC_PREFETCH_MAPS", "1") == "0" or torch.is_grad_enabled():
        return
    key = ("analysis_prefetched",)
    if x_map._cache.get(key):
        return
    dev = x_map.device
    main = torch.cuda.current_stream(dev)
    skey = (dev, main.cuda_stream)
    side = _SIDE_STREAMS.get(skey)
    if side is None:
        side = _SIDE_STREAMS[skey] = torch.cuda.Stream(device=dev)
    x_map.table()              # shared with the main stream's own maps of x_map: build it there, before the fork
    side.wait_stream(main)
    # one (host event, stream event) pair per level and one for the rest: the coding thread takes level L's maps up as soon as THEY are
    # there (join_analysis_level) and runs that level's convolutions while the helper goes on with the coarser ones — until round 4's
    # end the main stream stood still for ~2.4 ms at the start of every encode, waiting for all levels behind a single event
    syncs = [_LevelSync() for _ in range(levels + 1)]

    def job():
        try:
            with torch.cuda.stream(side):
                m, sets = x_map, [x_map]
                for lv in range(levels):
                    d = m.down()
                    m.mfma_kernel_map(d, 3)                          # the stride-2 convolution onto the coarser set
                    d.mfma_kernel_map(d, 3)                          # the stride-1 convolutions on it
                    sets.append(d)
                    m = d
                    syncs[lv].publish(side.record_event())
                y_map = sets[3] if len(sets) > 3 else None           # stride 8: the latents' set; h_s ends on it
                u = m
                for _ in range(hyper_ups if y_map is not None and levels >= 5 else 0):
                    c = u.up(2)                                      # h_s: generative transposed convolutions, kernel 2
                    u.mfma_kernel_map(c, 2, True)
                    c.mfma_kernel_map(c, 3)
                    u = c
                if y_map is not None and u is not m:
                    u.mfma_kernel_map(y_map, 3)                      # h_s's last layer, evaluated at the latents' coordinates
                syncs[levels].publish(side.record_event())
        except BaseException as e:
            for sy in syncs:
                sy.fail(e)
            raise

    x_map._cache[("analysis_syncs",)] = syncs
    if PREFETCH_THREAD and x_map.n >= PREFETCH_THREAD_MIN_ROWS:
        x_map._cache[("prefetch_job",)] = _helper(dev).submit(job)
    else:
        job()
    x_map._cache[key] = True


class _LevelSync:
    """what one level of prefetch_analysis_maps hands over: set once its maps are in the caches, with the side stream's event behind them"""

    def __init__(self):
        import threading
        self.ready = threading.Event()
        self.event = None
        self.err = None

    def publish(self, event):
        self.event = event
        self.ready.set()

    def fail(self, err):
        if not self.ready.is_set():
            self.err = err
            self.ready.set()


def join_analysis_level(x_map, level):
    """Before the coding thread touches the coordinate set / maps of stride 2^level below ``x_map`` (level 1 = the first stride-2 set;
    level = -1: everything, incl. the hyper-latents' sets): wait for the helper to have put them into the caches, and make the current
    stream wait for the side stream's kernels behind them.  No-op when nothing was prefetched."""
    syncs = x_map._cache.get(("analysis_syncs",))
    if not syncs:
        return
    last = level == -1 or level >= len(syncs)
    sy = syncs[-1] if last else syncs[level - 1]
    sy.ready.wait()
    if sy.err is not None:
        x_map._cache.pop(("analysis_syncs",), None)
        x_map._cache.pop(("prefetch_job",), None)
        raise sy.err
    torch.cuda.current_stream(x_map.device).wait_event(sy.event)
    if last:
        x_map._cache.pop(("analysis_syncs",), None)
        _join_prefetch(x_map)


def _join_prefetch_first(x_map):
    """the first publication of prefetch_up_maps (candidates + parent -> candidate map): the rest is joined by _join_prefetch"""
    sy = x_map._cache.pop(("prefetch_first",), None)
    if sy is None:
        return
    sy.ready.wait()
    if sy.err is not None:
        x_map._cache.pop(("prefetch_job",), None)
        raise sy.err
    torch.cuda.current_stream(x_map.device).wait_event(sy.event)


def _join_prefetch(x_map):
    x_map._cache.pop(("prefetch_first",), None)
    done = x_map._cache.pop(("prefetch_job",), None)
    if done is not None:
        done.wait()
        if done.err is not None:
            raise done.err
    ev = x_map._cache.pop(("prefetch_event",), None)
    if ev is not None:
        torch.cuda.current_stream(x_map.device).wait_event(ev)


class GenerativeUpBlock(nn.Module):
    """model/blocks.py:78-181."""

    def __init__(self, N_in, N_out, predict=False, dense=True, condition_ablation=None):
        super().__init__()
        self.dense = dense
        self.condition_ablation = condition_ablation
        self.conv = MinkowskiGenerativeConvolutionTranspose(in_channels=N_in, out_channels=N_out, kernel_size=3,
                                                            stride=2, bias=True, dimension=3)
        self.conv_2 = ConvChain(_conv(N_out, N_out), MinkowskiReLU(), _conv(N_out, N_out))
        self.prune = MinkowskiPruning()
        self.predict = predict
        if predict:
            self.occ_predict = ConvChain(_conv(N_out, N_out), MinkowskiReLU(), _conv(N_out, N_out))

    def forward(self, x, coords=None, k=None, full_predictions=False):
        if not self.predict:
            return self._follow(x, coords)
        in_map = x.map
        _join_prefetch_first(in_map)
        x = self.conv(x)                                   # genConvT k3 s2 -> all candidates
        _join_prefetch(in_map)                             # the candidates' own maps (conv_2, the occupancy head)
        # blocks.py:156-175: dense (the shipped configs) refines the candidates before the occupancy head; dense=False
        # predicts on the raw candidates and refines the kept rows; condition_ablation drops conv_2 in either order
        if self.dense and self.condition_ablation is None:
            x = self.conv_2(x)
        # only channel 0 of the occupancy head is ever read (blocks.py:142): evaluate just that
        # column unless the caller wants the full tensor (training losses).
        pred = self.occ_predict(x, last_out_channels=None if full_predictions else 1)
        nb = x.map._nbatch if x.map._nbatch is not None else x.map.nbatch
        mask = sp.topk_mask(pred.F, pred.C, k, nb)
        # one item: the selection keeps exactly min(k, candidates) rows (ties are broken by coordinate) — no count to wait for
        kept = min(max(int(k[0]), 0), x.map.n) if nb == 1 and len(k) == 1 and not torch.is_grad_enabled() else None
        coords_kept, feats_kept, _, _ = sp.compact_rows(mask, x.C, x.F, expected=kept)
        up_map = CoordMap(coords_kept, x.map.stride, nbatch=x.map._nbatch)
        x = SparseTensor(feats_kept, coordinate_map=up_map)
        if not self.dense and self.condition_ablation is None:
            x = self.conv_2(x)
        return x, pred, up_map

    def _follow(self, Q, up_map):
        """predict=False (q_up_i, blocks.py:179-181): genConvT then prune to the kept coordinates.
        Evaluated directly at the kept coordinates (same values, no candidate rows materialised);
        the result lives on ``up_map`` so it stays row-aligned with the main branch."""
        if not isinstance(up_map, CoordMap):
            up_map = CoordMap(sp._as_int_coords(up_map), Q.map.stride // 2, nbatch=Q.map._nbatch)
        return self.conv(Q, out_map=up_map)


class ConditionEncoder(nn.Module):
    """model/blocks.py:185-251 (conv_layers are constructed but never run: blocks.py:241)."""

    def __init__(self, C_in, N_scales, N_features, condition_ablation=None):
        super().__init__()
        self.num_stages = len(N_scales)
        self.condition_ablation = condition_ablation
        self.pre_conv = ConvChain(_conv(C_in, N_features[0]), MinkowskiReLU())
        self.conv_layers = nn.ModuleList()
        self.predict_layers = nn.ModuleList()
        self.down_layers = nn.ModuleList()
        for i in range(self.num_stages):
            down = _conv(N_features[i], N_features[i + 1], 3, 2)
            self.down_layers.append(down)
            self._register_layers(down, f"down_layers_{i}")
            conv = ConvChain(_conv(N_features[i + 1], N_features[i + 1]), MinkowskiReLU(),
                             _conv(N_features[i + 1], N_features[i + 1]))
            self.conv_layers.append(conv)
            self._register_layers(conv, f"conv_layers_{i}")
            pred = ConvChain(_conv(N_features[i + 1], N_scales[i]), MinkowskiReLU(),
                             _conv(N_scales[i], N_scales[i], 1), MinkowskiReLU(),
                             _conv(N_scales[i], N_scales[i] * 2))
            self.predict_layers.append(pred)
            self._register_layers(pred, f"predict_layers_{i}")

    def _register_layers(self, layer, name):
        # the reference re-registers every parameter under a flat alias (blocks.py:228-231)
        for pid, param in layer.named_parameters():
            self.register_parameter(f"{name}_{pid}".replace(".", "_"), param)

    def begin(self, Q):
        """pre_conv on the q-map's own set (blocks.py:240)"""
        if self.condition_ablation not in (None, "condition_ablation"):
            raise ValueError(f"condition_ablation={self.condition_ablation!r}: the reference defines None and "
                             "'condition_ablation' only (blocks.py:244-247)")
        return self.pre_conv(Q)

    def stage(self, i, Q):
        """one level of the pyramid (blocks.py:242-249): -> (the q-map one stride down, that level's beta | gamma).  The analysis
        transform calls the levels one by one, each in front of its own ScaledBlock — the same values as the reference's loop, which runs
        all three levels first; the order lets a level's convolutions start while the coarser coordinate sets are still being built."""
        Q = self.down_layers[i](Q)
        # configs/Ablation_NoCondition_Convolution.yaml: the down-sampled q-map itself is the (1 + 1)-channel beta | gamma
        return Q, (self.predict_layers[i](Q) if self.condition_ablation is None else Q)

    def forward(self, Q):
        Q = self.begin(Q)
        beta_gammas = []
        for i in range(self.num_stages):
            Q, bg = self.stage(i, Q)
            beta_gammas.append(bg)
        return Q, beta_gammas
