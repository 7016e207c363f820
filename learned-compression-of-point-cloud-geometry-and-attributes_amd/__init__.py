"""MI355X-native encode/decode path of the joint geometry+attribute point-cloud codec.

Import name: ``pcc_amd`` (see /pcc_amd.py at the repo root — the directory name required by the
build contract is not a valid Python identifier).

Public surface (mirrors /root/reference/model/): ``ColorModel``, ``SparseTensor``, the
``Minkowski*`` layer classes, ``EntropyBottleneck`` / ``GaussianConditional``.
"""
import os as _os

# A frame in flight uses two HIP streams (its own and the map-prefetch side stream), a streamed sequence two frames: four busy
# streams beside the default one, and the runtime spreads streams over GPU_MAX_HW_QUEUES hardware queues (default 4) — two
# streams on one queue run one after the other.  Eight queues keep the two frames' streams apart whatever order they were
# created in.  (Looked at while bench.py's `streamed` record read 82 or 102-114 ms per frame from one process to the next;
# that turned out to be the record re-creating its worker threads — one run with eight queues still hit the bad mode — but
# four busy streams on four queues shared with the default stream is a collision waiting to happen.)  Read by the HIP
# runtime at its initialisation: set here, before anything touches the GPU.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib  # noqa: E402
from ._lib import build, lib  # noqa: F401,E402
from .sparse import (CoordMap, SparseTensor, MinkowskiConvolution, MinkowskiConvolutionTranspose,  # noqa: F401
                     MinkowskiGenerativeConvolutionTranspose, MinkowskiPruning, MinkowskiReLU, MinkowskiLeakyReLU)
from .entropy import EntropyBottleneck, GaussianConditional  # noqa: F401
from .blocks import ScaledBlock, GenerativeUpBlock, ConditionEncoder  # noqa: F401
from .transforms import AnalysisTransform, SparseSynthesisTransform  # noqa: F401
from .entropy_models import MeanScaleHyperprior, MeanScaleHyperprior_Map  # noqa: F401
from .model import ColorModel  # noqa: F401
from . import synthetic, utils, parallel  # noqa: F401

__all__ = ["ColorModel", "SparseTensor", "CoordMap", "build", "lib"]
