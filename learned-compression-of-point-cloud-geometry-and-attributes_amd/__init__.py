"""MI355X-native encode/decode path of the joint geometry+attribute point-cloud codec.

Import name: ``pcc_amd`` (see /pcc_amd.py at the repo root — the directory name required by the
build contract is not a valid Python identifier).

Public surface (mirrors /root/reference/model/): ``ColorModel``, ``SparseTensor``, the
``Minkowski*`` layer classes, ``EntropyBottleneck`` / ``GaussianConditional``.
"""
from . import _lib
from ._lib import build, lib  # noqa: F401
from .sparse import (CoordMap, SparseTensor, MinkowskiConvolution, MinkowskiConvolutionTranspose,  # noqa: F401
                     MinkowskiGenerativeConvolutionTranspose, MinkowskiPruning, MinkowskiReLU, MinkowskiLeakyReLU)
from .entropy import EntropyBottleneck, GaussianConditional  # noqa: F401
from .blocks import ScaledBlock, GenerativeUpBlock, ConditionEncoder  # noqa: F401
from .transforms import AnalysisTransform, SparseSynthesisTransform  # noqa: F401
from .entropy_models import MeanScaleHyperprior_Map  # noqa: F401
from .model import ColorModel  # noqa: F401
from . import synthetic, utils, parallel  # noqa: F401

__all__ = ["ColorModel", "SparseTensor", "CoordMap", "build", "lib"]
