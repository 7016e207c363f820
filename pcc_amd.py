"""Import alias: ``import pcc_amd`` loads the package that lives in the directory
``learned-compression-of-point-cloud-geometry-and-attributes_amd/`` (a name the build contract
fixes but Python cannot import directly because of the hyphens)."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "learned-compression-of-point-cloud-geometry-and-attributes_amd")
_spec = importlib.util.spec_from_file_location("pcc_amd", os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["pcc_amd"] = _mod
_spec.loader.exec_module(_mod)
