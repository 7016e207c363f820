import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def pcc():
    import pcc_amd
    return pcc_amd


@pytest.fixture(scope="session")
def seeded_state_dict(pcc):
    """state_dict of the seeded ColorModel (CPU tensors)."""
    model = pcc.synthetic.make_model(seed=0, device="cpu")
    return {k: v.detach().cpu() for k, v in model.state_dict().items()}


@pytest.fixture(scope="session")
def oracle_codec(seeded_state_dict):
    from oracle.codec import Codec
    codec = Codec(seeded_state_dict)
    codec.update()
    return codec
