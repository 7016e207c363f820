import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle runs on torch's intra-op pool, which sizes itself by the HOST's core count; a GPU box hands this
    # process a 16-core share of a much larger host, and a pool of hundreds of threads on 16 cores makes the oracle's
    # many small operators several times slower.  Same rule as bench.py's cpu_baseline leg.
    import torch
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, 16)))


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
