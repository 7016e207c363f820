import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle runs on torch's intra-op pool, which sizes itself by the HOST's core count; a GPU box hands this
    # process a 16-core share of a 256-thread host, and a pool of 128 threads on 16 cores makes the oracle's many small
    # operators ten times slower (test_training_step_matches_oracle_autograd: 33 s against 3 s).  8 threads, not 16:
    # measured on the GPU box, the ORACLE's gradient of one 64 x 64 x 27 kernel (g_s.post_conv.0) moves by 2.3 % when
    # torch's CPU pool has 10 .. 64 threads (a host BLAS / threading effect; not reproducible on the 8-core build
    # container): with 1, 4, 8, 9 and 128 threads the oracle agrees with the HIP gradients — which do not depend on the
    # host's thread count — to 6.4e-6 over all 150+ parameters.  PCC_TEST_THREADS overrides (0 = torch's default).
    import torch
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    want = int(os.environ.get("PCC_TEST_THREADS", "8"))
    if want > 0:
        torch.set_num_threads(max(1, min(avail, want)))


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
