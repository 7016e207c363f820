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


_CONFIG2_WORKER = {}


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        # the BLAS-order oracle on the config-2 frame (2.5 minutes of host cores, nothing from the GPU) starts NOW, as a
        # background process, when its test is part of the session: it then runs beside the tests in front of it
        if any("test_full_config2_frame_vs_oracle" in it.nodeid for it in items) and len(items) > 20 and not _CONFIG2_WORKER:
            import subprocess
            import tempfile
            out = os.path.join(tempfile.mkdtemp(prefix="pcc_config2_"), "blas_oracle.npz")
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            threads = max(1, min(8, avail // 2))
            _CONFIG2_WORKER["out"] = out
            _CONFIG2_WORKER["err"] = open(out + ".stderr", "wb")
            _CONFIG2_WORKER["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_config2_blas_worker.py"), out,
                                                        str(threads)], stdout=subprocess.DEVNULL, stderr=_CONFIG2_WORKER["err"])
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


def pytest_sessionfinish(session, exitstatus):
    proc = _CONFIG2_WORKER.get("proc")
    if proc is not None and proc.poll() is None:
        proc.kill()


@pytest.fixture(scope="session")
def config2_blas_reference():
    """what the background worker computed (tests/_config2_blas_worker.py), or None when it was not started or failed —
    the test then runs the oracle itself"""
    proc = _CONFIG2_WORKER.get("proc")
    if proc is None:
        return None
    try:
        proc.wait(timeout=900)
    except Exception:
        proc.kill()
        return None
    if proc.returncode != 0 or not os.path.exists(_CONFIG2_WORKER["out"]):
        try:
            with open(_CONFIG2_WORKER["out"] + ".stderr", "rb") as f:
                print("config-2 oracle worker failed:", f.read().decode(errors="replace")[-2000:])
        except OSError:
            pass
        return None
    import numpy as np
    z = np.load(_CONFIG2_WORKER["out"])
    return {k: z[k] for k in z.files}
