"""The BLAS-order oracle on the config-2 frame (N = 850,824) as a background process of the GPU test session.

The oracle needs ~2.5 minutes of host cores for this frame and nothing from the GPU; tests/conftest.py starts this script when
the session begins (only when tests/test_hip_codec.py::test_full_config2_frame_vs_oracle is selected on a GPU box), so that the
oracle runs beside the first hundred tests instead of in front of the hundred-and-fifth.  CPU only: the GPU is hidden from this
process.  Output: one .npz with everything tests/_parity.compare_codec reads from the oracle.  Test infrastructure only."""
import os
import sys

os.environ["HIP_VISIBLE_DEVICES"] = ""
os.environ["CUDA_VISIBLE_DEVICES"] = ""
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main(out_path, threads):
    torch.set_num_threads(threads)
    import pcc_amd
    from oracle.codec import Codec
    syn = pcc_amd.synthetic
    model = syn.make_model(seed=0, device="cpu")
    codec = Codec({k: v.detach().cpu() for k, v in model.state_dict().items()})
    codec.update()
    pts = syn.sphere_shell(**syn.CONFIG2)
    qc, qf = syn.uniform_qmap(pts[:, :3], 0.5, 0.5)
    strings, shape, k, coords = codec.compress(pts, qc, qf)
    rec = codec.decompress(coords, strings, shape, k)
    y, Q = codec.last_dec["y_hat"], codec.last_dec["Q_hat"]
    tmp = out_path + ".tmp.npz"
    np.savez(tmp, y_stream=np.frombuffer(strings[0][0], dtype=np.uint8), z_stream=np.frombuffer(strings[1][0], dtype=np.uint8),
             shape=np.asarray(shape, dtype=np.int64), k=np.asarray(k, dtype=np.int64), coords=coords, rec=rec,
             y_C=y.C, y_F=y.F.numpy(), Q_C=Q.C, Q_F=Q.F.numpy(), threads=np.asarray([threads]))
    os.replace(tmp, out_path)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8)
