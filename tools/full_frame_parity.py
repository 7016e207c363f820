#!/usr/bin/env python3
"""The stage-by-stage codec comparison of tests/_parity.py (encoder decisions / decoder on identical latents / end to end)
on the WHOLE config-2 frame (N = 850,824): the oracle needs ~2.5 minutes of the GPU box's host cores, so this is a
profile run, not a test.   python tools/full_frame_parity.py out.json [threads]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
torch.set_num_threads(threads)
import pcc_amd
from oracle.codec import Codec
from _parity import compare_codec
dev = "cuda:0"
syn = pcc_amd.synthetic
model = syn.make_model(0, dev); model.update()
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
codec = Codec(sd); codec.update()
pts = syn.sphere_shell(**syn.CONFIG2)
qc, qf = syn.uniform_qmap(pts[:, :3], 0.5, 0.5)
t0 = time.time()
r = compare_codec(pcc_amd, model, codec, pts, qc, qf, "config 2", dev)
out = {"frame": "config 2: 1024^3 shell, N=%d, q=(0.5,0.5), seeded weights" % pts.shape[0], "oracle_threads": threads,
       "seconds": time.time() - t0, "bpp": {"hip": float(r["bpp"]), "oracle": float(r["o_bpp"])}, "streams_byte_equal": bool(r["streams_equal"]),
       "latents_rounded_differently": int(r["n_sym"]),
       "voxels_differing": {"hip_decoder_on_oracle_latents": int(r["flips_same"]), "own_streams": int(r["flips"])},
       "d1_psnr_db": {"hip": float(r["m"]["sym_psnr_mse"]), "oracle": float(r["om"]["sym_psnr_mse"])},
       "y_psnr_db": {"hip": float(r["m"]["sym_y_psnr"]), "oracle": float(r["om"]["sym_y_psnr"])},
       "note": "every assertion of tests/_parity.py:compare_codec held (structure, k, latent coordinates exact; bpp 2e-3; latents; "
               "decoder on identical latents within 1e-3 dB + voxel-flip bound; end to end)"}
print(json.dumps(out, indent=1))
json.dump(out, open(sys.argv[1], "w"), indent=1)
