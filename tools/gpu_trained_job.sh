#!/bin/bash
# One gpurun call: train for ~2 minutes on synthetic cubes (bf16 operands), then the bench with those weights
# (fp32 codec; the decoder's candidate sets are then the true surface's children, not the sparse sets of seeded weights).
#   gpurun --timeout 900 -- "bash tools/gpu_trained_job.sh"
TAG=${1:-r3}
set -x
python tools/train.py --steps 1500 --bf16 --log-every 250 --out /tmp/w2min.pt 2>&1 | tail -8
PCC_BENCH_TOP=14 python bench.py --weights /tmp/w2min.pt --no-cpu-baseline --no-blocks-record --no-train-record --no-mid-frame-record --no-hbm-record --no-live-pmc --breakdown > gpurun_out/${TAG}_bench_trained.json 2> gpurun_out/${TAG}_bench_trained.err
grep -E "conv_mfma|conv total|rows" gpurun_out/${TAG}_bench_trained.err | head -24
TAG=$TAG python - <<'PY'
import json, os
j = json.loads(open("gpurun_out/%s_bench_trained.json" % os.environ["TAG"]).read().strip().splitlines()[-1])     # this run's file, not a glob
print({k: j[k] for k in ("value", "ms_per_step", "t_enc_ms", "t_dec_ms", "bpp")})
print(j["roofline"]["frac"], j["roofline"]["achieved"], j["roofline"]["kernel"])
print(j["split_bf16"]["value"], j["split_bf16"]["ms_per_step"], j["split_bf16"]["vs_f32_same_frame"])
PY
