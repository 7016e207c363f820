#!/bin/bash
# One gpurun call: the training step (BASELINE config 5 shape on one GPU) in fp32 and with bf16 operands, then a kernel-stats
# profile of the bf16 step (profiles/rNN_train_step_bf16_kernel_stats.csv).
#   gpurun --timeout 900 -- "bash tools/gpu_train_profile_job.sh"
TAG=${1:-r3}
set -x
python tools/train_bench.py --batch 8 --block 256 > gpurun_out/${TAG}_train_f32.json 2> gpurun_out/${TAG}_train_f32.err; echo rc=$?
PCC_TRAIN_BF16=1 python tools/train_bench.py --batch 8 --block 256 > gpurun_out/${TAG}_train_bf16.json 2> gpurun_out/${TAG}_train_bf16.err; echo rc=$?
tail -1 gpurun_out/${TAG}_train_f32.json; tail -1 gpurun_out/${TAG}_train_bf16.json
cd /tmp && export TMPDIR=/tmp
export PCC_TRAIN_BF16=1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_train_prof -o t -- python3 $GRAFT_REPO_ROOT/tools/train_bench.py --batch 8 --block 256 --steps 5 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/${TAG}_train_prof/t_results.db gpurun_out/${TAG}_train_prof/t_kernel_trace.csv
head -30 gpurun_out/${TAG}_train_prof/t_kernel_stats.csv | cut -c1-160
