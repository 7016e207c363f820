#!/bin/bash
# One gpurun call: matrix-pipe counters of the TRAINING step (tools/train_bench.py, fp32 and bf16 operands): separate rocprofv3 --pmc
# passes with --kernel-trace, the program itself after `--`.  Reduce with tools/pmc_mfma.py <dir> <tag> out.json.
#   gpurun --timeout 900 -- "bash tools/gpu_train_pmc_job.sh [tag]"
TAG=${1:-r3tr}
cd /tmp && export TMPDIR=/tmp
pass() {   # name, env assignment or "-", counters...
  name=$1; shift; mode=$1; shift
  if [ "$mode" = "bf16" ]; then export PCC_TRAIN_BF16=1; else unset PCC_TRAIN_BF16; fi
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_${mode}_pmc_$name -o p -- python3 $GRAFT_REPO_ROOT/tools/train_bench.py --batch 8 --block 256 --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_${mode}_pmc_$name.log 2>&1
  echo "pmc $mode $name rc=$?"
  rm -f $GRAFT_REPO_ROOT/gpurun_out/${TAG}_${mode}_pmc_$name/*.db
}
pass MFMA_BUSY f32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE && pass WAVE f32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY && \
pass MFMA_BUSY bf16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE && pass WAVE bf16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
