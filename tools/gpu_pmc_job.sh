#!/bin/bash
# One gpurun call: the separate --pmc passes of the bench command, kernel trace only beside them (MI355X_MICROARCH.md: never
# combined with other trace domains; the program itself after `--`).  Passes: FETCH_SIZE, WRITE_SIZE (HBM traffic),
# GRBM_GUI_ACTIVE (delivered clock), and the matrix-pipe passes — SQ_VALU_MFMA_BUSY_CYCLES with GRBM_GUI_ACTIVE (the
# MfmaUtil quotient of rocprofiler-sdk's counter_defs.yaml: sum of per-SIMD MFMA-busy cycles / (GUI-active cycles x SIMDs)),
# SQ_INSTS_VALU_MFMA_MOPS_F32 with SQ_BUSY_CYCLES, and the wave-time split SQ_WAVE_CYCLES / SQ_WAIT_ANY /
# SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY.  Reduce afterwards in the build container: tools/pmc_traffic.py, tools/pmc_clock.py,
# tools/pmc_mfma.py (they stamp commit + kernel-source hash).
#   gpurun --timeout 1100 -- "bash tools/gpu_pmc_job.sh [round tag, default r3]"
TAG=${1:-r3}
cd /tmp && export TMPDIR=/tmp
pass() {   # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$name -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-blocks-record --no-x3-record --no-streamed-record --no-small-frame-record --no-mid-frame-record --no-train-record --no-hbm-record --no-live-pmc > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$name.log 2>&1
  rc=$?
  echo "pmc $name rc=$rc"
  rm -f $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$name/*.db
  return $rc                 # (the && chain below stops at the first failed pass: ADVICE r3)
}
pass FETCH_SIZE FETCH_SIZE && pass WRITE_SIZE WRITE_SIZE && pass GRBM_GUI_ACTIVE GRBM_GUI_ACTIVE && \
pass MFMA_BUSY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE && pass MFMA_OPS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES && \
pass WAVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
