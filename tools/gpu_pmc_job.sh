#!/bin/bash
# One gpurun call: the three separate --pmc passes (FETCH_SIZE, WRITE_SIZE, GRBM_GUI_ACTIVE) of the bench command,
# kernel trace only beside them (MI355X_MICROARCH.md: never combined with other trace domains).  Reduce afterwards in the
# build container: tools/pmc_traffic.py / tools/pmc_clock.py (they stamp commit + kernel-source hash).
#   gpurun --timeout 900 -- "bash tools/gpu_pmc_job.sh"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-blocks-record --no-x3-record --no-streamed-record > /dev/null 2>&1
  echo "pmc $c rc=$?"
  rm -f $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_$c/*.db
done
