#!/bin/bash
# One gpurun call: smoke, the default bench line, and a marked kernel trace cut into phases (profiles/rNN_*).
#   gpurun --timeout 900 -- "bash tools/gpu_profile_job.sh"
set -x
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r2_bench_default.json 2> gpurun_out/r2_bench_default.err; echo rc=$?
tail -2 gpurun_out/r2_bench_default.err
cd /tmp && export TMPDIR=/tmp
PCC_BENCH_MARK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_prof5 -o b -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-blocks-record --no-x3-record --no-streamed-record > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py gpurun_out/r2_prof5/b_kernel_trace.csv --json gpurun_out/r2_gaps5.json > gpurun_out/r2_gaps5.txt 2>&1
rm -f gpurun_out/r2_prof5/b_results.db
grep "^==" gpurun_out/r2_gaps5.txt
