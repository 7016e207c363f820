#!/bin/bash
# One gpurun call: smoke, the default bench line, a kernel-trace + stats profile of the same command, and a marked kernel trace
# cut into phases (profiles/rNN_*).   gpurun --timeout 1100 -- "bash tools/gpu_profile_job.sh [tag]"
TAG=${1:-r3}
set -x
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo rc=$?
tail -2 gpurun_out/${TAG}_bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_stats -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-blocks-record --no-x3-record --no-streamed-record --no-small-frame-record --no-mid-frame-record --no-train-record --no-hbm-record --no-live-pmc > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_under_rocprof.log 2>&1
PCC_BENCH_MARK=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_marked -o b -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-blocks-record --no-x3-record --no-streamed-record --no-small-frame-record --no-mid-frame-record --no-train-record --no-hbm-record --no-live-pmc > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py gpurun_out/${TAG}_prof_marked/b_kernel_trace.csv --json gpurun_out/${TAG}_gaps.json > gpurun_out/${TAG}_gaps.txt 2>&1
rm -f gpurun_out/${TAG}_prof_stats/*.db gpurun_out/${TAG}_prof_marked/*.db gpurun_out/${TAG}_prof_stats/b_kernel_trace.csv
grep "^==" gpurun_out/${TAG}_gaps.txt
