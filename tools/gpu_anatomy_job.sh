#!/bin/bash
# One gpurun call: marked kernel traces (PCC_BENCH_MARK=1) of the config-2 frame and of the mid-size frame (125,672 points), cut
# into encode / decode windows by tools/trace_gaps.py; plus rocprofv3 --stats of the config-2 command.
#   gpurun --timeout 900 -- "bash tools/gpu_anatomy_job.sh [tag]"
TAG=${1:-r4}
OFF="--no-cpu-baseline --no-blocks-record --no-x3-record --no-streamed-record --no-small-frame-record --no-mid-frame-record --no-train-record --no-hbm-record --no-live-pmc"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_stats -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 $OFF > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_under_rocprof.log 2>&1; echo "stats rc=$?"
PCC_BENCH_MARK=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_marked -o b -- python3 $GRAFT_REPO_ROOT/bench.py $OFF > /dev/null 2>&1; echo "marked rc=$?"
PCC_BENCH_MARK=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_marked_mid -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload mid --steps 8 --warmup 3 $OFF > /dev/null 2>&1; echo "marked mid rc=$?"
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py gpurun_out/${TAG}_prof_marked/b_kernel_trace.csv --json gpurun_out/${TAG}_gaps.json > gpurun_out/${TAG}_gaps.txt 2>&1
python tools/trace_gaps.py gpurun_out/${TAG}_prof_marked_mid/b_kernel_trace.csv --skip-steps 3 --json gpurun_out/${TAG}_gaps_mid.json --list gpurun_out/${TAG}_mid_launch_sequence.txt > gpurun_out/${TAG}_gaps_mid.txt 2>&1
rm -f gpurun_out/${TAG}_prof_stats/*.db gpurun_out/${TAG}_prof_marked/*.db gpurun_out/${TAG}_prof_marked_mid/*.db gpurun_out/${TAG}_prof_stats/b_kernel_trace.csv
grep "^==" gpurun_out/${TAG}_gaps.txt gpurun_out/${TAG}_gaps_mid.txt
