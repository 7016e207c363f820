#!/bin/bash
# One gpurun call: marked kernel traces of the small workloads (config 1: 4,904 points; mid: 125,672 points) cut into phases,
# with the launch sequence of one step — the launch-latency chains that floor small frames and every rank of the
# spatial-block mode.   gpurun --timeout 600 -- "bash tools/gpu_small_frame_job.sh [tag]"
TAG=${1:-r3}
cd /tmp && export TMPDIR=/tmp
for W in config1 mid; do
  PCC_BENCH_MARK=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_small_$W -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 6 --warmup 3 --no-cpu-baseline --no-blocks-record --no-x3-record --no-streamed-record --no-small-frame-record > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_small_$W.json 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_small_$W.err
  echo "trace $W rc=$?"
  rm -f $GRAFT_REPO_ROOT/gpurun_out/${TAG}_small_$W/*.db
  python3 $GRAFT_REPO_ROOT/tools/trace_gaps.py $GRAFT_REPO_ROOT/gpurun_out/${TAG}_small_$W/b_kernel_trace.csv --list $GRAFT_REPO_ROOT/gpurun_out/${TAG}_small_${W}_list.txt > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_small_${W}_gaps.txt 2>&1
  python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-blocks-record --no-x3-record --no-streamed-record --no-small-frame-record 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', j['ms_per_step'], j['t_enc_ms'], j['t_dec_ms'])"
done
