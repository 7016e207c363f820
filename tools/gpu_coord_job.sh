#!/bin/bash
# One gpurun call: the coordinate-side micro-benchmark (tools/coord_bench.py: config-2-sized candidate sets) plain, under
# rocprofv3 --kernel-trace --stats, and under separate --pmc passes (kernel trace only beside them; the program itself after --).
#   gpurun --timeout 900 -- "bash tools/gpu_coord_job.sh [tag]"
TAG=${1:-r4}
python tools/coord_bench.py > gpurun_out/${TAG}_coord_bench.txt 2>&1; echo "coord_bench rc=$?"; cat gpurun_out/${TAG}_coord_bench.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_coord_stats -o c -- python3 $GRAFT_REPO_ROOT/tools/coord_bench.py > /dev/null 2>&1; echo "stats rc=$?"
pass() {   # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_coord_pmc_$name -o p -- python3 $GRAFT_REPO_ROOT/tools/coord_bench.py > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_coord_pmc_$name.log 2>&1
  rc=$?
  echo "pmc $name rc=$rc"
  rm -f $GRAFT_REPO_ROOT/gpurun_out/${TAG}_coord_pmc_$name/*.db
  return $rc
}
pass FETCH FETCH_SIZE && pass WRITE WRITE_SIZE && pass TCC TCC_HIT_sum TCC_MISS_sum && pass TCCREQ TCC_REQ_sum TCC_EA0_RDREQ_sum && \
pass WAVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY && pass BUSY GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES && \
pass INST SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU && pass TCP TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum
rm -f $GRAFT_REPO_ROOT/gpurun_out/${TAG}_coord_stats/*.db
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/${TAG}_coord_pmc_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
        for (kn, cn), (n, v) in sorted(agg.items()):
            if any(s in kn for s in ("kernel_map27", "radix_scatter", "radix_count", "unique_insert", "group_masks", "mask_bit")):
                print(f"{kn:62s} {cn:24s} launches {n:4d}  mean {v / n:16.1f}")
PY
