#!/bin/bash
# Round-3 evidence, collected on the MI355X box via gpurun (run from the repo root):
#   profiles/collect_r03.sh <tag> [workload] [dtype] [extra bench flags...]
# Kernel trace and PMC counters are separate runs; each --pmc pass carries one counter; the program follows `--` directly.
# PMC=0 skips the counter passes (kernel trace + bench line only).
set -e
export TMPDIR=/tmp
TAG=${1:-a}
WL=${2:-cfg3}
DT=${3:-bf16}
shift 3 || true
EXTRA="$@"
O=gpurun_out/r03_${TAG}_${WL}_${DT}
mkdir -p $O
python bench.py --workload $WL --dtype $DT $EXTRA > $O/bench.json 2> $O/bench.err
tail -c 700 $O/bench.json
STEPS=${STEPS:-40}
WARM=${WARM:-10}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 bench.py --workload $WL --dtype $DT --no-cpu-baseline --steps $STEPS --warmup $WARM $EXTRA > $O/trace.log 2>&1
python profiles/tools/kernel_summary.py $O/trace $((STEPS + WARM + 1)) > $O/kernel_summary.txt
cp $(find $O/trace -name '*kernel_stats.csv' | head -n1) $O/kernel_stats.csv
if [ "${PMC:-1}" = "1" ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 bench.py --workload $WL --dtype $DT --no-cpu-baseline --steps 6 --warmup 2 $EXTRA > $O/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 bench.py --workload $WL --dtype $DT --no-cpu-baseline --steps 6 --warmup 2 $EXTRA > $O/pmc_write.log 2>&1
  python profiles/tools/pmc_by_kernel.py $O/pmc_fetch $O/pmc_write $O/pmc_by_kernel.json ${DOMINANT:-gemm_nt_wide_kernel} $O/pmc_dominant.json > $O/pmc_by_kernel.txt
  python profiles/tools/pmc_for_bench.py $O/pmc_by_kernel.json $O/pmc_for_bench.json >> $O/pmc_by_kernel.txt   # -> profiles/r03_pmc_<workload>_<dtype>.json
fi
rm -rf $O/trace $O/pmc_fetch $O/pmc_write   # raw traces are large; the summaries above are what gets committed
head -n 52 $O/kernel_summary.txt
