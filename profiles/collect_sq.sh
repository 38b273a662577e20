#!/bin/bash
# SQ counter passes of one bench command (which pipe bounds each kernel): profiles/collect_sq.sh <tag> [workload] [dtype]
export TMPDIR=/tmp
TAG=${1:-a}; WL=${2:-cfg3}; DT=${3:-bf16}
O=gpurun_out/r03_sq_${TAG}_${WL}_${DT}
mkdir -p $O
rocprofv3 -L > $O/counters_available.txt 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o run -- python3 bench.py --workload $WL --dtype $DT --no-cpu-baseline --no-f32-leg --steps 4 --warmup 2 > $O/p$i.log 2>&1 || echo "pass $i ($set) failed"
done
python profiles/tools/sq_counters.py $O/sq_by_kernel.json $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6 > $O/sq_by_kernel.txt
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6
head -n 120 $O/sq_by_kernel.txt
