#!/bin/bash
# Round-1 evidence, collected on the MI355X box via gpurun (run from the repo root).  Kernel trace and PMC counters are
# separate runs; each --pmc pass carries one counter.
set -e
export TMPDIR=/tmp
O=gpurun_out/final
mkdir -p $O
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --workload cfg3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --workload cfg5 --no-cpu-baseline --steps 40 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python bench.py --workload a5 --steps 40 > $O/bench_a5.json 2> $O/bench_a5.err
python bench.py --graph --no-cpu-baseline > $O/bench_cfg2_graph.json 2> $O/bench_cfg2_graph.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 bench.py --no-cpu-baseline > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/pmc_write.log 2>&1
python profiles/tools/pmc_summary.py $O/pmc_fetch $O/pmc_write gemm_nt_kernel $O/pmc_gemm_nt.json
python profiles/tools/kernel_summary.py $O/trace 121 > $O/kernel_summary.txt
cp $(find $O/trace -name '*kernel_stats.csv' | head -n1) $O/kernel_stats.csv
rm -rf $O/trace/*kernel_trace.csv $O/pmc_fetch $O/pmc_write   # raw traces are large; the summaries above are what gets committed
tail -n 3 $O/kernel_summary.txt
