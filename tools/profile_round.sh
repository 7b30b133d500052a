#!/bin/bash
# Refresh the judged profiles of the bench command (run on the GPU box through gpurun):
#   kernel trace + stats (CSV), then FETCH_SIZE and WRITE_SIZE in separate PMC passes, then the default bench line.
# Outputs under gpurun_out/prof_round/; copy the summaries into profiles/ afterwards (tools/pmc_summary.py).
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_round
rm -rf $O && mkdir -p $O
# Kernel durations IN LINE (PB_WGRAD_OVERLAP=0: every kernel alone on the chip - what bench.py's HIP-event pass measures and
# what `roofline` is about) and as the timed steps run them (each block's weight gradients on a second stream beside the next
# block's chain: two kernels share the chip, single durations stretch, the step shrinks).  The PMC passes count bytes: in line.
export PB_WGRAD_OVERLAP=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph > $O/trace.log 2>&1 || { echo trace failed; tail -5 $O/trace.log; exit 1; }
echo "trace (in line) done"
export PB_WGRAD_OVERLAP=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_overlap -o trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph > $O/trace_overlap.log 2>&1 || { echo overlap trace failed; tail -5 $O/trace_overlap.log; exit 1; }
echo "trace (side stream) done"
export PB_WGRAD_OVERLAP=0
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $O/fetch.log 2>&1 || { echo fetch failed; tail -5 $O/fetch.log; exit 1; }
echo "fetch done"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $O/write.log 2>&1 || { echo write failed; tail -5 $O/write.log; exit 1; }
echo "write done"
cd $R
python3 tools/pmc_summary.py $(find $O/fetch -name "*counter_collection.csv" | head -1) $(find $O/write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > $O/pmc_summary.txt || { echo pmc summary failed; exit 1; }
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cp $(find $O/trace_overlap -name "*kernel_stats.csv" | head -1) $O/kernel_stats_overlap.csv
unset PB_WGRAD_OVERLAP
PB_PMC_TRAFFIC=$O/pmc_traffic.json PB_BENCH_DETAIL=$O/per_shape.txt timeout -k 10 600 python bench.py > $O/bench.jsonl 2> $O/bench.err || { echo bench failed; tail -5 $O/bench.err; exit 1; }
tail -c 600 $O/bench.jsonl
find $O -name "*.csv" | head -20
du -sh $O
