#!/bin/bash
# rocprofv3 kernel trace of the bench command in HIP-graph mode (the timed configuration): gpurun_out/<dir>/trace_kernel_trace.csv
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o trace -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $O/trace.log 2>&1 || { echo trace failed; tail -5 $O/trace.log; exit 1; }
cp $(find $O/trace -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv
python3 $R/tools/graph_streams.py $O/kernel_trace.csv 7
python3 $R/tools/graph_timeline.py $O/kernel_trace.csv 7
