#!/bin/bash
# rocprofv3 kernel trace + stats of one tools/model_bench.py workload: bash tools/trace_model.sh <outdir-under-gpurun_out> <workload> [args]
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W=$1
# (the program's stdout only goes into the families file: rocprofv3 logs to stderr)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$W -o trace -- python3 $R/tools/model_bench.py "$@" > $O/$W.out 2> $O/$W.err || { echo trace failed; tail -5 $O/$W.err; exit 1; }
cp $(find $O/trace_$W -name "*kernel_stats.csv" | head -1) $O/${W}_kernel_stats.csv
grep -v -e "amdgpu.ids" -e "^W2026" -e "^E2026" -e "^I2026" $O/$W.out > $O/${W}_families.txt
head -12 $O/${W}_kernel_stats.csv | cut -c1-150
