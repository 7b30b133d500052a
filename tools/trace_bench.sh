#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command (eager launches): gpurun_out/<dir>/trace_kernel_stats.csv
# usage (on the GPU box, via gpurun): bash tools/trace_bench.sh <outdir-under-gpurun_out> [bench args]
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph "$@" > $O/trace.log 2>&1 || { echo trace failed; tail -5 $O/trace.log; exit 1; }
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
cp $f $O/trace_kernel_stats.csv
python3 - "$O/trace_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"kernels {len(rows)} calls {calls} (per step of 12: {calls/12:.1f}) total {tot/1e6:.2f} ms ({tot/12e6:.3f} ms/step)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print("%-70s calls %5d avg %8.1f us  tot/step %7.3f ms" % (r["Name"][:70], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 12e6))
PY
