#!/bin/bash
# Instruction-cache behaviour per kernel on the bench step (one PMC pass, kernel-trace only; the program goes after `--`).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_icache; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O -o ic -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $O/run.log 2>&1 || { echo failed; tail -8 $O/run.log; exit 1; }
cd $R && python3 - "$O" <<'PY'
import csv, glob, sys, collections
sys.path.insert(0, "tools")
from pmc_summary import base_name
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = base_name(r["Kernel_Name"]); d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQC_ICACHE_REQ": n[k] += 1
print("# rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- python3 bench.py --steps 2 --warmup 1 --no-graph")
print("%-30s %6s %14s %14s %9s %16s" % ("kernel", "calls", "icache_req", "icache_miss", "miss %", "miss/wave_cycle"))
for k, c in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:30]:
    req, miss = c.get("SQC_ICACHE_REQ", 0), c.get("SQC_ICACHE_MISSES", 0)
    print("%-30s %6d %14.3e %14.3e %8.2f%% %16.3e" % (k, n[k], req, miss, 100 * miss / max(req, 1), miss / max(c.get("SQ_WAVE_CYCLES", 1), 1)))
PY
