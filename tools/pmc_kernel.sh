#!/bin/bash
# SQ activity counters of one program's kernels (one PMC pass, kernel-trace only; the program goes after `--`):
#   tools/pmc_kernel.sh <outdir-name> <kernel-substring> python3 <script> [args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; K=$2; shift 2; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O -o pmc -- "$@" > $O/run.log 2>&1 || { echo failed; tail -8 $O/run.log; exit 1; }
cd $R && python3 - "$O" "$K" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if sys.argv[2] not in k: continue
    k = k[:70]
    d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in d.items():
    wc = c["SQ_WAVE_CYCLES"]
    print(k, "calls", n[k])
    for name in ("SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
        print("   %-22s %.3e  (%.1f %% of wave cycles)" % (name, c[name], 100 * c[name] / max(wc, 1)))
    print("   SQ_WAVE_CYCLES         %.3e" % wc)
PY
