#!/bin/bash
# SQ activity counters per kernel on the bench step (one PMC pass, kernel-trace only): where do the waves spend time?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_sq; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O -o sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $O/run.log 2>&1 || { echo failed; tail -5 $O/run.log; exit 1; }
cd $R && python3 tools/pmc_sq_summary.py $(find $O -name "*counter_collection.csv" | head -1) > $O/sq_activity.txt && head -30 $O/sq_activity.txt
