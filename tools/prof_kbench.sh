#!/bin/bash
# kernel trace of a tools/kbench.py case: bash tools/prof_kbench.sh <case>
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_kb; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -o kb -- python3 $GRAFT_REPO_ROOT/tools/kbench.py $1 > $O/run.log 2>&1
echo rc $?; ls $O
