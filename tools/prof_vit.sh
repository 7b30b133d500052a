cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_vit
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_vit -o vit -- python3 $GRAFT_REPO_ROOT/tools/model_bench.py mobilevit_s_train --batch 8 --steps 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_vit/run.log 2>&1
echo rc $?
find $GRAFT_REPO_ROOT/gpurun_out/prof_vit -name "*kernel_stats*" | head
