#!/bin/bash
# Round-4 evidence run (GPU box): model benches with the final build, gradient-error distributions, SQ counters of the step,
# attention bench, forced data-parallel bench.  Outputs under gpurun_out/r4ev/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ev; mkdir -p $O; cd $R
for w in "mobilevit_s_train 8" "mobilevit_xxs_fwd 16" "mobilevit_xxs_fwd 64" "movinet_a2_fwd 6" "movinet_a2_train 4" "mobilenet_small_train 32"; do
  set -- $w
  timeout -k 10 300 python tools/model_bench.py $1 --batch $2 --steps 5 --graph > $O/mb_$1_b$2.txt 2>&1; echo "== $1 b$2 rc $?"; grep "replayed" $O/mb_$1_b$2.txt
done
timeout -k 10 600 python tools/grad_err.py --bf16 > $O/grad_err_bf16.txt 2>&1; echo "grad_err bf16 rc $?"
timeout -k 10 600 python tools/grad_err.py > $O/grad_err_fp32.txt 2>&1; echo "grad_err fp32 rc $?"
timeout -k 10 300 python tools/attn_bench.py > $O/attention_bench.txt 2>&1; echo "attn rc $?"
PB_BENCH_FORCE_DP=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_force_dp.jsonl 2> $O/bench_force_dp.err; echo "dp rc $?"
timeout -k 10 400 bash tools/pmc_sq.sh > $O/pmc_sq.log 2>&1; echo "sq rc $?"; cp gpurun_out/pmc_sq/sq_activity.txt $O/sq_activity.txt 2>/dev/null
timeout -k 10 300 python tools/probes/dw_staircase.py 672 16 14 14 3 1 > $O/dw_staircase_default.txt 2>&1
PB_DW_RING_ALL=1 timeout -k 10 300 python tools/probes/dw_staircase.py 672 16 14 14 3 1 > $O/dw_staircase_ring.txt 2>&1
echo done
