# A/B the dispatch knobs on the bench step (one line per setting: clips/s, ms/step)
mkdir -p gpurun_out/r2h
run() { echo -n "$1: "; env $1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])" || exit 1; }
run "PB_NOP=1"
run "PB_EW_SUM_ROWS=8"
run "PB_EW_SUM_ROWS=32"
run "PB_WGRAD_NS=4"
run "PB_WGRAD_NS=6"
run "PB_WGRAD_SMALL_TILES=8"
run "PB_WGRAD_SMALL_NS=4"
run "PB_WGRAD_WS_MINROWS=1024"
run "PB_WGRAD_WS_MINROWS=256"
run "PB_GEMM_STREAM_CG=4"
run "PB_GEMM_STREAM_CG=8"
run "PB_GEMM_WIDE=0"
run "PB_DW_ROWS_NARROW=0"
run "PB_DW_ROWS_NARROW=5"
run "PB_NOP=2"
