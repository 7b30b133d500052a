mkdir -p gpurun_out/r2g
for cfg in "1 0" "0 0" "1 1"; do
  set -- $cfg
  echo "== KLOOP=$1 HYB=$2"
  PB_GEMM_KLOOP=$1 PB_KLOOP_HYB=$2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2g/bench_k$1$2.jsonl 2> gpurun_out/r2g/bench_k.err || { tail -5 gpurun_out/r2g/bench_k.err; exit 1; }
  python - "gpurun_out/r2g/bench_k$1$2.jsonl" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"],d["ms_per_step"])
for k,v in sorted(d["kernel_families"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:6]:
    print("  %-28s n=%3d %7.3f ms  %6.0f GB/s"%(k,v["launches_per_step"],v["ms_per_step"],v["GBps"]))
PY
done
