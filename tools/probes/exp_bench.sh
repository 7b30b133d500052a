mkdir -p gpurun_out/r2g
PB_BENCH_DETAIL=gpurun_out/r2g/per_shape.txt timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2g/bench_now.jsonl 2> gpurun_out/r2g/bench_now.err || { tail -5 gpurun_out/r2g/bench_now.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2g/bench_now.jsonl").read().strip().splitlines()[-1])
print(d["value"],d["ms_per_step"])
for k,v in sorted(d["kernel_families"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:16]:
    print("%-28s n=%3d %7.3f ms  %6.0f GB/s"%(k,v["launches_per_step"],v["ms_per_step"],v["GBps"]))
PY
