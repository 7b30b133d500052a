#!/bin/bash
# Run GPU steps in order, logging each to gpurun_out/; ordinary failures (exit 1) do not stop
# the sequence, a timeout/kill (124/137/139...) does.
mkdir -p gpurun_out
run() {
  local name=$1; shift
  echo "=== $name: $*"
  timeout -k 10 "${STEP_TIMEOUT:-900}" "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "=== $name exit $rc"; tail -n "${TAIL:-12}" gpurun_out/$name.log
  if [ $rc -ge 124 ]; then echo "stopping after $name (rc $rc)"; exit $rc; fi
  return 0
}
for step in "$@"; do
  case $step in
    kernels) run kernels python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider ;;
    models) run models python -m pytest tests/test_models_gpu.py -q -m gpu -p no:cacheprovider ;;
    alltests) run alltests python -m pytest tests -q -m gpu -p no:cacheprovider ;;
    smoke) run smoke python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench) PB_BENCH_DETAIL=gpurun_out/bench_detail.txt run bench python bench.py --steps ${BENCH_STEPS:-10} --warmup 3 ;;
    bench_small) run bench_small python bench.py --steps 5 --warmup 2 --batch 8 --no-cpu-baseline ;;
    *) echo "unknown step $step" ;;
  esac
done
