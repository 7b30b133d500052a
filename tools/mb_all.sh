#!/bin/bash
# per-family timing of the non-headline workloads (tools/model_bench.py); usage: bash tools/mb_all.sh [workload batch]...
mkdir -p gpurun_out
if [ $# -eq 0 ]; then set -- "mobilevit_s_train 4" "mobilevit_xxs_fwd 8" "movinet_a2_fwd 4" "mobilenet_small_train 32"; fi
for w in "$@"; do
  set -- $w
  timeout -k 10 300 python tools/model_bench.py $1 --batch $2 --steps 3 ${MB_FLAGS:-} --detail gpurun_out/mb_$1.txt > gpurun_out/mb_$1.log 2>&1
  rc=$?; echo "== $1 rc $rc"; tail -n ${TAIL:-14} gpurun_out/mb_$1.log
  if [ $rc -ge 124 ]; then exit $rc; fi
done
