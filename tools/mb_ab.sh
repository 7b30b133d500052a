#!/bin/bash
# A/B of env knobs on a tools/model_bench.py workload: bash tools/mb_ab.sh "<workload> <batch>" <grep pattern> "ENV=1 ..." "ENV=2 ..."
mkdir -p gpurun_out
wl=$1; pat=$2; shift 2
i=0
for cfg in "$@"; do
  set -- $wl
  env $cfg timeout -k 10 300 python tools/model_bench.py $1 --batch $2 --steps 3 --detail gpurun_out/mbab_$i.txt > gpurun_out/mbab_$i.log 2>&1
  rc=$?
  echo "== [$cfg] rc $rc $(grep 'ms/step eager' gpurun_out/mbab_$i.log | sed 's/.*): //')"
  grep -E "$pat" gpurun_out/mbab_$i.log | head -8
  grep -E "$pat" gpurun_out/mbab_$i.txt | head -${DETAIL:-16}
  if [ $rc -ge 124 ]; then exit $rc; fi
  i=$((i+1))
done
