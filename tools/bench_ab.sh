#!/bin/bash
# A/B environment settings on the real training step: tools/bench_ab.sh "A=1" "B=2 C=3" ...
# each run: bench.py --steps 6 --warmup 2, per-shape detail kept in gpurun_out/ab_<i>.txt
mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  echo "== [$i] $cfg"
  env $cfg PB_BENCH_DETAIL=gpurun_out/ab_$i.txt timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$i.log 2>&1 || { echo "run $i failed"; tail -5 gpurun_out/ab_$i.log; exit 1; }
  python - <<PY
import json
for l in open('gpurun_out/ab_$i.log'):
    if l.startswith('{'):
        d = json.loads(l)
        fam = d['kernel_families']
        print('   %.1f clips/s  %.3f ms/step  host %.2f ms' % (d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step', 0)))
        print('   ' + '  '.join('%s %.2f' % (k.replace('_kernel', ''), v['ms_per_step']) for k, v in list(fam.items())[:8]))
PY
  i=$((i+1))
done
