mkdir -p gpurun_out/r2g
for ns in 0 3 4; do echo "== PB_GEMM_DMA_NS=$ns"; PB_GEMM_DMA_NS=$ns timeout -k 10 200 python tools/kbench.py pw_big 2>&1 | grep "K672\|K960" || exit 1; done > gpurun_out/r2g/exp1.log 2>&1
