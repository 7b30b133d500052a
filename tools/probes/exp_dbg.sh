mkdir -p gpurun_out/r2g
for d in 0 1 2 4 8 3 11 15; do echo "== DBG=$d"; PB_GEMM_DMA_DBG=$d timeout -k 10 200 python tools/kbench.py pw_big 2>&1 | grep "pw_fwd" | grep "K672 N112 trans=0 rs=0\|K960" || exit 1; done > gpurun_out/r2g/exp_dbg.log 2>&1
