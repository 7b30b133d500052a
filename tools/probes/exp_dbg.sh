mkdir -p gpurun_out/r2g
for d in 0 1 4 5 13; do echo "== DBG=$d"; PB_GEMM_DMA_DBG=$d timeout -k 10 200 python tools/kbench.py pw_big pw_small 2>&1 | grep "pw_fwd" | grep -v "rs=1" || exit 1; done > gpurun_out/r2g/exp_dbg.log 2>&1
