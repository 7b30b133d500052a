mkdir -p gpurun_out/r2g
for h in 0 1; do echo "== HYB=$h"; PB_KLOOP_HYB=$h timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "dma_ring or training_projection or residual" 2>&1 | tail -2; PB_KLOOP_HYB=$h timeout -k 10 200 python tools/kbench.py pw_big 2>&1 | grep "pw_fwd" | grep "K672\|K960" || exit 1; done > gpurun_out/r2g/exp_hyb.log 2>&1
