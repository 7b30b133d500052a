mkdir -p gpurun_out/r2g
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "dw" > gpurun_out/r2g/dw.log 2>&1 || { tail -30 gpurun_out/r2g/dw.log; exit 1; }
tail -2 gpurun_out/r2g/dw.log
timeout -k 10 300 python tools/kbench.py dw > gpurun_out/r2g/kb_dw.log 2>&1 || exit 1
grep -v amdgpu.ids gpurun_out/r2g/kb_dw.log
