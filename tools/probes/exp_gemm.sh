mkdir -p gpurun_out/r2g
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "pw_" > gpurun_out/r2g/gemm.log 2>&1 || { tail -30 gpurun_out/r2g/gemm.log; exit 1; }
tail -3 gpurun_out/r2g/gemm.log
timeout -k 10 300 python tools/kbench.py pw_big pw_small > gpurun_out/r2g/kb_gemm.log 2>&1 || exit 1
grep -v amdgpu.ids gpurun_out/r2g/kb_gemm.log
