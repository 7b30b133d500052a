mkdir -p gpurun_out/r2g
for nf in "" 1; do echo "== NOFLUSH=$nf"; PB_KB_NOFLUSH=$nf timeout -k 10 200 python tools/kbench.py pw_big 2>&1 | grep "pw_fwd" | grep -v "rs=1" || exit 1; done > gpurun_out/r2g/exp_nf.log 2>&1
