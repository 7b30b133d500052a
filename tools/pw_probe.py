"""Phase timers of the streaming GEMM (library built with PB_BUILD_DEFINES=-DPB_STREAM_PROBE): one launch per layer shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picklebot_amd import kernels as K
dev = "cuda:0"
for M, Kd, N, stats in [(100352, 112, 672, False), (3211264, 16, 64, False), (23520, 160, 960, False), (602112, 24, 72, False),
                        (250880, 40, 120, False), (602112, 72, 24, True), (2007040, 64, 16, False), (29792, 160, 960, True)]:
    x = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
    w = torch.randn(N, Kd, device=dev)
    st = torch.zeros(K.REPL, 2, N, device=dev) if stats else None
    for _ in range(2):
        y = K.pw_fwd(x, w, stats=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = K.pw_fwd(x, w, stats=st); e1.record(); torch.cuda.synchronize()
    print("M%d K%d N%d stats=%d: %.1f us" % (M, Kd, N, stats, e0.elapsed_time(e1) * 1e3), flush=True)
