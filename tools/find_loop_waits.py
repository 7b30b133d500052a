"""Which kernels wait for EVERYTHING (s_waitcnt vmcnt(0)) inside a loop?  For kernels that count vmcnt by hand around LDS-DMA
rings this is the signature of a compiler-tracked register load whose wait landed in the loop (the ring stops being a ring).
    python tools/find_loop_waits.py picklebot_amd/lib/asm/*.s"""
import re
import sys

for path in sys.argv[1:]:
    lines = open(path).read().split("\n")
    cur, start = None, 0
    funcs = []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            if cur:
                funcs.append((cur, start, i))
            cur, start = m.group(1), i
    if cur:
        funcs.append((cur, start, len(lines)))
    for name, a, b in funcs:
        body = lines[a:b]
        if not any("global_load_lds" in l for l in body):
            continue
        labels = {}
        for i, l in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = i
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
            if m:
                t = m.group(1) or m.group(2)
                if t in labels and labels[t] < i:
                    loops.append((labels[t], i))
        hits = []
        for lo, hi in loops:
            n0 = sum(1 for l in body[lo:hi] if re.search(r"s_waitcnt.*vmcnt\(0\)", l))
            dma = sum(1 for l in body[lo:hi] if "global_load_lds" in l)
            if n0 and dma:
                hits.append((hi - lo, n0, dma))
        if hits:
            short = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)[:90]
            print("%-24s %-92s loops with DMA and vmcnt(0): %s" % (path.split("/")[-1], short, sorted(set(hits))[:4]))
