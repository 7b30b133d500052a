"""Per-basic-block instruction mix of one kernel in a -save-temps .s file (where does a wave spend its issue slots?).
    python tools/isa_blocks.py file.s kernel_symbol_substring [min_instructions]
"""
import collections
import re
import sys


def main():
    asm = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2]
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    inside, blocks, cur = False, [], None
    for l in asm:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            inside = want in m.group(1)
            if inside:
                print("kernel", m.group(1))
                cur = ("entry", [])
            continue
        if not inside:
            continue
        if l.startswith(".Lfunc_end"):
            blocks.append(cur)
            break
        lm = re.match(r"^(\.LBB\d+_\d+):", l)
        if lm:
            blocks.append(cur)
            cur = (lm.group(1), [])
        elif l.startswith("\t") and not l.strip().startswith((".", ";")):
            cur[1].append(l.strip())
    for name, ins in blocks:
        c = collections.Counter()
        for i in ins:
            op = i.split()[0]
            if op.startswith("v_mfma"): c["mfma"] += 1
            elif op.startswith("v_accvgpr"): c["acc_mov"] += 1
            elif op.startswith("v_"): c["valu"] += 1
            elif op.startswith("ds_"): c["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
            elif op.startswith("s_waitcnt"): c["wait"] += 1
            elif op.startswith("s_"): c["salu"] += 1
            else: c["other"] += 1
        if len(ins) >= floor:
            tail = [i for i in ins if i.startswith(("s_cbranch", "s_branch"))]
            print(f"{name:12s} {len(ins):5d}  " + " ".join(f"{k}={v}" for k, v in sorted(c.items())) + "   " + " ".join(tail[-2:]))


main()
