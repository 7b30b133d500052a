"""Lint the gfx950 ISA of every kernel for a code-generation hazard found on MI355X (round 2):

hipcc (ROCm 7.2) splits live ranges of VGPRs into AGPRs when a kernel needs more than 256 registers, and it can
place the copy (v_accvgpr_write) at the top of a control-flow JOIN block *before* the `s_or_b64 exec, exec, ...` that
re-enables the lanes which skipped the branch.  Those lanes never save their value, and the later v_accvgpr_read
hands them whatever the previous kernel left in the AGPR: results (or loop trip counts - a hang) that depend on what
ran before.  Observed in dw_rows_wgrad_kernel<float,3,1,14>: waves spinning in the frame loop after a stale `z`.

usage: python tools/lint_isa.py [file.hip ...]   (default: every picklebot_amd/csrc/*.hip)
Reports, per kernel: VGPR count, AGPR copies, and the join blocks with vector instructions ahead of the exec restore.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "picklebot_amd", "csrc")


def compile_to_asm(src, outdir):
    base = os.path.basename(src)[:-4]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    "-c", src, "-save-temps", "-o", base + ".o"], cwd=outdir, check=True, capture_output=True)
    return os.path.join(outdir, base + "-hip-amdgcn-amd-amdhsa-gfx950.s")


def kernels(asm):
    cur, body = None, []
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            body.append(line.rstrip("\n"))
            if line.startswith(".Lfunc_end"):
                yield cur, body
                cur = None


def lint(name, body):
    text = "\n".join(body)
    m = re.search(r"\.vgpr_count:\s+(\d+)", text)
    hazards = []
    i = 0
    while i < len(body):
        if re.match(r"^\.LBB\d+_\d+:", body[i]):
            label = body[i].split(":")[0]
            vec = []
            j = i + 1
            while j < len(body) and not re.match(r"^\.LBB\d+_\d+:", body[j]):
                ins = body[j].strip()
                if re.match(r"s_or_b64 exec, exec", ins) or re.match(r"s_mov_b64 exec", ins):
                    if vec:
                        hazards.append((label, vec[:4]))
                    break
                if re.match(r"(s_cbranch|s_branch|s_endpgm|s_andn2_b64 exec|s_and_saveexec|s_and_b64 exec)", ins):
                    break
                if re.match(r"v_accvgpr_write", ins):
                    vec.append(ins)
                j += 1
        i += 1
    return hazards


def main():
    srcs = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    total = 0
    with tempfile.TemporaryDirectory() as td:
        for src in srcs:
            asm = compile_to_asm(os.path.abspath(src), td)
            meta = open(asm).read()
            counts = dict(zip(re.findall(r"\.name:\s+(\S+)", meta), re.findall(r"\.vgpr_count:\s+(\d+)", meta)))
            for name, body in kernels(asm):
                hz = lint(name, body)
                nacc = sum(1 for l in body if "v_accvgpr_write" in l)
                if hz:
                    total += len(hz)
                    print(f"{os.path.basename(src)}: {name[:90]} vgprs={counts.get(name, '?')} agpr_copies={nacc}: {len(hz)} join block(s) with "
                          f"vector copies ahead of the exec restore, e.g. {hz[0][0]} {hz[0][1][:2]}")
    print(f"{total} suspicious join blocks")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
