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


def lint_asm(asm):
    """-> (hazards: [(kernel, label, first copies)], wide: [(kernel, vgprs)] for kernels needing > 256 VGPRs)."""
    meta = open(asm).read()
    counts = dict(zip(re.findall(r"\.name:\s+(\S+)", meta), (int(v) for v in re.findall(r"\.vgpr_count:\s+(\d+)", meta))))
    hazards, wide = [], []
    for name, body in kernels(asm):
        for label, vec in lint(name, body):
            hazards.append((name, label, vec[:2]))
        if counts.get(name, 0) > 256:
            wide.append((name, counts[name]))
    return hazards, wide


def main():
    """Without arguments: lint the assembly the build keeps (picklebot_amd/lib/asm/*.s); with .hip files: compile them first."""
    total = 0
    asmdir = os.path.join(ROOT, "picklebot_amd", "lib", "asm")
    with tempfile.TemporaryDirectory() as td:
        if sys.argv[1:]:
            asms = [compile_to_asm(os.path.abspath(src), td) for src in sys.argv[1:]]
        else:
            asms = sorted(os.path.join(asmdir, f) for f in os.listdir(asmdir) if f.endswith(".s"))
        for asm in asms:
            hazards, wide = lint_asm(asm)
            for name, vg in wide:
                print(f"{os.path.basename(asm)}: {name[:100]} needs {vg} VGPRs (> 256: live ranges split into AGPRs)")
            for name, label, vec in hazards:
                total += 1
                print(f"{os.path.basename(asm)}: {name[:90]}: join block {label} has vector copies ahead of the exec restore: {vec}")
    print(f"{total} suspicious join blocks")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
