"""Build the C-ABI HIP library in-tree: picklebot_amd/lib/libpicklebot_amd.so (gfx950 only).

    python -m picklebot_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects are cached per source (rebuilt when the source,
a header or the flags change), so an incremental build takes seconds.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
ASMDIR = os.path.join(LIBDIR, "asm")           # gfx950 ISA of every source (tools/lint_isa.py, tests/test_isa_lint_cpu.py)
LIB = os.path.join(LIBDIR, "libpicklebot_amd.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-I", INCLUDE]
FLAGS += os.environ.get("PB_BUILD_DEFINES", "").split()      # e.g. -DPB_STEM_PROBE / -DPB_STREAM_PROBE: phase timers (experiments)


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return "hipcc"


def _digest(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def build(force=False, verbose=True, require_asm=False):
    """require_asm: also rebuild sources whose kept assembly (lib/asm/*.s, not shipped to the GPU box) is missing - the ISA
    lint test asks for it; an ordinary build never recompiles just for that."""
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(ASMDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    hipcc = _hipcc()
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        stamp = obj + ".sha"
        dig = _digest([src] + headers, " ".join(FLAGS))
        objs.append(obj)
        asm = os.path.join(ASMDIR, os.path.basename(src)[:-4] + ".s")
        if (force or not os.path.exists(obj) or not os.path.exists(stamp) or open(stamp).read() != dig
                or (require_asm and not os.path.exists(asm) and os.access(ASMDIR, os.W_OK))):
            jobs.append((src, obj, stamp, dig))

    def compile_one(job):
        src, obj, stamp, dig = job
        # -save-temps=obj: the device assembly is kept (lib/asm/<name>.s) for the ISA lint; the other temporaries are removed
        cmd = [hipcc] + FLAGS + ["-save-temps=obj", "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        base = os.path.basename(obj)[:-2]
        for f in os.listdir(OBJDIR):
            if f.startswith((base + "-hip-", base + "-host-", base + ".hip-")):
                full = os.path.join(OBJDIR, f)
                if f.endswith("gfx950.s"):
                    os.replace(full, os.path.join(ASMDIR, base + ".s"))
                else:
                    os.remove(full)
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        with open(stamp, "w") as f:
            f.write(dig)
        return src

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for src in ex.map(compile_one, jobs):
                if verbose:
                    print("compiled", os.path.relpath(src, HERE))
    if jobs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("linked", os.path.relpath(LIB, HERE))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
