"""CPU: the gfx950 ISA of every kernel is linted for the code-generation hazard that hung a kernel in round 2
(tools/lint_isa.py): hipcc splits VGPR live ranges into AGPRs in kernels that need more than 256
registers and can place the save (v_accvgpr_write) at the top of a control-flow JOIN block, ahead of the exec
restore - lanes that skipped the branch then read a stale AGPR (wrong results or a loop that never ends).
The build keeps the device assembly (picklebot_amd/lib/asm/*.s), so this costs no compilation."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_vector_copies_ahead_of_exec_restore_in_join_blocks():
    from picklebot_amd import build
    import lint_isa
    build.build(verbose=False, require_asm=True)
    srcs = [os.path.basename(s)[:-4] for s in build.sources()]
    asms = [os.path.join(build.ASMDIR, b + ".s") for b in srcs]
    missing = [a for a in asms if not os.path.exists(a)]
    if missing and not os.access(build.ASMDIR, os.W_OK):
        pytest.skip("assembly not kept by this build (read-only tree)")
    assert not missing, f"the build did not keep the ISA of {missing}"
    hazards, wide = [], []
    for a in asms:
        h, w = lint_isa.lint_asm(a)
        hazards += [(os.path.basename(a),) + t for t in h]
        wide += [(os.path.basename(a),) + t for t in w]
    for f, name, vg in wide:
        print(f"{f}: {name[:100]} needs {vg} VGPRs")
    assert not hazards, "join blocks with v_accvgpr_write ahead of the exec restore:\n" + "\n".join(map(str, hazards))
