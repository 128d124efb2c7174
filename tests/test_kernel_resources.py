"""Guard on the register footprint of the headline kernels (Ant / Quadruped: fp32 ABI, MAXC = 1, quad mapping).  With all 512
registers of a SIMD in use the step kernel's speed follows its spills (256 -> 416 B/lane of scratch cost
11 %), and small edits anywhere in the lane program move the allocator -- this test makes such a change visible on the CPU tier.
Reads the object __graft_entry__.build() leaves in dojo.jl_amd/csrc/build (skipped when it is not there)."""
import os
import re
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "dojo.jl_amd", "csrc", "build", "k_float_1_1.o")
TOOL = os.path.join(ROOT, "tools", "kernel_resources.sh")


@pytest.mark.skipif(not (os.path.exists(OBJ) and os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf")), reason="no built object / no llvm tools")
def test_headline_kernels_keep_their_register_footprint():
    out = subprocess.run(["bash", TOOL, OBJ], capture_output=True, text=True, timeout=300).stdout
    res = {}
    for ln in out.splitlines():
        m = re.search(r"\.name:\s+(dojo_\w+?_kernel)I.*?\.private_segment_fixed_size:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", ln)
        lds = re.search(r"\.group_segment_fixed_size:\s+(\d+)", ln)
        if m:
            res[m.group(1)] = (int(m.group(2)), int(m.group(3)), int(lds.group(1)) if lds else None)
    assert "dojo_step_kernel" in res and "dojo_grad_kernel" in res, out
    scratch, spills, lds = res["dojo_step_kernel"]
    # (round 6: the kernel carries both layouts of the factorization's level passes -- factorize_quad and factorize_rows, chosen per mechanism at run
    #  time -- 352 B / 103 spilled; the quad-only build (-DDJ_ROWS=0) has 256 / 43, the rows-only one (-DDJ_ROWS=1) 288 / 76 and runs no faster than this
    #  one on the same box, profiles/README.md r06)
    # (later in round 6: the device atan with its coefficients in SGPRs -- dojo_math.hpp tatan -- took the spilled coefficient moves away: 176 B / 6 spilled)
    assert scratch <= 256 and spills <= 40, ("step kernel: scratch %d B/lane, %d spilled VGPRs (was 176 / 6)" % (scratch, spills))
    assert lds <= 40960, lds                                  # four workgroups (one wave per SIMD) per CU
    scratch, spills, lds = res["dojo_grad_kernel"]
    assert scratch <= 1024 and lds <= 40960, (scratch, lds)   # (its spills sit in the once-per-step prologue -- linearization, LU-form factorization, data blocks)
    # ... and not in the pipelined sweeps: with one wave per SIMD nothing hides a scratch round trip (30 of them per pipeline step ran
    # the sweeps at half speed).  tools/isa_loops.py lists the loops of the kernel with their instruction mix.
    import ast
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_loops.py"), OBJ, "dojo_grad_kernel", "900"], capture_output=True, text=True, timeout=300).stdout
    loops = []
    for ln in out.splitlines():
        if ln.startswith("loop"):
            a, b = [int(x) for x in re.findall(r"\d+", ln[:ln.index("{")])[:2]]
            loops.append((a, b, ast.literal_eval(ln[ln.index("{"):])))
    cand = [l for l in loops if 1000 <= l[2]["n"] <= 2000 and l[2]["dpp"] >= 100 and l[2]["glob"] >= 10]   # the up-sweep's branch step and the down-sweep's
    # innermost only: the out-of-line blocks of a loop's conditional loads jump back into it, which the loop finder reports as a second, outer
    # "loop" that also spans the (once-per-kernel) preheader
    sweeps = [l[2] for l in cand if not any(o is not l and l[0] <= o[0] and o[1] <= l[1] for o in cand)]
    assert len(sweeps) >= 2 and all(m["scratch"] == 0 for m in sweeps), sweeps
