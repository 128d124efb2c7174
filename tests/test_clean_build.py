"""The tree compiles from clean (CPU tier; hipcc cross-compiles gfx950 without a GPU).

build() reuses dojo.jl_amd/csrc/libdojo_hip.so when its content stamp matches the sources, so a run that finds the shipped library current
never exercises the compiler.  This test does, for the objects the BASELINE metric runs through: the headline kernel variant (fp32 ABI, one
contact per body, single-wavefront quad mapping: Ant / Quadruped) and the host / C-ABI object, from this tree's sources into an empty
temporary directory with the flags of __graft_entry__.build_hip -- then checks that the fresh kernel object has the register / LDS
footprint tests/test_kernel_resources.py guards (and the one of the object the shipped library was linked from, when that is there), and
that the fresh host object defines every entry point include/dojo_hip.h declares.  (The full 25-object build takes ~2 minutes on 16 cores:
`python -c "import __graft_entry__ as g; g.build_hip(force=True)"`; profiles/README.md records a run of it on the GPU box.)"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dojo.jl_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include")]


def _resources(obj):
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "kernel_resources.sh"), obj], capture_output=True, text=True, timeout=300).stdout
    res = {}
    for ln in out.splitlines():
        m = re.search(r"\.name:\s+(dojo_\w+?_kernel)I.*?\.private_segment_fixed_size:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", ln)
        lds = re.search(r"\.group_segment_fixed_size:\s+(\d+)", ln)
        if m:
            res[m.group(1)] = (int(lds.group(1)), int(m.group(2)), int(m.group(3)))
    return res


@pytest.mark.skipif(HIPCC is None or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"), reason="no hipcc / llvm tools")
def test_headline_objects_compile_from_clean(tmp_path):
    kobj, hobj = str(tmp_path / "k_float_1_1.o"), str(tmp_path / "host.o")
    flags = ["-DDJ_TIO=float", "-DDJ_MAXC=1", "-DDJ_QUAD=1", "-DDJ_TSD=0", "-DDJ_LINEAR=0", "-DDJ_SS=0"]         # __graft_entry__.build_hip's variant (float, 1, 1, 0)
    jobs = [subprocess.Popen([HIPCC] + COMMON + flags + ["-c", os.path.join(CSRC, "dojo_kernels.hip"), "-o", kobj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True),
            subprocess.Popen([HIPCC] + COMMON + ["-c", os.path.join(CSRC, "dojo_hip.hip"), "-o", hobj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)]
    for j in jobs:
        out, _ = j.communicate(timeout=900)
        assert j.returncode == 0, out[-2000:]
    res = _resources(kobj)
    assert {"dojo_step_kernel", "dojo_grad_kernel", "dojo_cgrad_kernel"} <= set(res), res
    lds, scratch, spills = res["dojo_step_kernel"]
    assert lds <= 40960 and scratch <= 256 and spills <= 40, res["dojo_step_kernel"]        # (tests/test_kernel_resources.py: round 6, both layouts of the level passes)
    lds, scratch, spills = res["dojo_grad_kernel"]
    assert lds <= 40960 and scratch <= 1024, res["dojo_grad_kernel"]
    shipped = os.path.join(CSRC, "build", "k_float_1_1.o")
    if os.path.exists(shipped) and os.path.getmtime(shipped) >= max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".hpp", ".hip"))):
        assert _resources(shipped) == res                       # the object the library was linked from is this compilation
    # the C ABI: every entry point of the header is defined by the fresh host object
    hdr = open(os.path.join(ROOT, "include", "dojo_hip.h")).read()
    declared = set(re.findall(r"\b(dojo_[a-z0-9_]+)\s*\(", hdr))
    syms = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--symbols", "--wide", hobj], capture_output=True, text=True).stdout
    defined = {m.group(1) for m in re.finditer(r"FUNC\s+GLOBAL\s+\w+\s+(?!UND)\d+\s+(dojo_[a-z0-9_]+)\b", syms)}
    missing = {d_ for d_ in declared if d_ not in defined and not d_.startswith("dojo_launch_")}
    assert not missing, sorted(missing)


# one object of every other FAMILY of kernel builds (__graft_entry__.build_hip's variant table): the feature code behind -DDJ_TSD / the general
# lane-mapping builds / LinearContact / body-body contacts compiles from this tree as well, and exports its launcher
FAMILIES = [("tsd", ["-DDJ_TIO=float", "-DDJ_MAXC=1", "-DDJ_QUAD=1", "-DDJ_TSD=1", "-DDJ_LINEAR=0", "-DDJ_SS=0", "-DDJ_MLIM=0", "-DDJ_CUT=0"], "dojo_launch_tsd_float_1_1"),
            ("gen", ["-DDJ_TIO=float", "-DDJ_MAXC=4", "-DDJ_QUAD=0", "-DDJ_TSD=1", "-DDJ_LINEAR=0", "-DDJ_SS=1", "-DDJ_MLIM=1", "-DDJ_CUT=1"], "dojo_launch_gen_float_4_0"),
            ("two_wavefronts", ["-DDJ_TIO=float", "-DDJ_MAXC=4", "-DDJ_QUAD=2", "-DDJ_TSD=0", "-DDJ_LINEAR=0", "-DDJ_SS=0", "-DDJ_MLIM=0", "-DDJ_CUT=0"], "dojo_launch_float_4_2")]


@pytest.mark.skipif(HIPCC is None or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"), reason="no hipcc / llvm tools")
def test_one_object_of_every_build_family_compiles_from_clean(tmp_path):
    jobs = []
    for name, flags, _ in FAMILIES:
        obj = str(tmp_path / ("k_%s.o" % name))
        jobs.append((obj, subprocess.Popen([HIPCC] + COMMON + flags + ["-c", os.path.join(CSRC, "dojo_kernels.hip"), "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for (obj, j), (name, _, launcher) in zip(jobs, FAMILIES):
        out, _ = j.communicate(timeout=1500)
        assert j.returncode == 0, name + ": " + out[-2000:]
        syms = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--symbols", "--wide", obj], capture_output=True, text=True).stdout
        assert re.search(r"FUNC\s+GLOBAL\s+\w+\s+(?!UND)\d+\s+" + launcher + r"\b", syms), (name, launcher)
        res = _resources(obj)
        # (LDS of a workgroup; the two-wavefront layout carries the row passes' staging areas of both wavefronts: two workgroups per CU = 81 920 B each)
        assert "dojo_step_kernel" in res and res["dojo_step_kernel"][0] <= (81920 if name == "two_wavefronts" else 65536), (name, res)
