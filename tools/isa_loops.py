"""instruction mix of the loops of one kernel in an object of dojo_kernels.hip: tools/isa_loops.py build/k_float_1_1.o dojo_grad_kernel [min_instr]"""
import re, sys, subprocess, tempfile, os, bisect
o, kern = sys.argv[1], sys.argv[2]; minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 400
t = tempfile.mkdtemp()
subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", o, t + "/fb.bin"])
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + t + "/fb.bin", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + t + "/k.co"])
lines = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", t + "/k.co"]).decode().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r'^[0-9a-f]+ <.*' + kern + 'I', l)][0]
end = [i for i, l in enumerate(lines) if i > start and re.match(r'^[0-9a-f]+ <', l)][0]
ins = []
for l in lines[start + 1:end]:
    m = re.search(r'//\s*([0-9A-Fa-f]+):', l)
    if m: ins.append((int(m.group(1), 16), l.split('//')[0].strip()))
addrs = [a for a, _ in ins]
loops = set()
for i, (a, tx) in enumerate(ins):
    m = re.match(r's_cbranch_\w+\s+(\d+)|s_branch\s+(\d+)', tx)
    if m:
        off = int(m.group(1) or m.group(2)); off = off - 65536 if off >= 32768 else off
        tgt = a + 4 + off * 4
        if tgt < a: loops.add((bisect.bisect_left(addrs, tgt), i))
def mix(seg):
    c = dict(n=len(seg), f64=0, dpp=0, ds=0, glob=0, scratch=0, acc=0, cnd=0, salu=0, other_valu=0, wait=0)
    for _, tx in seg:
        op = tx.split()[0] if tx else ""
        if "scratch_" in op: c["scratch"] += 1
        elif re.match(r"v_(fma|fmac|mul|add|rcp|rsq|sqrt|max|min|div\w*|trig\w*|ldexp|frexp\w*)_f64", op): c["f64"] += 1
        elif "dpp" in tx: c["dpp"] += 1
        elif op.startswith("ds_"): c["ds"] += 1
        elif op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"): c["glob"] += 1
        elif "accvgpr" in op: c["acc"] += 1
        elif op.startswith("v_cndmask"): c["cnd"] += 1
        elif op.startswith("s_waitcnt") or op.startswith("s_nop"): c["wait"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith("v_"): c["other_valu"] += 1
    return c
print(kern, "instructions:", len(ins), mix(ins))
# keep outermost representatives of distinct regions: sort by start, drop loops nested with nearly equal extent
L = sorted(loops, key=lambda x: (x[0], -x[1]))
shown = []
for j, i in L:
    if i - j + 1 < minlen: continue
    if any(abs(j - a) < 120 and abs(i - b) < 120 for a, b in shown): continue
    shown.append((j, i))
    print("loop [%5d..%5d]" % (j, i), mix(ins[j:i + 1]))
