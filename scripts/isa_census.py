#!/usr/bin/env python3
"""Instruction census of a kernel of cactus_amd/csrc/mb_kernels.hip from the compiler's own assembly (no GPU needed):
   python scripts/isa_census.py [kernel substring, default k_ydrop2E] > profiles/rNN_<kernel>_isa_census.txt
Compiles the device side with the Makefile's flags (hipcc -S --cuda-device-only), cuts the kernel out, and prints per basic block the
number of vector / scalar / other instructions with its branches, and the opcode histogram of the largest blocks."""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1] if len(sys.argv) > 1 else "k_ydrop2E"
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                    "--cuda-device-only", "-S", "-o", out, os.path.join(ROOT, "cactus_amd", "csrc", "mb_kernels.hip")], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN2mb\d+%s.*:\s*;" % re.escape(want), l) or re.match(r"^_ZN2mb\w*%s\w*:" % re.escape(want), l))
name = lines[start].split(":")[0]
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
blocks, cur = [], ("entry", [])
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur); cur = (m.group(1), [])
    else:
        t = l.split(";")[0].strip()
        if t and not t.startswith("."):
            cur[1].append(t)
blocks.append(cur)
meta = {k: v for k, v in re.findall(r"\.set %s\.(\w+), (\d+)" % re.escape(name), "\n".join(lines))}
print("kernel %s\n  VGPRs %s, SGPRs %s, scratch %s bytes" % (name, meta.get("num_vgpr"), meta.get("numbered_sgpr"), meta.get("private_seg_size")))
tot = collections.Counter()
print("\nbasic blocks (instructions: vector / scalar / memory+other; branches):")
for nm, ins in blocks:
    v = sum(1 for i in ins if i.startswith("v_")); s = sum(1 for i in ins if i.startswith("s_")); o = len(ins) - v - s
    tot.update(v=v, s=s, o=o)
    br = " | ".join(i for i in ins if i.startswith("s_cbranch") or i.startswith("s_branch"))
    print("  %-12s %4d = %4d v + %3d s + %2d o   %s" % (nm, len(ins), v, s, o, br))
print("  total        %4d = %4d v + %3d s + %2d o" % (sum(tot.values()), tot["v"], tot["s"], tot["o"]))
for nm, ins in sorted(blocks, key=lambda b: -len(b[1]))[:4]:
    c = collections.Counter(i.split()[0] for i in ins)
    print("\nopcodes of %s (%d instructions): " % (nm, len(ins)) + ", ".join("%d %s" % (n, op) for op, n in c.most_common()))
