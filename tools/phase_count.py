#!/usr/bin/env python3
"""Per-phase instruction mix of a kernel from its gfx950 disassembly.

    tools/phase_count.py orbx_describe.hip _Z10k_describeILi0E

compiles the translation unit with -DORBX_PHASE_MARKERS (assembler comments at the phase boundaries, see the PHASE macro) and
counts, between consecutive markers of the named kernel, the instructions by unit: VALU (v_*), LDS (ds_*), VMEM (global_* /
buffer_* / flat_*), SALU / SMEM (s_*).  The count is static; it equals the executed count where the code is straight-line (fully
unrolled loops), and the branches of cold paths (edge keypoints, level-wide blur) are listed under the phase that contains them."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, kern = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "tools", "_build", "dis", os.path.basename(src).replace(".hip", "_phases.s"))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-DORBX_PHASE_MARKERS",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "orb_slam2v2-1_amd", "csrc"), "--cuda-device-only",
                           "-S", "-o", out, os.path.join(ROOT, "orb_slam2v2-1_amd", "csrc", src)] + sys.argv[3:], stderr=subprocess.DEVNULL)
    lines = open(out).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(kern) and re.match(r"\S+:(\s|$)", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    phase, counts, order = "entry", collections.OrderedDict(), []
    slow = collections.Counter()
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        m = re.match(r"; ORBX_PHASE (\S+)", t)
        if m:
            phase = m.group(1)
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        unit = "VALU" if op.startswith("v_") else "LDS" if op.startswith("ds_") else "VMEM" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) \
            else "SMEM" if op.startswith("s_load") or op.startswith("s_buffer") else "SALU" if op.startswith("s_") else "other"
        c = counts.setdefault(phase, collections.Counter())
        c[unit] += 1
        if unit == "VALU":
            c["op:" + re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", op)] += 1
    print("%-12s %6s %5s %5s %5s %5s" % ("phase", "VALU", "LDS", "VMEM", "SALU", "SMEM"))
    tot = collections.Counter()
    for ph, c in counts.items():
        print("%-12s %6d %5d %5d %5d %5d" % (ph, c["VALU"], c["LDS"], c["VMEM"], c["SALU"], c["SMEM"]))
        for u in ("VALU", "LDS", "VMEM", "SALU", "SMEM"):
            tot[u] += c[u]
    print("%-12s %6d %5d %5d %5d %5d" % ("total", tot["VALU"], tot["LDS"], tot["VMEM"], tot["SALU"], tot["SMEM"]))
    for ph, c in counts.items():
        ops = sorted(((v, k[3:]) for k, v in c.items() if k.startswith("op:")), reverse=True)[:8]
        print("%-12s %s" % (ph, ", ".join("%s x%d" % (k, v) for v, k in ops)))


if __name__ == "__main__":
    main()
