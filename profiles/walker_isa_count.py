"""Instruction budget of the path-extension walker's inner loop, counted in the ISA hipcc emits.

A lone wavefront issues one instruction every 5 cycles on MI355X (scratch/ubench6.hip, ubench7.hip), so
instructions per step x 5 is the floor of k_walk_spec's cycles per step.  Usage (no GPU needed):
    python profiles/walker_isa_count.py > profiles/r1_walker_isa.json
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def loop_of(lines):
    """the unrolled group loop of the depth-2 walker: the one whose resolve masks the symbol with `& 3`"""
    ff = [i for i, l in enumerate(lines) if "s_ff1_i32_b64" in l]
    for i in ff:
        if any("s_and_b32" in lines[j] and lines[j].rstrip().endswith(", 3") for j in range(i, i + 4)):
            a = i
            while not lines[a].startswith(".LBB"):
                a -= 1
            b = i
            while "s_cbranch" not in lines[b]:
                b += 1
            return [l.strip() for l in lines[a + 1:b + 1] if l.strip() and not l.strip().startswith(";")]
    return None


def main():
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "-I" + os.path.join(ROOT, "include"), "-Wno-int-to-pointer-cast", "-S", "--cuda-device-only",
                               "-o", asm, os.path.join(ROOT, "gretel_amd", "csrc", "gretel_hip.hip")],
                              stderr=subprocess.DEVNULL)
        text = open(asm).read()
    out = {"cycles_per_instruction_lone_wave": 5.0, "source": "scratch/ubench6.hip, scratch/ubench7.hip (MI355X)", "L": {}}
    for lc in range(2, 17):
        m = re.search(r"^_Z11k_walk_specILi%dE.*?^\.Lfunc_end" % lc, text, re.S | re.M)
        if not m:
            continue
        body = loop_of(m.group(0).split("\n"))
        if not body:
            continue
        steps = sum("s_ff1_i32_b64" in l for l in body)
        kinds = collections.Counter(l.split()[0] for l in body)
        out["L"][str(lc)] = {"instructions_per_loop": len(body), "steps_per_loop": steps,
                             "instructions_per_step": len(body) / steps,
                             "issue_floor_cycles_per_step": 5.0 * len(body) / steps,
                             "by_opcode": dict(kinds.most_common())}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
