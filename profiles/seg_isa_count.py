"""Instruction budget of k_seg (the dominant kernel of the headline config), counted in the ISA hipcc emits -- no GPU needed:

    python profiles/seg_isa_count.py > profiles/r4_seg_isa.json

k_seg<L> is bound by vector-ALU issue, not by HBM: per position it evaluates Next[t][state] for all R^L states (binary64 adds,
compares, selects) and then walks all R^L entry states through the segment by table lookups.  The two inner loops are found in
the assembly (the Next-table loop: the innermost loop with v_add_f64; the state walk: the innermost loop whose LDS reads are
byte/short Next entries), their instructions counted by class and priced with the issue costs of a wave64 on a SIMD-32:
binary64 VALU 4 cycles (measured on this part: DESIGN.md section 4.1), other VALU 2, LDS/SALU/branch overlapped with VALU
issue when four waves share a SIMD (counted, priced 0 in the floor).  bench.py turns that into microseconds per launch for the
window at hand and reports measured / floor.
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F64 = re.compile(r"^v_(add|mul|fma|max|min|cmp\w*|div\w*|rcp|trig\w*|ldexp|frexp\w*|cvt\w*)_f64|^v_cmp_\w+_f64")


def klass(op):
    if F64.match(op) or op.endswith("_f64") or "_f64_" in op:
        return "valu_f64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    return "salu"


def loops_of(lines):
    """innermost loops: [label line .. backward branch to that label] with no other label in between that is itself a target
    of a backward branch inside"""
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    out = []
    for i, l in enumerate(lines):
        m = re.match(r"\s*s_c?branch\w*\s+(\.LBB\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i))
    inner = [(a, b) for a, b in out if not any(a < c and d < b for c, d in out if (c, d) != (a, b))]
    return inner


def body(lines, a, b):
    return [l.strip().split()[0] for l in lines[a + 1:b + 1] if l.strip() and not l.strip().startswith((";", "."))]


def main():
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "-I" + os.path.join(ROOT, "include"), "-Wno-int-to-pointer-cast", "-S", "--cuda-device-only",
                               "-o", asm, os.path.join(ROOT, "gretel_amd", "csrc", "gretel_hip.hip")], stderr=subprocess.DEVNULL)
        text = open(asm).read()
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    out = {"git_head": head, "cost_cycles_per_wave_instruction": {"valu_f64": 4, "valu": 2, "lds": 0, "salu": 0, "vmem": 0, "wait": 0},
           "note": "costs: wave64 on a SIMD-32; LDS / scalar / branch instructions issue beside the VALU when four waves share a SIMD", "L": {}}
    def two_loops(lines):
        per_radix = {}
        nexts, walks = [], []
        for a, b in loops_of(lines):
            ops = body(lines, a, b)
            kinds = collections.Counter(klass(o) for o in ops)
            if any(o == "v_add_f64" for o in ops) and kinds["valu_f64"] >= 8 and not kinds["vmem"] and \
                    not any(o.startswith(("v_fma_f64", "v_mul_f64", "v_div", "v_rcp")) for o in ops):
                nexts.append((a, ops, kinds))       # (adds and compares only: the reweight loops of k_rwseg take logarithms and touch memory)
            elif any(o in ("ds_read_u8", "ds_read_u16", "ds_read_u8_d16", "ds_read_u16_d16", "ds_read_u8_d16_hi") for o in ops) and not any(o.startswith("global_store") for o in ops[:3]):
                walks.append((a, ops, kinds))
        # which loop belongs to which radix, by what it does rather than by where the compiler put it (round 4: the four-rank
        # body is emitted LAST): the Next-table loop of seg_body<4> has the fewest binary64 operations, the five-symbol one the
        # next more (the NaN-safe instantiation and the mixed-radix body are larger still); the state walk of seg_body<4> reads
        # one-byte entries, the five-symbol one two-byte entries
        def has(t, *prefixes):
            return any(o.startswith(prefixes) for o in t[1])
        mixed = lambda t: has(t, "v_readfirstlane", "v_readlane")            # the mixed-radix body hands its radices round in scalar registers
        n4 = [t for t in nexts if has(t, "ds_write_b8") and not mixed(t)]     # one-byte Next entries: four ranks
        # (two-byte entries: the NaN-safe five-symbol instantiation is the largest, the plain one the second largest; what is smaller belongs to other bodies)
        n5 = sorted((t for t in nexts if has(t, "ds_write_b16") and not mixed(t)), key=lambda t: -t[2]["valu_f64"])[1:2]
        # (the walk of a whole word of picks: the longest of the loops that read entries -- the shorter ones are its tails)
        w4 = sorted((t for t in walks if has(t, "ds_read_u8") and not mixed(t) and not has(t, "ds_write")), key=lambda t: -len(t[1]))
        w5 = sorted((t for t in walks if has(t, "ds_read_u16") and not mixed(t) and not has(t, "ds_write")), key=lambda t: -len(t[1]))
        picks = {"next_table_loop": [n4[0] if n4 else None, n5[0] if n5 else None],
                 "state_walk_loop": [w4[0] if w4 else None, w5[0] if w5 else None]}
        for name, lst in picks.items():
            for q, t in enumerate(lst):
                if t is None:
                    continue
                a, ops, kinds = t
                radix = "R4" if q == 0 else "R5"
                cyc = sum(out["cost_cycles_per_wave_instruction"][k] * v for k, v in kinds.items())
                per_radix.setdefault(radix, {})[name] = {"instructions": len(ops), "by_class": dict(kinds), "issue_cycles_per_iteration": cyc,
                                                         "by_opcode": dict(collections.Counter(ops).most_common(12))}
        return per_radix

    for lc in range(1, 7):
        m = re.search(r"^_Z5k_segILi%dELb0EEv10seg_params:.*?^\.Lfunc_end" % lc, text, re.S | re.M)
        if m:
            out["L"][str(lc)] = two_loops(m.group(0).split("\n"))
    # k_rwseg<float, L, false>: the same two loops as compiled inside the fused kernel (the reweight of the path before runs in front)
    out["rwseg"] = {"L": {}}
    for lc in range(1, 7):
        m = re.search(r"^_Z7k_rwsegIfLi%dELb0EEv10seg_params10rws_params:.*?^\.Lfunc_end" % lc, text, re.S | re.M)
        if m:
            out["rwseg"]["L"][str(lc)] = two_loops(m.group(0).split("\n"))
    # the candidate-pool walker (lag counts 6..24): instructions per step of its unrolled block loop (LC steps per trip)
    out["cwalk"] = {"cost_cycles_per_instruction": {"one_wave_per_simd": 5.0, "two_waves_per_simd": 2.5},
                    "note": "one wavefront walks 16 pool entries; a lone wave issues an instruction every ~5 cycles (scratch/ubench6/7.hip), "
                            "two workgroups per CU (L <= 13) interleave two walkers per SIMD", "L": {}}
    for lc in range(6, 25):
        m = re.search(r"^_Z7k_cwalkILi%dELi4EEv9cw_params:.*?^\.Lfunc_end" % lc, text, re.S | re.M)
        if not m:
            continue
        lines = m.group(0).split("\n")
        best = None
        for a, b in loops_of(lines):
            ops = body(lines, a, b)
            adds = sum(o == "v_add_f64" for o in ops)
            if adds >= (lc - 1) * 2 and (best is None or len(ops) > len(best)):
                best = ops
        if best:
            adds = sum(o == "v_add_f64" for o in best)
            steps = max(1, round(adds / (lc - 1)))
            out["cwalk"]["L"][str(lc)] = {"instructions_per_trip": len(best), "steps_per_trip": steps, "instructions_per_step": len(best) / steps,
                                          "by_class": dict(collections.Counter(klass(o) for o in best))}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
