#!/usr/bin/env python3
"""include/gh_logtab.inc: the 128-entry (1/c, log c) table of the double-precision log() of the Arm Optimized Routines
(math/log_data.c, N = 128; Copyright (c) 2018 Arm Limited, MIT OR Apache-2.0 WITH LLVM-exception), which glibc 2.28 and later
carry unchanged as `__log_data.tab` (sysdeps/ieee754/dbl-64/e_log_data.c).  include/gh_detlog.h restates that log10 around it
so that the kernels' log10 is, bit for bit, the function Python's math.log10 calls on an x86-64 host with FMA -- which is what
the reference's arg-max and likelihood sums are made of (gretel/gretel.py:2,155-174,185-186).

    --verify      (default; tests/test_detlog.py runs it) check the committed table's CONSTRUCTION with exact arithmetic, no libm
                  and no third-party source needed: every c = 1/invc lies within 2^29 ulp of the centre of its subinterval
                  (the published search range) and every log c equals round(2^43 ln(1/invc)) / 2^43 -- the second column
                  follows from the first;
    --from-libm   cross-check only: read `__log_data.tab` out of the running system's libm.so.6 and compare it with the
                  committed file (the first build of the table was made this way, from Ubuntu GLIBC 2.35-0ubuntu3.11);
    --consts      with --from-libm: print the polynomial coefficients of that libm.

Nothing here writes the .inc any more: the committed file is the table, carried with its origin.

(--from-libm locates the table by its neighbours, not by an address: it follows ln2hi, ln2lo, the five coefficients of the
main polynomial and the eleven of the near-one polynomial.)  tests/test_detlog.py holds the restatement to the running
libm on tens of millions of arguments.
"""
import ctypes.util
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "include", "gh_logtab.inc")


def find_libm():
    for p in ("/lib/x86_64-linux-gnu/libm.so.6", "/lib64/libm.so.6", "/usr/lib/x86_64-linux-gnu/libm.so.6", "/usr/lib64/libm.so.6"):
        if os.path.exists(p):
            return p
    raise SystemExit("libm.so.6 not found (looked for %s)" % ctypes.util.find_library("m"))


def extract(path):
    blob = open(path, "rb").read()
    # __log_data starts: ln2hi = 0x1.62e42fefa3800p-1, ln2lo = 0x1.ef35793c76730p-45, poly[0] = -0x1.0000000000001p-1
    head = struct.pack("<3d", float.fromhex("0x1.62e42fefa3800p-1"), float.fromhex("0x1.ef35793c76730p-45"),
                       float.fromhex("-0x1.0000000000001p-1"))
    at = blob.find(head)
    if at < 0 or blob.find(head, at + 1) >= 0:
        raise SystemExit("%s: __log_data not found (or not unique): not a glibc >= 2.28 libm?" % path)
    vals = struct.unpack_from("<%dd" % (2 + 5 + 11 + 256), blob, at)
    return dict(ln2hi=vals[0], ln2lo=vals[1], A=vals[2:7], B=vals[7:18], tab=vals[18:])


def committed():
    import re
    txt = open(OUT).read()
    txt = txt[txt.index("*/") + 2:]
    vals = [float.fromhex(x) for x in re.findall(r"-?0x[0-9a-f.]+p[+-]\d+", txt)]
    if len(vals) != 256:
        raise SystemExit("%s: %d values, expected 256" % (OUT, len(vals)))
    return vals


def verify():
    """The construction of the table, with exact (80-digit) arithmetic.  Returns a list of complaints (empty = fine)."""
    from decimal import Decimal, getcontext
    getcontext().prec = 80
    vals = committed()
    bad = []
    for i in range(128):
        invc, logc = vals[2 * i], vals[2 * i + 1]
        c = Decimal(1) / Decimal(invc)
        want = float((c.ln() * (Decimal(2) ** 43)).to_integral_value(rounding="ROUND_HALF_EVEN")) / 2.0 ** 43
        if want != logc:
            bad.append("entry %d: log c = %s, round(2^43 ln(1/invc))/2^43 = %s" % (i, logc.hex(), want.hex()))
        lo = struct.unpack("<d", struct.pack("<Q", 0x3FE6000000000000 + (i << 45)))[0]
        hi = struct.unpack("<d", struct.pack("<Q", 0x3FE6000000000000 + ((i + 1) << 45)))[0]
        centre = Decimal(lo + hi) / 2
        if abs(c / centre - 1) > Decimal(2) ** -22:         # 2^29 ulp of a number in [1, 2) is 2^-23
            bad.append("entry %d: c = %s is not at the centre %s of its subinterval" % (i, c, centre))
    return bad


if __name__ == "__main__":
    if "--from-libm" in sys.argv:
        d = extract(find_libm())
        ok = [float.fromhex(v.hex()) for v in d["tab"]] == committed()
        print("libm table %s the committed include/gh_logtab.inc" % ("==" if ok else "DIFFERS FROM"))
        if "--consts" in sys.argv:
            print("ln2hi", d["ln2hi"].hex(), "ln2lo", d["ln2lo"].hex())
            print("A", [a.hex() for a in d["A"]])
            print("B", [b.hex() for b in d["B"]])
        sys.exit(0 if ok else 1)
    complaints = verify()
    print("\n".join(complaints) if complaints else "include/gh_logtab.inc: 128 entries, construction verified")
    sys.exit(1 if complaints else 0)
    print("wrote", OUT)
