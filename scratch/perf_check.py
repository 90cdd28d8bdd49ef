"""Quick regression check of the headline figures against the committed profiles (GPU box, repo root):
    python scratch/perf_check.py [tolerance]        # default: flags anything more than 7 % below the committed figure
Runs bench.py for C3 (with the five-candidate window leg), C2 and a short C5 and compares with profiles/r4_bench_*.json."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tol = float(sys.argv[1]) if len(sys.argv) > 1 else 0.07


def line(args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-spec-matrix", "--no-cpu-baseline", "--no-e2e"] + args,
                         capture_output=True, text=True, timeout=1800)
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def ref(name):
    return json.loads(open(os.path.join(ROOT, "profiles", "r4_bench_%s.json" % name)).read().strip().splitlines()[-1])


rows = []
d, r = line(["--steps", "20"]), ref("default")
rows.append(("C3 haplotypes/s", d["value"], r["value"]))
rows.append(("C3 five-candidate window", d["wide_window"]["value"], r["wide_window"]["value"]))
rows.append(("C3 256 windows (spins)", d["throughput_mode_256"]["value"], r["throughput_mode_256"]["value"]))
d, r = line(["--config", "C2", "--steps", "20", "--no-throughput-leg"]), ref("c2")
rows.append(("C2 haplotypes/s", d["value"], r["value"]))
d, r = line(["--config", "C5", "--steps", "1", "--warmup", "1", "--no-throughput-leg"]), ref("c5")
rows.append(("C5 haplotypes/s", d["value"], r["value"]))
bad = 0
for name, now, was in rows:
    t = max(tol, 0.15) if "256 windows" in name else tol       # (that leg reads 121k..136k on the same build)
    flag = "" if now >= (1 - t) * was else "   <-- REGRESSION"
    bad += bool(flag)
    print("%-28s %10.0f   committed %10.0f   %+5.1f %%%s" % (name, now, was, 100 * (now / was - 1), flag))
sys.exit(1 if bad else 0)
