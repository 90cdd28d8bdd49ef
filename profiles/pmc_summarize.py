"""Per-kernel HBM bytes per launch from the two rocprofv3 --pmc passes of profiles/collect.sh.

FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts half of the bytes of wide coalesced
reads (MI355X_MICROARCH.md, HBM / rocprofv3 section), hence the x2.  Usage:
    python profiles/pmc_summarize.py gpurun_out/prof_r1 > profiles/r1_pmc_traffic.json
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for path in glob.glob("%s/pmc_%s/**/*counter_collection.csv" % (root, ctr), recursive=True):
            for row in csv.DictReader(open(path)):
                if row["Counter_Name"] != ctr:
                    continue
                k = short(row["Kernel_Name"])
                acc[k][ctr] += float(row["Counter_Value"])
                launches[k][ctr] += 1
    out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 1 "
                      "--warmup 0 --no-cpu-baseline --no-throughput-leg (two separate passes, profiles/collect.sh)",
           "correction": "FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads), WRITE_SIZE as is, x1024 (KB -> bytes)",
           "kernels": {}}
    for k in acc:
        n = max(launches[k].values())
        f = acc[k]["FETCH_SIZE"] / max(1, launches[k]["FETCH_SIZE"])
        w = acc[k]["WRITE_SIZE"] / max(1, launches[k]["WRITE_SIZE"])
        out["kernels"][k] = {"FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "launches": n,
                             "hbm_bytes_per_launch_corrected": (2.0 * f + w) * 1024.0}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r1")
