"""Per-kernel HBM bytes per launch from the two rocprofv3 --pmc passes of profiles/collect.sh.

FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts half of the bytes of wide coalesced
reads (MI355X_MICROARCH.md, HBM / rocprofv3 section), hence the x2.  Usage:
    python profiles/pmc_summarize.py gpurun_out/prof_r4 [pmc] > profiles/r4_pmc_traffic.json
(second argument: the prefix of the two pass directories, pmc -> pmc_FETCH_SIZE / pmc_WRITE_SIZE; pmc_c5, pmc_b256 likewise).
The summary records the build it belongs to: kernel_source_sha (what bench.py printed in that very run: a hash over the
kernel sources, the GPU box has no .git) and the git HEAD of the tree the summary is made in -- bench.py quotes a
traffic figure only for the build it was measured on.
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


def main(root, prefix="pmc"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for path in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (root, prefix, ctr), recursive=True):
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
    import os
    import subprocess
    try:
        line = [l for l in open("%s/%s_FETCH_SIZE.json" % (root, prefix)) if l.startswith("{")][-1]
        doc = json.loads(line)
        out["kernel_source_sha"] = doc["roofline"]["kernel_source_sha"]
        # batched runs (bench.py --batch B --paths P): what one launch of the window pipeline covered -- bench.py scales its
        # traffic figure to the launch it times
        cfg = doc.get("config", {})
        if "windows_per_gpu" in cfg:
            out["windows"], out["paths"] = cfg["windows_per_gpu"], cfg["paths"]
    except Exception:
        out["kernel_source_sha"] = None
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out["git_head"] = subprocess.run(["git", "-C", here, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    out["git_dirty"] = bool(subprocess.run(["git", "-C", here, "status", "--porcelain", "--", "gretel_amd/csrc", "include"], capture_output=True, text=True).stdout.strip())
    for k in acc:
        n = max(launches[k].values())
        f = acc[k]["FETCH_SIZE"] / max(1, launches[k]["FETCH_SIZE"])
        w = acc[k]["WRITE_SIZE"] / max(1, launches[k]["WRITE_SIZE"])
        out["kernels"][k] = {"FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "launches": n,
                             "hbm_bytes_per_launch_corrected": (2.0 * f + w) * 1024.0}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r4", sys.argv[2] if len(sys.argv) > 2 else "pmc")
