#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_r01.sh into profiles/ (tracked).

    python tools/summarize_pmc.py gpurun_out/prof_r01 r01

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, verbatim),
profiles/<tag>_pmc_summary.csv (per-kernel averages of the PMC passes for the dgmi kernels) and
profiles/traffic.json (HBM/fabric bytes per launch of the dominant kernel, corrected as
MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE is in KiB and under-counts wide coalesced reads
by exactly 2x on gfx950; WRITE_SIZE is exact; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(out, tag + "_kernel_stats.csv"))


def agg(sub, counter):
    d = collections.defaultdict(list)
    for path in glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and "dgmi" in r["Kernel_Name"]:
                d[r["Kernel_Name"]].append((float(r["Counter_Value"]), int(r["VGPR_Count"]), int(r["SGPR_Count"])))
    return d


fetch, write = agg("pmc_fetch", "FETCH_SIZE"), agg("pmc_write", "WRITE_SIZE")
hit, miss = agg("pmc_l2", "TCC_HIT_sum"), agg("pmc_l2", "TCC_MISS_sum")
rows = []
for k in sorted(fetch):
    f = [v[0] for v in fetch[k]]
    w = [v[0] for v in write.get(k, [(0, 0, 0)])]
    h, m = sum(v[0] for v in hit.get(k, [])), sum(v[0] for v in miss.get(k, []))
    favg, wavg = sum(f) / len(f), sum(w) / len(w)
    name = k.replace("void ", "").replace("dgmi::(anonymous namespace)::", "")
    name = name[:name.find(">") + 1] if "<" in name.split("(")[0] else name.split("(")[0]
    rows.append({"kernel": name,
                 "launches": len(f), "FETCH_SIZE_KiB_avg": round(favg, 1), "WRITE_SIZE_KiB_avg": round(wavg, 1),
                 "hbm_bytes_corrected_avg": int((2 * favg + wavg) * 1024),
                 "L2_hit_rate": round(h / (h + m), 4) if h + m else "",
                 "VGPR": fetch[k][0][1], "SGPR": fetch[k][0][2]})
with open(os.path.join(out, tag + "_pmc_summary.csv"), "w", newline="") as fh:
    wr = csv.DictWriter(fh, fieldnames=list(rows[0]))
    wr.writeheader()
    wr.writerows(rows)

# dominant products (GCMC, unweighted, source-scaled): the 50k-source direction runs the 32-lane form, the
# 100k-source direction the 16-lane form (two column passes: grid.y) — average them by launch count
main = [r for r in rows if r["kernel"].startswith(("spmm_sliced_vec4_kernel<32, false, true, false>",
                                                   "spmm_sliced_vec4_kernel<16, false, true, false>"))]
red = [r for r in rows if r["kernel"].startswith("reduce_planes_kernel<true")]
dom = None
if main and red:
    n = sum(r["launches"] for r in main)
    gather = sum(r["hbm_bytes_corrected_avg"] * r["launches"] for r in main) / n
    dom = [{"kernel": " | ".join(r["kernel"] for r in main) + " + " + red[0]["kernel"],
            "hbm_bytes_corrected_avg": int(gather + red[0]["hbm_bytes_corrected_avg"]),
            "L2_hit_rate": {r["kernel"]: r["L2_hit_rate"] for r in main}}]
if dom:
    json.dump({"kernel": dom[0]["kernel"], "hbm_bytes_per_launch": dom[0]["hbm_bytes_corrected_avg"],
               "source": "profiles/%s_pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, "
                         "bench.py --steps 3 --warmup 1)" % tag,
               "correction": "(2*FETCH_SIZE + WRITE_SIZE)*1024; FETCH_SIZE counts L2->fabric requests, Infinity-Cache hits included",
               "L2_hit_rate": dom[0]["L2_hit_rate"]}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
for r in rows:
    print(r)
