#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_r03.sh into profiles/ (tracked).

    python tools/summarize_pmc.py gpurun_out/prof_r03 r03

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, verbatim),
profiles/<tag>_pmc_summary.csv (per-kernel averages of the PMC passes for the dgmi kernels) and
profiles/traffic.json (fabric bytes per launch of the dominant kernel pair, corrected as
MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE is in KiB and under-counts wide coalesced reads
by exactly 2x on gfx950; WRITE_SIZE is exact; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; beside it the raw
L2 -> memory-side request counters split by target, and the sha256 of the kernel sources the profile was taken
with, so that bench.py can say when the committed figure no longer describes the kernels it launches).
"""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

src, tag = sys.argv[1], sys.argv[2]
write_traffic = "--no-traffic" not in sys.argv[3:]  # side profiles (the edge-dropped step) must not replace traffic.json
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
rd, rd_dram, rd32 = agg("pmc_rd_dram", "TCC_EA0_RDREQ_sum"), agg("pmc_rd_dram", "TCC_EA0_RDREQ_DRAM_sum"), agg("pmc_rd_dram", "TCC_EA0_RDREQ_32B_sum")
wr, wr_dram, wr64 = agg("pmc_wr_dram", "TCC_EA0_WRREQ_sum"), agg("pmc_wr_dram", "TCC_EA0_WRREQ_DRAM_sum"), agg("pmc_wr_dram", "TCC_EA0_WRREQ_64B_sum")
mean = lambda d, k: (sum(v[0] for v in d[k]) / len(d[k])) if d.get(k) else None
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
                 "RDREQ_avg": mean(rd, k), "RDREQ_DRAM_avg": mean(rd_dram, k), "RDREQ_32B_avg": mean(rd32, k),
                 "WRREQ_avg": mean(wr, k), "WRREQ_DRAM_avg": mean(wr_dram, k), "WRREQ_64B_avg": mean(wr64, k),
                 "VGPR": fetch[k][0][1], "SGPR": fetch[k][0][2]})
with open(os.path.join(out, tag + "_pmc_summary.csv"), "w", newline="") as fh:
    wr = csv.DictWriter(fh, fieldnames=list(rows[0]))
    wr.writeheader()
    wr.writerows(rows)

# dominant products (GCMC, unweighted, source-scaled): the 50k-source direction runs the 32-lane form, the
# 100k-source direction the 16-lane form (two column passes: grid.y) — average them by launch count
main = [r for r in rows if r["kernel"].startswith(("spmm_sliced_vec4_kernel<32, 0, false, false", "spmm_sliced_vec4_kernel<16, 0, false, false",
                                                   "spmm_sliced_vec4_kernel<32, false, false, false",   # (rounds 1-3: bool HAS_VALS)
                                                   "spmm_sliced_vec4_kernel<16, false, false, false"))]
red = [r for r in rows if r["kernel"].startswith("reduce_planes_kernel<true")]
pre = [r for r in rows if r["kernel"].startswith("scale_rows_kernel")]  # the row-scale pass ahead of the gather
dom = None
if main and red:
    n = sum(r["launches"] for r in main)
    gather = sum(r["hbm_bytes_corrected_avg"] * r["launches"] for r in main) / n
    dom = [{"kernel": (pre[0]["kernel"] + " + " if pre else "") + " | ".join(r["kernel"] for r in main) + " + " + red[0]["kernel"],
            "hbm_bytes_corrected_avg": int(gather + red[0]["hbm_bytes_corrected_avg"] + (pre[0]["hbm_bytes_corrected_avg"] if pre else 0)),
            "L2_hit_rate": {r["kernel"]: r["L2_hit_rate"] for r in main}}]
def src_hashes():
    out = {}
    for f in ("dgmi_sliced.hip", "dgmi_spmm.hip", "dgmi_segment.h", "dgmi_kernels.h", "dgmi_keep.h"):
        with open(os.path.join(root, "dream_gnn_amd", "csrc", f), "rb") as fh:
            out[f] = hashlib.sha256(fh.read()).hexdigest()[:16]
    return out


def git_head():
    try:
        return subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        return None


if dom and write_traffic:
    n = sum(r["launches"] for r in main)
    wavg = lambda key: (sum((r[key] or 0) * r["launches"] for r in main) / n) if all(r[key] is not None for r in main) else None
    req = {"gather_kernel": {k: wavg(k) for k in ("RDREQ_avg", "RDREQ_DRAM_avg", "RDREQ_32B_avg", "WRREQ_avg", "WRREQ_DRAM_avg", "WRREQ_64B_avg")},
           "reduce_kernel": {k: red[0][k] for k in ("RDREQ_avg", "RDREQ_DRAM_avg", "RDREQ_32B_avg", "WRREQ_avg", "WRREQ_DRAM_avg", "WRREQ_64B_avg")}}
    dram_rd = sum((req[p]["RDREQ_DRAM_avg"] or 0) for p in req)
    all_rd = sum((req[p]["RDREQ_avg"] or 0) for p in req)
    dram_wr = sum((req[p]["WRREQ_DRAM_avg"] or 0) for p in req)
    all_wr = sum((req[p]["WRREQ_avg"] or 0) for p in req)
    json.dump({"kernel": dom[0]["kernel"], "fabric_bytes_per_launch": dom[0]["hbm_bytes_corrected_avg"],
               "source": "profiles/%s_pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, "
                         "bench.py --steps 3 --warmup 1)" % tag,
               "correction": "(2*FETCH_SIZE + WRITE_SIZE)*1024; FETCH_SIZE counts L2->fabric requests, Infinity-Cache hits included",
               "requests_per_launch": req,
               "dram_targeted_share": {"reads": round(dram_rd / all_rd, 4) if all_rd else None,
                                       "writes": round(dram_wr / all_wr, 4) if all_wr else None,
                                       "meaning": "TCC_EA0_{RD,WR}REQ_DRAM / TCC_EA0_{RD,WR}REQ: the share of the L2's memory-side "
                                                  "requests addressed to local DRAM (vs. IO / xGMI). The Infinity Cache sits BEHIND this "
                                                  "interface, so the TCC counters cannot separate its hits from HBM accesses; gfx950 "
                                                  "exposes no MALL / UMC counter to rocprofv3 (rocprofv3 -L: none). The HBM-only figure is "
                                                  "therefore bounded, not measured: >= the compulsory bytes, <= fabric_bytes_per_launch"},
               "profiled_at_commit": git_head(), "kernel_source_sha256_16": src_hashes(),
               "L2_hit_rate": dom[0]["L2_hit_rate"]}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
for r in rows:
    print(r)
