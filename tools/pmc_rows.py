"""Print per-dispatch counter values of dgmi SpMM kernels from a rocprofv3 --pmc csv directory."""
import collections, csv, glob, sys
d = collections.OrderedDict()
for path in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "spmm" not in n and "reduce_planes" not in n:
            continue
        key = (int(r["Dispatch_Id"]), n.split("(")[0].replace("void dgmi::(anonymous namespace)::", "")[:60])
        d.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (disp, name), c in sorted(d.items()):
    extra = ""
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        extra = " L2hit=%.3f" % (c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]))
    print(disp, name, " ".join("%s=%.0f" % kv for kv in c.items()) + extra)
