"""Average duration per position of a repeating launch pattern, from a rocprofv3 kernel-trace CSV:
trace_by_grid.py <csv> <period> [<repeats to use from the end>]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
period = int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rows = rows[-period * reps:]
for pos in range(period):
    sel = rows[pos::period]
    name = sel[0]["Kernel_Name"]
    short = name.split("::")[-1][:60]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel]
    gaps = [int(r["Start_Timestamp"]) for r in sel]
    print("%d %-62s grid %8s x%s  avg %8.1f us  (min %.1f max %.1f)" % (pos, short, sel[0].get("Grid_Size_X", "?"), sel[0].get("Grid_Size_Y", "?"),
                                                                     sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3))
