"""Per-step kernel table from a rocprofv3 --kernel-trace --stats directory.  usage: python scripts/kernel_stats.py DIR STEPS [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
steps = int(sys.argv[2])
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 26]:
    print("%-88s %5d/step %8.2f ms/step %6.2f%%" % (r["Name"][:88], int(r["Calls"]) // steps, float(r["TotalDurationNs"]) / 1e6 / steps, float(r["Percentage"])))
print("total ms/step %.2f" % (tot / 1e6 / steps))
