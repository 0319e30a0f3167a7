"""Durations of the weight-gradient launches of the last training step in a rocprofv3 --kernel-trace CSV, in launch order."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
w = [r for r in rows if "wgrad_bf16_kernel" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 75
tot = 0.0
for r in w[-n:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(f"{r['Kernel_Name'][17:45]:30s} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8s}  {d:9.1f} us")
print("total", tot)
