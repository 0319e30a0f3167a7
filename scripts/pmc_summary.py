"""Sums rocprofv3 --pmc counters per kernel name.  usage: python scripts/pmc_summary.py DIR [DIR ...]  (counter_collection.csv of each pass)"""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
passes = collections.defaultdict(lambda: collections.defaultdict(set))
calls = collections.Counter()
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            passes[k][r["Counter_Name"]].add(d)
            seen.add((k, r["Dispatch_Id"]))
        for k, _ in seen:
            calls[k] = max(calls[k], sum(1 for kk, _ in seen if kk == k))
for k in agg:  # a counter collected in several passes: mean over the passes
    for c in agg[k]:
        agg[k][c] /= len(passes[k][c])
names = sorted({c for v in agg.values() for c in v})
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    if not any(t in k for t in ("conv", "resblock", "stem", "wgrad", "bn_")):
        continue
    print(f"{k}  (n={calls[k]})")
    base = v.get("SQ_WAVE_CYCLES") or v.get("SQ_BUSY_CYCLES") or 1.0
    for c in names:
        if c in v:
            print(f"    {c:34s} {v[c]:16.4g}   /SQ_WAVE_CYCLES {v[c] / base if 'SQ_WAVE_CYCLES' in v else float('nan'):8.4f}")
