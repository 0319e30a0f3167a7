"""Sums rocprofv3 --pmc counters per kernel name.  usage: python scripts/pmc_summary.py DIR [DIR ...]  (counter_collection.csv of each pass)"""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
passes = collections.defaultdict(lambda: collections.defaultdict(set))
calls = collections.Counter()
dur = collections.defaultdict(lambda: collections.defaultdict(float))   # kernel -> pass dir -> summed dispatch time [ns] (profiled pass)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            passes[k][r["Counter_Name"]].add(d)
            if (k, r["Dispatch_Id"]) not in seen and r.get("Start_Timestamp") and r.get("End_Timestamp"):
                dur[k][d] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
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
    # derived (MI355X_MICROARCH.md, DVFS give-back): GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts cycles
    # per SIMD (1024 SIMDs); both from the SAME profiled pass where possible
    if "GRBM_GUI_ACTIVE" in v:
        d_gui = sorted(passes[k]["GRBM_GUI_ACTIVE"])[0]
        if dur[k].get(d_gui):
            print(f"    {'-> clock held [GHz] (GUI_ACTIVE/8/t)':34s} {v['GRBM_GUI_ACTIVE'] / 8.0 / dur[k][d_gui]:16.3f}   (profiled pass, {dur[k][d_gui] / calls[k] / 1e3:.1f} us per launch)")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            print(f"    {'-> MFMA pipes busy (of 1024 SIMDs)':34s} {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0):16.3f}")
