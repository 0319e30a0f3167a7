"""HBM traffic per launch of the training step's dominant kernel family (weight gradient of the 3x3 stride-1 layers) from two
rocprofv3 --pmc passes over `bench.py --mode train` (FETCH_SIZE, WRITE_SIZE; FETCH_SIZE doubled: gfx950 counts the 128-B
requests of wide coalesced streams at 64 B, MI355X_MICROARCH.md section HBM).
usage: python scripts/hbm_traffic_train.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> [--json out.json]"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(d, cname):
    f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0]
    out = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == cname:
            out[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    return list(out.values())


fs, ws = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
groups = collections.OrderedDict()
for (n, fv), (n2, wv) in zip(fs, ws):
    assert n == n2, (n, n2)
    key = n.split("(")[0][:70]
    g = groups.setdefault(key, [0, 0.0, 0.0])
    g[0] += 1
    g[1] += fv * 1024 * 2
    g[2] += wv * 1024
tot_f = sum(g[1] for g in groups.values())
tot_w = sum(g[2] for g in groups.values())
print(f"{'kernel':72s} launches   fetch GB (x2)   write GB   MB per launch")
for k, (c, f, w) in sorted(groups.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:24]:
    print(f"{k:72s} {c:6d} {f / 1e9:13.2f} {w / 1e9:10.2f} {(f + w) / c / 1e6:12.1f}")
print(f"all kernels of the profiled run (warm-up + timed steps): fetch {tot_f / 1e9:.1f} GB, write {tot_w / 1e9:.1f} GB")
fam = [(c, f, w) for k, (c, f, w) in groups.items() if "wgrad_bf16_kernel<3, 1," in k]
fc, ff, fw = (sum(x[i] for x in fam) for i in range(3))
print(f"weight gradient of the 3x3 s1 family: {fc} launches, {(ff + fw) / fc / 1e6:.1f} MB HBM traffic per launch")
if "--json" in sys.argv:
    import bench
    json.dump({"wgrad3x3_bytes_per_launch": round((ff + fw) / fc), "csrc_sha16": bench.csrc_sha16(), "git_head": os.environ.get("AY_GIT_HEAD", "unknown"),
               "source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + --pmc WRITE_SIZE, separate passes, bench.py --mode train B=32 1024^2"},
              open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
