"""Per-layer HBM traffic of the last bench step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
FETCH_SIZE is doubled (gfx950 counts 128-B requests of wide coalesced streams at 64 B: MI355X_MICROARCH.md §HBM).
usage: python scripts/hbm_traffic.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> [--json out.json]"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amyloid_yolo_paper_amd import cfg_gen
from amyloid_yolo_paper_amd.models import Darknet


def load(d, cname):
    f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0]
    out = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == cname:
            out[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    items = list(out.values())
    idx = [i for i, v in enumerate(items) if "stem" in v[0]]
    return items[idx[-1]:]


fs, ws = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
m = Darknet(cfg_gen.write_cfg(3, "/tmp/cfgt"))
convs = [(i, e) for i, e in enumerate(m._graph) if e["type"] == "convolutional"]
ci = 0
tf = tw = ti = 0.0
fam_bytes, fam_n = 0.0, 0
print("layer                          fetch MB (x2)  write MB | algorithmic: in MB  out MB  residual MB | fetch/(in+res)")
for (n, fv), (n2, wv) in zip(fs, ws):
    assert n == n2
    if "conv_bf16" in n or "conv3x3_m16" in n or "stem" in n or "resblock" in n:
        i, e = convs[ci]; ci += 1
        if "resblock" in n:  # 1x1 + 3x3 + shortcut in one kernel: x in, out out, the 1x1's output never leaves the CU
            i, e1 = convs[ci]; ci += 1
            e = dict(e1, cin=e["cin"], fuse_into_shortcut=False)
        fused_stem = "stem_s2_fused" in n   # layers 0 and 1 in one kernel: fp32 image in, layer-1 output out
        if fused_stem:
            i, e = convs[ci]; ci += 1
            e = dict(e, cin=3, stride=2)
        S = 1024 >> e["log2_down"]; Sin = S * e["stride"]
        inb = 64 * Sin * Sin * e["cin"] * (4 if (i == 0 or fused_stem) else 2) / 1e6
        outb = 64 * S * S * max(e["cout"], 32) * (4 if not e["bn"] else 2) / 1e6
        resb = outb if e["fuse_into_shortcut"] else 0
        fmb, wmb = fv * 1024 * 2 / 1e6, wv * 1024 / 1e6
        tf += fmb; tw += wmb; ti += inb + outb + resb
        if e["k"] == 3 and e["stride"] == 1 and e["cout"] % 128 == 0:
            fam_bytes += (fmb + wmb) * 1e6; fam_n += 1
        print(f"L{i:3d} {e['cin']:4d}->{e['cout']:4d} k{e['k']} s{e['stride']} @{S:4d}  {fmb:9.0f} {wmb:9.0f} | {inb:8.0f} {outb:7.0f} {resb:7.0f} | {fmb/(inb+resb):5.2f}")
print(f"total fetch {tf/1e3:.2f} GB  write {tw/1e3:.2f} GB  algorithmic {ti/1e3:.2f} GB")
print(f"3x3 s1 128-channel family: {fam_n} launches, {fam_bytes/fam_n/1e6:.1f} MB HBM traffic per launch")
if "--json" in sys.argv:
    import bench
    json.dump({"conv3x3s1_bn128_bytes_per_launch": round(fam_bytes / fam_n),
               "csrc_sha16": bench.csrc_sha16(), "git_head": os.environ.get("AY_GIT_HEAD", "unknown"),
               "source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + --pmc WRITE_SIZE, separate passes, bench.py B=64",
               "total_fetch_gb_per_step": round(tf / 1e3, 2), "total_write_gb_per_step": round(tw / 1e3, 2)},
              open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
