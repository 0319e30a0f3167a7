"""Debug helper: run one conv case through the C ABI and print where it differs from torch."""
import sys, ctypes as C
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from amyloid_yolo_paper_amd import _lib
from amyloid_yolo_paper_amd._lib import ConvDesc, check
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
def bf16r(t): return t.to(torch.bfloat16).to(torch.float32)
cin, cout, k, stride, H, B = [int(v) for v in sys.argv[1:7]]
has_res = len(sys.argv) > 7 and sys.argv[7] == "res"
dev = torch.device("cuda:0"); L = _lib.lib(); st = _lib.stream_ptr()
g = torch.Generator().manual_seed(1)
x = bf16r(torch.randn(B, cin, H, H, generator=g)); w = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
scale = torch.rand(cout, generator=g) + 0.5; shift = torch.randn(cout, generator=g) * 0.1
pad = (k - 1) // 2; Ho = (H + 2 * pad - k) // stride + 1
res = bf16r(torch.randn(B, cout, Ho, Ho, generator=g)) if has_res else None
ref = F.leaky_relu(F.conv2d(x, bf16r(w), None, stride, pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), 0.1)
if has_res: ref = ref + res
cpad = (cout + 31) // 32 * 32
xb = torch.empty(B, cin // 16, H, H, 16, device=dev, dtype=torch.bfloat16); xd, wd = x.to(dev), w.to(dev)
check(L.ay_nchw_f32_to_blocked_bf16(ptr(xd), ptr(xb), B, cin, H, H, st))
packed = torch.empty(L.ay_packed_weight_bytes(cpad, cin, k), device=dev, dtype=torch.uint8)
check(L.ay_pack_conv_weights_bf16(ptr(wd), ptr(packed), cout, cpad, cin, k, st))
sc = torch.zeros(cpad, device=dev); sh = torch.zeros(cpad, device=dev); sc[:cout], sh[:cout] = scale.to(dev), shift.to(dev)
rb = None
if has_res:
    rb = torch.empty(B, cpad // 16, Ho, Ho, 16, device=dev, dtype=torch.bfloat16); rd = res.to(dev)
    check(L.ay_nchw_f32_to_blocked_bf16(ptr(rd), ptr(rb), B, cout, Ho, Ho, st))
for rep in range(3):
    ob = torch.full((B, cpad // 16, Ho, Ho, 16), float("nan"), device=dev, dtype=torch.bfloat16)
    d = ConvDesc(B, cin, cout, H, H, Ho, Ho, k, stride, 1, 0, cpad)
    check(L.ay_conv_fwd_bf16(C.byref(d), ptr(xb), ptr(packed), ptr(sc), ptr(sh), ptr(rb), ptr(ob), st))
    got = torch.empty(B, cout, Ho, Ho, device=dev)
    check(L.ay_blocked_bf16_to_nchw_f32(ptr(ob), ptr(got), B, cout, Ho, Ho, st))
    got = got.cpu(); r = bf16r(ref)
    bad = ~((got - r).abs() <= r.abs() * 2 ** -7 + 2e-3)
    idx = bad.nonzero()
    print("rep", rep, "bad", int(bad.sum()), "of", bad.numel(), "nan", int(torch.isnan(got).sum()))
    if len(idx):
        print(" b", sorted(set(idx[:, 0].tolist())), "\n ch", sorted(set(idx[:, 1].tolist()))[:40], "\n y", sorted(set(idx[:, 2].tolist())), "\n x", sorted(set(idx[:, 3].tolist())))
        print(" sample", [(tuple(i.tolist()), float(got[tuple(i)]), float(r[tuple(i)])) for i in idx[:6]])
