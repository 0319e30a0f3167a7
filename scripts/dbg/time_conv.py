"""Times one bf16 convolution through the C ABI (HIP events, 20 launches).  usage: time_conv.py cin cout k stride H B [res]"""
import sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, ".")
from amyloid_yolo_paper_amd import _lib
from amyloid_yolo_paper_amd._lib import ConvDesc, check
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
cin, cout, k, stride, H, B = [int(v) for v in sys.argv[1:7]]
has_res = len(sys.argv) > 7 and sys.argv[7] == "res"
dev = torch.device("cuda:0"); L = _lib.lib(); st = _lib.stream_ptr()
pad = (k - 1) // 2; Ho = (H + 2 * pad - k) // stride + 1
xb = torch.randn(B, cin // 16, H, H, 16, device=dev).to(torch.bfloat16)
w = torch.randn(cout, cin, k, k, device=dev) / np.sqrt(cin * k * k)
packed = torch.empty(L.ay_packed_weight_bytes(cout, cin, k), device=dev, dtype=torch.uint8)
check(L.ay_pack_conv_weights_bf16(ptr(w), ptr(packed), cout, cout, cin, k, st))
sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
rb = torch.randn(B, cout // 16, Ho, Ho, 16, device=dev).to(torch.bfloat16) if has_res else None
ob = torch.empty(B, cout // 16, Ho, Ho, 16, device=dev, dtype=torch.bfloat16)
d = ConvDesc(B, cin, cout, H, H, Ho, Ho, k, stride, 1, 0, cout)
for _ in range(3):
    check(L.ay_conv_fwd_bf16(C.byref(d), ptr(xb), ptr(packed), ptr(sc), ptr(sh), ptr(rb), ptr(ob), st))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    check(L.ay_conv_fwd_bf16(C.byref(d), ptr(xb), ptr(packed), ptr(sc), ptr(sh), ptr(rb), ptr(ob), st))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
fl = 2.0 * B * Ho * Ho * cout * cin * k * k
print("%4d->%4d k%d s%d @%4d B%d %s: %.1f us  %.0f TFLOP/s" % (cin, cout, k, stride, H, B, "res" if has_res else "   ", ms * 1e3, fl / ms / 1e9))
