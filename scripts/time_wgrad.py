"""Time one ay_conv_wgrad_bf16 launch.  usage: python scripts/time_wgrad.py cin cout H B [k stride]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amyloid_yolo_paper_amd import _lib
from amyloid_yolo_paper_amd._lib import ConvDesc, check, ptr
cin, cout, H, B = (int(v) for v in sys.argv[1:5])
k = int(sys.argv[5]) if len(sys.argv) > 5 else 3
st_ = int(sys.argv[6]) if len(sys.argv) > 6 else 1
L = _lib.lib()
dev = torch.device("cuda", 0)
Ho = H // st_
x = torch.randn(B, cin // 16, H, H, 16, device=dev).to(torch.bfloat16)
dz = torch.randn(B, cout // 16, Ho, Ho, 16, device=dev).to(torch.bfloat16)
dw = torch.empty(cout, cin, k, k, device=dev)
d = ConvDesc(B, cin, cout, H, H, Ho, Ho, k, st_, 1, 0, cout)
s = _lib.stream_ptr()
for _ in range(3):
    check(L.ay_conv_wgrad_bf16(C.byref(d), ptr(x), ptr(dz), ptr(dw), s))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
N = 10
for _ in range(N):
    check(L.ay_conv_wgrad_bf16(C.byref(d), ptr(x), ptr(dz), ptr(dw), s))
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
fl = 2.0 * B * Ho * Ho * cin * cout * k * k
print(f"wgrad {cin}->{cout} k{k} s{st_} H{H} B{B}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TF/s  AY_WDBG={os.environ.get('AY_WDBG')}")
