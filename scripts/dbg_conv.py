import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from amyloid_yolo_paper_amd import _lib
from amyloid_yolo_paper_amd._lib import ConvDesc, check, ptr
L=_lib.lib(); dev=torch.device('cuda',0); st=_lib.stream_ptr()
def bf(t): return t.to(torch.bfloat16).to(torch.float32)
def run(cin,cout,k,stride,H,leaky,has_res,B=2):
    g=torch.Generator().manual_seed(1)
    x=bf(torch.randn(B,cin,H,H,generator=g)); w=torch.randn(cout,cin,k,k,generator=g)/np.sqrt(cin*k*k)
    scale=torch.rand(cout,generator=g)+0.5; shift=torch.randn(cout,generator=g)*0.1
    pad=(k-1)//2; Ho=(H+2*pad-k)//stride+1
    res=bf(torch.randn(B,cout,Ho,Ho,generator=g)) if has_res else None
    ref=F.conv2d(x,bf(w),None,stride,pad)*scale.view(1,-1,1,1)+shift.view(1,-1,1,1)
    if leaky: ref=F.leaky_relu(ref,0.1)
    pre=ref.clone()
    if has_res: ref=ref+res
    ref=bf(ref)
    cpad=(cout+31)//32*32
    xd=x.to(dev); wd=w.to(dev)
    xb=torch.empty(B,cin//16,H,H,16,device=dev,dtype=torch.bfloat16)
    check(L.ay_nchw_f32_to_blocked_bf16(ptr(xd),ptr(xb),B,cin,H,H,st))
    packed=torch.empty(L.ay_packed_weight_bytes(cpad,cin,k),device=dev,dtype=torch.uint8)
    check(L.ay_pack_conv_weights_bf16(ptr(wd),ptr(packed),cout,cpad,cin,k,st))
    sc=torch.zeros(cpad,device=dev); sh=torch.zeros(cpad,device=dev); sc[:cout]=scale.to(dev); sh[:cout]=shift.to(dev)
    rb=None
    if has_res:
        rd=res.to(dev)
        rb=torch.empty(B,cpad//16,Ho,Ho,16,device=dev,dtype=torch.bfloat16)
        check(L.ay_nchw_f32_to_blocked_bf16(ptr(rd),ptr(rb),B,cout,Ho,Ho,st))
    ob=torch.zeros(B,cpad//16,Ho,Ho,16,device=dev,dtype=torch.bfloat16)
    d=ConvDesc(B,cin,cout,H,H,Ho,Ho,k,stride,int(leaky),0,cpad)
    check(L.ay_conv_fwd_bf16(C.byref(d),ptr(xb),ptr(packed),ptr(sc),ptr(sh),ptr(rb),ptr(ob),st))
    got=torch.empty(B,cout,Ho,Ho,device=dev)
    check(L.ay_blocked_bf16_to_nchw_f32(ptr(ob),ptr(got),B,cout,Ho,Ho,st))
    got=got.cpu()
    err=(got-ref).abs()
    print(case:=(cin,cout,k,stride,H,leaky,has_res),'max err',err.max().item(),'frac bad',(err>ref.abs()*2**-7+1e-3).float().mean().item())
    bad=(err>ref.abs()*2**-7+1e-3)
    if bad.any():
        idx=bad.nonzero()
        print(' bad by batch',[int((idx[:,0]==b).sum()) for b in range(B)])
        print(' bad by ch%16',[int(((idx[:,1]%16)==c).sum()) for c in range(16)])
        print(' bad by ch//16',[int(((idx[:,1]//16)==c).sum()) for c in range(cout//16)])
        print(' first bad', idx[:5].tolist())
        i=idx[0]; print(' got',got[tuple(i)].item(),'ref',ref[tuple(i)].item(),'pre',pre[tuple(i)].item(), 'res', res[tuple(i)].item() if has_res else None)
        # does got equal pre + some other residual element?
for c in [(32,64,3,1,40,True,True),(32,64,3,1,32,True,True),(64,128,3,1,32,True,True),(64,128,3,1,32,True,False),(64,64,1,1,32,True,True)]:
    run(*c)
