"""GPU (-m gpu): kernels of the bf16 MFMA training path against torch-CPU autograd on bf16-rounded operands."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from amyloid_yolo_paper_amd import _lib
from amyloid_yolo_paper_amd._lib import ConvDesc, check, ptr

pytestmark = pytest.mark.gpu


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def to_blocked(t, dev, cpad=None):
    """NCHW f32 (CPU) -> blocked bf16 device tensor [B][C/16][H][W][16]"""
    L = _lib.lib()
    B, Cc, H, W = t.shape
    cp = (Cc + 15) // 16 if cpad is None else cpad // 16
    out = torch.zeros(B, cp, H, W, 16, device=dev, dtype=torch.bfloat16)
    td = t.to(dev).contiguous()
    check(L.ay_nchw_f32_to_blocked_bf16(ptr(td), ptr(out), B, Cc, H, W, _lib.stream_ptr()))
    torch.cuda.synchronize()
    return out


def from_blocked(tb, Cc):
    L = _lib.lib()
    B, _, H, W, _ = tb.shape
    out = torch.empty(B, Cc, H, W, device=tb.device)
    check(L.ay_blocked_bf16_to_nchw_f32(ptr(tb), ptr(out), B, Cc, H, W, _lib.stream_ptr()))
    return out.cpu()


WGRAD_CASES = [(3, 32, 3, 1, 40, 2), (32, 64, 3, 1, 40, 2), (64, 128, 3, 1, 32, 3), (128, 256, 3, 2, 26, 2), (256, 128, 1, 1, 26, 2),
               (32, 64, 3, 2, 64, 2), (1024, 24, 1, 1, 13, 2), (16, 48, 3, 1, 9, 1), (512, 1024, 3, 1, 8, 2)]


@pytest.mark.parametrize("case", WGRAD_CASES, ids=lambda c: "x".join(map(str, c)))
def test_wgrad_mfma(case):
    """dW from the tr-read MFMA kernel vs autograd of F.conv2d on the same bf16 operands (fp32 accumulate on both sides;
    split-K atomics change the order): 2e-3 of the gradient scale."""
    cin, cout, k, s, H, B = case
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(cin + cout + H)
    x = bf(torch.randn(B, cin, H, H, generator=g))
    w = torch.zeros(cout, cin, k, k, requires_grad=True)
    y = F.conv2d(x, w, None, s, (k - 1) // 2)
    dz = bf(torch.randn(y.shape, generator=g))
    y.backward(dz)
    Ho = y.shape[2]
    cpad = (cout + 31) // 32 * 32
    xb, dzb = to_blocked(x, dev), to_blocked(dz, dev, cpad)
    dw = torch.full((cout, cin, k, k), float("nan"), device=dev)
    d = ConvDesc(B, cin, cout, H, H, Ho, Ho, k, s, 0, 0, cpad)
    check(L.ay_conv_wgrad_bf16(C.byref(d), ptr(xb), ptr(dzb), ptr(dw), _lib.stream_ptr()), "wgrad")
    got, ref = dw.cpu(), w.grad
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= 2e-3 * float(ref.abs().max()), float((got - ref).abs().max() / ref.abs().max())


DGRAD_CASES = [(32, 64, 3, 1, 40), (128, 256, 3, 1, 13), (256, 128, 1, 1, 26), (1024, 24, 1, 1, 13), (64, 128, 3, 2, 32), (128, 256, 3, 2, 26)]


@pytest.mark.parametrize("case", DGRAD_CASES, ids=lambda c: "x".join(map(str, c)))
def test_dgrad_through_forward_kernel(case):
    """data gradient = the forward MFMA kernel on re-packed (flipped, transposed) filters; stride 2 through zero insertion;
    accumulation into an existing gradient through the residual operand."""
    cin, cout, k, s, H = case
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    st = _lib.stream_ptr()
    B = 2
    g = torch.Generator().manual_seed(cin * 3 + cout + H)
    w = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cout * k * k)
    x = torch.zeros(B, cin, H, H, requires_grad=True)
    y = F.conv2d(x, bf(w), None, s, (k - 1) // 2)
    dz = bf(torch.randn(y.shape, generator=g))
    y.backward(dz)
    Ho = y.shape[2]
    prev = bf(torch.randn(B, cin, H, H, generator=g))           # gradient already accumulated from another consumer
    ref = bf(x.grad + prev)
    cpad = (cout + 31) // 32 * 32                                 # planes of dz
    cin_pad = (cin + 31) // 32 * 32
    dzb = to_blocked(dz, dev, cpad)
    if s == 2:
        up = torch.empty(B, cpad // 16, H, H, 16, device=dev, dtype=torch.bfloat16)
        check(L.ay_zero_insert_bf16(ptr(dzb), ptr(up), B, cpad, Ho, Ho, H, H, st))
        dzb = up
    wd = w.to(dev)
    kin = (cout + 15) // 16 * 16
    packed = torch.empty((kin // 16) * k * k * 2 * cin_pad * 8 * 2, device=dev, dtype=torch.uint8)
    check(L.ay_pack_dgrad_weights_bf16(ptr(wd), ptr(packed), cout, cin, cin_pad, k, st))
    ones, zeros = torch.ones(cin_pad, device=dev), torch.zeros(cin_pad, device=dev)
    dx = to_blocked(prev, dev, cin_pad)
    d = ConvDesc(B, kin if kin == cpad else cpad, cin, H, H, H, H, k, 1, 0, 0, cin_pad)
    check(L.ay_conv_fwd_bf16(C.byref(d), ptr(dzb), ptr(packed), ptr(ones), ptr(zeros), ptr(dx), ptr(dx), st), "dgrad")
    got = from_blocked(dx, cin)
    err = (got - ref).abs()
    assert bool((err <= ref.abs() * 2.0 ** -7 + 2e-3).all()), float(err.max())


def test_bn_train_bf16_fwd_bwd_and_plumbing():
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    st = _lib.stream_ptr()
    g = torch.Generator().manual_seed(4)
    B, Cc, H = 3, 48, 10
    z = bf(torch.randn(B, Cc, H, H, generator=g) * 2 + 0.3)
    skip = bf(torch.randn(B, Cc, H, H, generator=g))
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    rm, rv = torch.zeros(Cc), torch.ones(Cc)
    dy = bf(torch.randn(B, Cc, H, H, generator=g))
    zr, gr, br = z.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    pre = F.batch_norm(zr, rm_ref, rv_ref, gr, br, True, 0.9, 1e-5)
    yr = F.leaky_relu(pre, 0.1) + skip
    yr.backward(dy)
    zb, sb, dyb = to_blocked(z, dev), to_blocked(skip, dev), to_blocked(dy, dev)
    gd, bd, rmd, rvd = gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev)
    yb = torch.empty_like(zb)
    mean, invstd = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    ws = torch.empty(2 * Cc, device=dev, dtype=torch.float64)
    check(L.ay_bn_train_fwd_bf16(ptr(zb), ptr(gd), ptr(bd), ptr(rmd), ptr(rvd), C.c_float(0.9), C.c_float(1e-5), 1, ptr(sb), ptr(yb), ptr(mean),
                                 ptr(invstd), ptr(ws), B, Cc, H, H, st))
    got = from_blocked(yb, Cc)
    want = bf(yr.detach())
    assert bool(((got - want).abs() <= want.abs() * 2.0 ** -7 + 1e-3).all())
    assert float((rmd.cpu() - rm_ref).abs().max()) < 1e-5 and float((rvd.cpu() - rv_ref).abs().max()) < 1e-4
    dzb = torch.empty_like(zb)
    dg, db = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    check(L.ay_bn_train_bwd_bf16(ptr(dyb), ptr(zb), ptr(gd), ptr(bd), ptr(mean), ptr(invstd), 1, ptr(dzb), ptr(dg), ptr(db), ptr(ws), B, Cc, H, H, st))
    assert float((dg.cpu() - gr.grad).abs().max()) <= 1e-3 * float(gr.grad.abs().max())
    assert float((db.cpu() - br.grad).abs().max()) <= 1e-3 * float(br.grad.abs().max())
    gz = from_blocked(dzb, Cc)
    assert bool(((gz - zr.grad).abs() <= zr.grad.abs() * 2.0 ** -7 + 2e-3).all())
    # accumulate + slice/upsample backward
    a, b_ = bf(torch.randn(2, 32, 4, 4, generator=g)), bf(torch.randn(2, 32, 4, 4, generator=g))
    ab, bb = to_blocked(a, dev), to_blocked(b_, dev)
    check(L.ay_accumulate_bf16(ptr(ab), ptr(bb), ab.numel(), st))
    assert torch.equal(from_blocked(ab, 32), bf(a + b_))
    dout = bf(torch.randn(2, 48, 8, 8, generator=g))
    db1 = to_blocked(dout, dev)
    d1 = torch.zeros(2, 1, 4, 4, 16, device=dev, dtype=torch.bfloat16)
    check(L.ay_slice_accumulate_bf16(ptr(db1), ptr(d1), 2, 16, 48, 0, 8, 8, 1, 0, st))
    want = bf(dout[:, :16].reshape(2, 16, 4, 2, 4, 2).sum((3, 5)))
    assert float((from_blocked(d1, 16) - want).abs().max()) <= 2e-2
    d2 = to_blocked(bf(torch.ones(2, 32, 8, 8)), dev)
    check(L.ay_slice_accumulate_bf16(ptr(db1), ptr(d2), 2, 32, 48, 16, 8, 8, 0, 1, st))
    assert torch.equal(from_blocked(d2, 32), bf(dout[:, 16:] + 1.0))


def test_bf16_train_step_tracks_fp32_step(tmp_cfg_dir):
    """One training step on the bf16 MFMA path vs the fp32 reference-precision path (same model, weights, batch).

    Every kernel of the bf16 path is pinned against autograd above; end to end the two paths cannot agree tightly below a
    LeakyReLU: bf16 activations (relative noise 2^-9) put ~0.5-1 % of the pre-activations of every layer on the other
    side of zero, where the slope differs 10x, so the gradients are evaluated at slightly different points of a
    piecewise-linear function and decorrelate by ~0.3 % cosine per layer on the way down (measured: 0.9997 at the heads,
    0.98-0.99 one block below, ~0.67 at layer 0 after 75 layers; scripts/dbg_train_bf16.py prints the whole profile).
    Checked here: loss within 10 %, heads >= 0.995, the block under each head >= 0.92, everything else >= 0.35 and finite,
    BN statistics close, and 8 Adam steps on a fixed batch reduce the loss on both paths to within 25 % of each other."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from amyloid_yolo_paper_amd import cfg_gen, parse_config, synth
    from amyloid_yolo_paper_amd.models import Darknet
    C_, S, B = 3, 256, 4
    cfg = cfg_gen.write_cfg(C_, tmp_cfg_dir)
    defs = parse_config.parse_model_config(cfg)
    wpath = os.path.join(tmp_cfg_dir, f"synth_c{C_}.weights")
    if not os.path.exists(wpath):
        synth.write_darknet_weights(wpath, defs, synth.synth_params(defs, seed=7), seen=0)
    x = torch.from_numpy(synth.synth_tiles(B, S, 10))
    tg = torch.from_numpy(synth.synth_targets(B, C_, seed=21, max_per_tile=6, min_per_tile=3, wh_range=(0.05, 0.4), grid=S // 8))
    res = {}
    for prec in ("fp32", "bf16"):
        m = Darknet(cfg, precision=prec).to("cuda")
        m.load_darknet_weights(wpath)
        m.train()
        loss, out = m(x, tg)
        loss.backward()
        grads = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()}
        stats = {n: b.detach().float().cpu() for n, b in m.named_buffers() if "running" in n}
        first = float(loss.item())
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        opt.step()
        opt.zero_grad()
        for _ in range(7):
            loss, _ = m(x, tg)
            loss.backward()
            opt.step()
            opt.zero_grad()
        res[prec] = (first, grads, stats, out, float(loss.item()))
    l32, g32, s32, o32, e32 = res["fp32"]
    l16, g16, s16, o16, e16 = res["bf16"]
    assert abs(l16 - l32) <= 0.10 * abs(l32), (l16, l32)
    assert e32 < l32 and e16 < l16 and abs(e16 - e32) <= 0.25 * abs(e32), (l32, e32, l16, e16)   # both learn, alike

    def cos(n):
        a, b_ = g16[n].reshape(-1), g32[n].reshape(-1)
        assert torch.isfinite(a).all(), n
        return float(torch.dot(a, b_) / (a.norm() * b_.norm() + 1e-30))

    for n in ("module_list.105.conv_105.weight", "module_list.93.conv_93.weight", "module_list.81.conv_81.weight"):
        assert cos(n) >= 0.995, (n, cos(n))
    for n in ("module_list.104.conv_104.weight", "module_list.92.conv_92.weight", "module_list.80.conv_80.weight"):
        assert cos(n) >= 0.92, (n, cos(n))
    for n in g32:
        if ".conv_" in n and n.endswith("weight"):
            assert cos(n) >= 0.35, (n, cos(n))
        else:   # BN / bias gradients are heavily cancelling sums (noise dominated in the early layers): finite is all we ask
            assert torch.isfinite(g16[n]).all(), n
    for n in ("module_list.1.batch_norm_1.running_mean", "module_list.80.batch_norm_80.running_var"):
        assert float((s16[n] - s32[n]).abs().max()) <= 0.10 * float(s32[n].abs().max()) + 1e-3, n
    d = np.abs(o16.numpy()[..., 4:] - o32.numpy()[..., 4:])
    assert np.quantile(d, 0.95) <= 0.1, float(np.quantile(d, 0.95))   # sigmoid outputs of gain-amplified logits: bulk agreement only
