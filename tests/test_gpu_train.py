"""GPU (-m gpu): the training step on the HIP path (fp32 reference-precision) against the reference's own training
step (tests/golden/train_*.npz, produced by oracle/gen_golden.py from the imported reference) and the CPU oracle."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import golden_cases as gc
from amyloid_yolo_paper_amd import _lib, cfg_gen, parse_config, synth
from amyloid_yolo_paper_amd._lib import check, ptr
from amyloid_yolo_paper_amd.models import Darknet
from oracle import boxes_oracle as bo

pytestmark = pytest.mark.gpu


def close(a, b, tol, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    assert err.max(initial=0.0) <= tol, (what, float(err.max()))


def _model(C_, cfg_dir):
    cfg = cfg_gen.write_cfg(C_, cfg_dir)
    defs = parse_config.parse_model_config(cfg)
    wpath = os.path.join(cfg_dir, f"synth_c{C_}.weights")
    if not os.path.exists(wpath):
        synth.write_darknet_weights(wpath, defs, synth.synth_params(defs, seed=7), seen=12345)
    m = Darknet(cfg, precision="fp32").to("cuda")
    m.load_darknet_weights(wpath)
    return m


GRAD_LAYERS = (0, 1, 2, 42, 73, 80, 81, 93, 104, 105)
_SPREAD = {}


def _trimmed_rel_l2(g, ref, keep=0.98):
    """relative L2 over the `keep` share of the elements with the smallest error: a LeakyReLU sign flip moves one filter row (0.1 % of
    a layer) by a lot and everything else by a little; a systematic error of a kernel moves every element"""
    e = np.abs(np.asarray(g, np.float64) - np.asarray(ref, np.float64)).reshape(-1)
    k = max(1, int(keep * e.size))
    idx = np.argpartition(e, k - 1)[:k]
    return float(np.linalg.norm(e[idx]) / max(np.linalg.norm(np.asarray(ref, np.float64).reshape(-1)[idx]), 1e-30))


def _contract_spread(tmp_cfg_dir):
    """What the fp32 contract itself leaves open, measured on the CPU oracle: the training step of every TRAIN_CASE evaluated twice --
    convolution sums in fp32 (ATen's order, = the reference) and in float64 -- and, per checked layer, the trimmed relative L2
    distance of the two filter gradients; the maximum over the cases (a flip is a chance event of a case) and, walking from the
    loss down, over the layers above (a layer's gradient carries every flip between it and the loss).  The HIP step is a third
    evaluation of the same contract in yet another summation order: it is held to a small multiple of this spread."""
    if _SPREAD:
        return _SPREAD
    from oracle.darknet_oracle import OracleDarknet
    per_layer = {li: 0.0 for li in GRAD_LAYERS}
    for name, C_, S, B, seed in gc.TRAIN_CASES:
        grads = []
        for f64 in (False, True):
            cfg = cfg_gen.write_cfg(C_, tmp_cfg_dir)
            o = OracleDarknet(cfg)
            o.set_params(synth.synth_params(parse_config.parse_model_config(cfg), seed=7))
            o.conv_f64 = f64
            o.require_grad()
            loss, _ = o.forward(torch.from_numpy(gc.model_inputs(S, B, 10)), torch.from_numpy(gc.train_targets(B, C_, S, seed)), train_bn=True)
            loss.backward()
            grads.append({li: o.params[li]["weight"].grad.numpy() for li in GRAD_LAYERS})
        for li in GRAD_LAYERS:
            per_layer[li] = max(per_layer[li], _trimmed_rel_l2(grads[0][li], grads[1][li]))
    run = 0.0
    for li in sorted(GRAD_LAYERS, reverse=True):
        run = max(run, per_layer[li])
        _SPREAD[li] = run
    return _SPREAD


@pytest.mark.parametrize("case", gc.TRAIN_CASES, ids=lambda c: c[0])
def test_train_step_vs_reference(golden_dir, tmp_cfg_dir, case):
    """loss within 1e-4, the 13 per-layer metrics within 2e-4, sampled gradients within 2e-3 of the layer's gradient
    scale (fp32 sums over up to 10^5 terms in a different order), BN running statistics within 1e-4."""
    name, C_, S, B, seed = case
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    m = _model(C_, tmp_cfg_dir)
    m.train()
    tg = torch.from_numpy(gc.train_targets(B, C_, S, seed))
    x = torch.from_numpy(gc.model_inputs(S, B, 10))
    loss, out = m(x, tg)
    assert not out.is_cuda and out.shape == (B, m.num_boxes(S), 5 + C_)
    # train-mode outputs: batch statistics over as few as B*G*G = 27..32 samples per channel divide a 1e-6 relative
    # summation-order difference of the convolution by a small sigma: 1e-3 here, 1e-4 in the eval-mode parity tests
    close(out.numpy(), z["out"], 1e-3, "train-mode outputs")
    loss.backward()
    close(loss.item(), z["loss"], 1e-4, "loss")
    keys = list(z["metric_keys"])
    got = np.array([[yl.metrics[k] for k in keys] for yl in m.yolo_layers])
    close(got, z["metrics"], 2e-4, "metrics")
    # Gradients.  The three linear heads (no LeakyReLU between them and the loss) must match to 2e-4 of the layer's
    # gradient scale.  Below a LeakyReLU the comparison is inherently looser: a pre-activation within ~1e-5 of zero
    # takes slope 1 on one side and 0.1 on the other, and a different fp32 summation order in the convolution flips
    # the sign of a handful of such elements per step (17 of ~10^7 in this case, scripts/dbg_train.py lists them).
    # Each flip changes dz at one element by up to 10x and spreads from there, so those layers are held to
    # 10 % of the gradient scale element-wise (one flipped sample of 32 moves a whole filter row) and 3 % in relative L2; the kernels themselves are pinned tightly,
    # one by one, in test_backward_kernels_vs_autograd below.
    # On top of those flip allowances every filter gradient is held, in TRIMMED relative L2 (the 2 % of the elements with the largest
    # error left out: the flipped rows), to 2.5x what the contract itself leaves open at that depth (_contract_spread: 3e-5 at layer
    # 80, 0.4 % at 73, 0.8 % from 42 down) -- a kernel that is 2 % off in a deep layer fails this, where the flat 3 % let it pass.
    spread = _contract_spread(tmp_cfg_dir)

    def check_grad(g, ref, tight, what, trimmed_bar=None):
        g, ref = np.asarray(g, np.float64), np.asarray(ref, np.float64)
        scale = max(np.abs(ref).max(), 1e-12)
        if tight:
            assert np.abs(g - ref).max() <= 2e-4 * scale, (what, float(np.abs(g - ref).max() / scale))
        else:
            assert np.abs(g - ref).max() <= 1e-1 * scale, (what, float(np.abs(g - ref).max() / scale))
            assert np.linalg.norm(g - ref) <= 3e-2 * np.linalg.norm(ref), (what, float(np.linalg.norm(g - ref) / np.linalg.norm(ref)))
        if trimmed_bar is not None:
            t = _trimmed_rel_l2(g, ref)
            print(f"{what}: trimmed rel L2 {t:.2e} (bar {trimmed_bar:.2e}), rel L2 {np.linalg.norm(g - ref) / np.linalg.norm(ref):.2e}")
            assert t <= trimmed_bar, (what, t, trimmed_bar)

    for li in GRAD_LAYERS:
        conv = m.module_list[li][0]
        g = conv.weight.grad.cpu().numpy()
        ref = z[f"gw{li}"]
        if ref.shape != g.shape:
            g = g.reshape(-1)[:: max(1, g.size // 65536)]
        tight = li in (81, 93, 105)
        check_grad(g, ref, tight, f"dW{li}", trimmed_bar=max(2.5 * spread[li], 2e-4))
        if conv.bias is not None:
            check_grad(conv.bias.grad.cpu().numpy(), z[f"gb{li}"], tight, f"db{li}")
        else:
            bn = m.module_list[li][1]
            check_grad(bn.weight.grad.cpu().numpy(), z[f"ggamma{li}"], False, f"dgamma{li}")
            check_grad(bn.bias.grad.cpu().numpy(), z[f"gbeta{li}"], False, f"dbeta{li}")
            close(bn.running_mean.cpu().numpy(), z[f"rmean{li}"], 1e-4, "running_mean")
            close(bn.running_var.cpu().numpy(), z[f"rvar{li}"], 1e-4, "running_var")
            assert int(bn.num_batches_tracked) == 1


def test_backward_kernels_vs_autograd():
    """every backward kernel of the fp32 training path against torch-CPU autograd of the same op on random data"""
    import torch.nn.functional as F
    from amyloid_yolo_paper_amd._lib import ConvDesc
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    st = _lib.stream_ptr()
    g = torch.Generator().manual_seed(3)

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / b.abs().max())

    # --- train-mode BN + leaky, forward and backward
    B, Cc, H = 3, 20, 7
    zt = torch.randn(B, Cc, H, H, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    rm, rv = torch.randn(Cc, generator=g), torch.rand(Cc, generator=g) + 0.5
    dy = torch.randn(B, Cc, H, H, generator=g)
    zr, gr, br = zt.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    yr = F.leaky_relu(F.batch_norm(zr, rm_ref, rv_ref, gr, br, True, 0.9, 1e-5), 0.1)
    yr.backward(dy)
    zd, gd, bd, rmd, rvd, dyd = (t.to(dev) for t in (zt, gamma, beta, rm, rv, dy))
    yd, mean, invstd = torch.empty_like(zd), torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    check(L.ay_bn_train_fwd_f32(ptr(zd), ptr(gd), ptr(bd), ptr(rmd), ptr(rvd), C.c_float(0.9), C.c_float(1e-5), 1, ptr(yd), ptr(mean),
                                ptr(invstd), B, Cc, H * H, st))
    assert rel(yd, yr.detach()) < 1e-5 and rel(rmd, rm_ref) < 1e-6 and rel(rvd, rv_ref) < 1e-6
    dzd, dgd, dbd = torch.empty_like(zd), torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    check(L.ay_bn_train_bwd_f32(ptr(dyd), ptr(yd), ptr(zd), ptr(gd), ptr(mean), ptr(invstd), 1, ptr(dzd), ptr(dgd), ptr(dbd), B, Cc, H * H, st))
    assert rel(dzd, zr.grad) < 2e-5 and rel(dgd, gr.grad) < 1e-5 and rel(dbd, br.grad) < 1e-5

    # --- conv dgrad / wgrad / bias grad: 3x3 s1, 3x3 s2 (odd size), 1x1
    for cin, cout, k, s, H in ((5, 7, 3, 1, 9), (6, 4, 3, 2, 11), (8, 3, 1, 1, 6), (4, 6, 3, 2, 8)):
        x = torch.randn(2, cin, H, H, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g)
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yr = F.conv2d(xr, wr, None, s, (k - 1) // 2)
        dz = torch.randn(yr.shape, generator=g)
        yr.backward(dz)
        Ho = yr.shape[2]
        d = ConvDesc(2, cin, cout, H, H, Ho, Ho, k, s, 0, 0, cout)
        xd, wd, dzd = x.to(dev), w.to(dev), dz.to(dev)
        dx = torch.full_like(xd, 1.0)
        check(L.ay_conv_dgrad_f32(C.byref(d), ptr(dzd), ptr(wd), ptr(dx), 1, st))       # accumulate onto ones
        assert rel(dx - 1.0, xr.grad) < 1e-5, (cin, cout, k, s)
        check(L.ay_conv_dgrad_f32(C.byref(d), ptr(dzd), ptr(wd), ptr(dx), 0, st))
        assert rel(dx, xr.grad) < 1e-5
        dw = torch.empty_like(wd)
        check(L.ay_conv_wgrad_f32(C.byref(d), ptr(xd), ptr(dzd), ptr(dw), st))
        assert rel(dw, wr.grad) < 1e-5, (cin, cout, k, s)
        db = torch.empty(cout, device=dev)
        check(L.ay_bias_grad_f32(ptr(dzd), ptr(db), 2, cout, Ho * Ho, st))
        assert rel(db, dz.sum((0, 2, 3))) < 1e-5

    # --- route / upsample copy and its backward, shortcut add, accumulate
    a = torch.randn(2, 4, 3, 3, generator=g)
    b_ = torch.randn(2, 5, 6, 6, generator=g)
    ar, brq = a.clone().requires_grad_(True), b_.clone().requires_grad_(True)
    cat = torch.cat([F.interpolate(ar, scale_factor=2, mode="nearest"), brq], 1)
    dcat = torch.randn(cat.shape, generator=g)
    cat.backward(dcat)
    ad, bd2, od = a.to(dev), b_.to(dev), torch.empty(2, 9, 6, 6, device=dev)
    check(L.ay_copy_channels_f32(ptr(ad), ptr(od), 2, 4, 9, 0, 6, 6, 1, st))
    check(L.ay_copy_channels_f32(ptr(bd2), ptr(od), 2, 5, 9, 4, 6, 6, 0, st))
    assert torch.equal(od.cpu(), cat.detach())
    dcd, da, db2 = dcat.to(dev), torch.zeros_like(ad), torch.zeros_like(bd2)
    check(L.ay_slice_accumulate_f32(ptr(dcd), ptr(da), 2, 4, 9, 0, 6, 6, 1, st))
    check(L.ay_slice_accumulate_f32(ptr(dcd), ptr(db2), 2, 5, 9, 4, 6, 6, 0, st))
    assert rel(da, ar.grad) < 1e-6 and torch.equal(db2.cpu(), brq.grad)
    s1, s2 = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    s1d, s2d, so = s1.to(dev), s2.to(dev), torch.empty(1000, device=dev)
    check(L.ay_add_f32(ptr(s1d), ptr(s2d), ptr(so), 1000, st))
    check(L.ay_accumulate_f32(ptr(s1d), ptr(s2d), 1000, st))
    assert torch.equal(so.cpu(), s1 + s2) and torch.equal(s1d.cpu(), s1 + s2)


def test_yolo_loss_kernel_duplicates_and_grad():
    """loss kernel alone vs the oracle's build_targets + torch autograd, with duplicate (b, anchor, cell) targets
    (last writer wins, multi-hot tcls) and targets in every image."""
    import torch.nn.functional as F
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    B, A, Cc, G, S = 3, 3, 3, 8, 64
    rng = np.random.Generator(np.random.PCG64(5))
    head = torch.from_numpy(rng.normal(0, 1, (B, A * (5 + Cc), G, G)).astype(np.float32))
    tg = np.array([[0, 1, .31, .33, .3, .4], [0, 2, .32, .34, .31, .39], [1, 0, .7, .2, .1, .15], [1, 2, .71, .21, .6, .7],
                   [2, 1, .5, .5, .9, .9], [2, 1, .12, .88, .05, .07], [0, 0, .9, .9, .2, .2]], np.float32)
    anchors = [(10, 13), (16, 30), (33, 23)]
    # oracle: decode + build_targets (NumPy) + the six loss terms with torch autograd on the raw head
    h = head.clone().requires_grad_(True)
    p = h.view(B, A, 5 + Cc, G, G).permute(0, 1, 3, 4, 2)
    sx, sy, w, hh = torch.sigmoid(p[..., 0]), torch.sigmoid(p[..., 1]), p[..., 2], p[..., 3]
    conf, cls = torch.sigmoid(p[..., 4]), torch.sigmoid(p[..., 5:])
    _, boxes, aux = bo.decode(head.numpy(), anchors, Cc, S)
    bt = bo.build_targets(boxes, aux["cls"], tg, aux["scaled_anchors"], 0.5)
    iou_scores, class_mask, obj, noobj, tx, ty, tw, th, tcls, tconf = [torch.from_numpy(np.ascontiguousarray(v)) for v in bt]
    loss = (F.mse_loss(sx[obj], tx[obj]) + F.mse_loss(sy[obj], ty[obj]) + F.mse_loss(w[obj], tw[obj]) + F.mse_loss(hh[obj], th[obj])
            + F.binary_cross_entropy(conf[obj], tconf[obj]) + 100 * F.binary_cross_entropy(conf[noobj], tconf[noobj])
            + F.binary_cross_entropy(cls[obj], tcls[obj]))
    loss.backward()
    hd, td = head.to(dev), torch.from_numpy(tg).to(dev)
    dhead = torch.empty_like(hd)
    sums = torch.empty(16, device=dev)
    ws = torch.empty(L.ay_yolo_loss_workspace_bytes(B, A, Cc, G), device=dev, dtype=torch.uint8)
    an = (C.c_float * 6)(*[float(v) for a in anchors for v in a])
    check(L.ay_yolo_loss_fwd_bwd(ptr(hd), ptr(td), tg.shape[0], B, A, Cc, G, S, an, C.c_float(0.5), C.c_float(1.0), ptr(dhead), ptr(sums),
                                 ptr(ws), ws.numel(), _lib.stream_ptr()))
    s = sums.cpu().numpy().astype(np.float64)
    got = (s[0] + s[1] + s[2] + s[3]) / s[7] + s[4] / s[7] + 100 * s[5] / s[8] + s[6] / (s[7] * Cc)
    assert int(s[7]) == int(obj.sum()) and int(s[8]) == int(noobj.sum())           # same masks (integer-exact)
    assert abs(got - loss.item()) <= 1e-5 * abs(loss.item())
    g = h.grad.numpy()
    assert np.abs(dhead.cpu().numpy() - g).max() <= 1e-5 * np.abs(g).max() + 1e-8
    assert int(s[9]) == int(class_mask[obj].sum())


def test_adam_flat_matches_torch():
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(10007, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref])
    p = p0.to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        gr = torch.randn(10007, generator=g)
        p_ref.grad = gr.clone()
        opt.step()
        grd = gr.to(dev)
        check(L.ay_adam_flat(ptr(p), ptr(grd), ptr(m), ptr(v), p.numel(), C.c_float(1e-3), C.c_float(0.9), C.c_float(0.999),
                             C.c_float(1e-8), step, C.c_float(1.0), _lib.stream_ptr()))
    assert (p.cpu() - p_ref.detach()).abs().max() <= 1e-6


@pytest.mark.parametrize("precision,box_loss", [("fp32", "mse"), ("bf16", "giou")])
def test_train_loop_checkpoint_and_eval(tmp_path, tmp_cfg_dir, precision, box_loss):
    """train() end to end on a tiny on-disk dataset: label txt format, collate, 3 optimiser steps with the reference's
    accumulation rule, per-layer metrics, state_dict checkpoint that the inference path loads back."""
    from PIL import Image
    from amyloid_yolo_paper_amd.train import train
    from amyloid_yolo_paper_amd.test import evaluate
    rng = np.random.Generator(np.random.PCG64(9))
    img_dir, lab_dir = tmp_path / "images", tmp_path / "labels"
    img_dir.mkdir()
    lab_dir.mkdir()
    paths = []
    for i in range(8):
        arr = synth.synth_tile(50 + i, 96)
        p = img_dir / f"t{i}.png"
        Image.fromarray(arr).save(p)
        n = int(rng.integers(1, 4))
        rows = [(int(rng.integers(0, 2)), *rng.uniform(0.2, 0.8, 2), *rng.uniform(0.1, 0.4, 2)) for _ in range(n)]
        (lab_dir / f"t{i}.txt").write_text("\n".join("%d %.6f %.6f %.6f %.6f" % r for r in rows) + "\n")
        paths.append(str(p))
    (tmp_path / "train.txt").write_text("\n".join(paths) + "\n")
    (tmp_path / "valid.txt").write_text("\n".join(paths[:4]) + "\n")
    (tmp_path / "classes.names").write_text("CAA\nCored\n")
    (tmp_path / "custom.data").write_text(f"classes= 2\ntrain={tmp_path}/train.txt\nvalid={tmp_path}/valid.txt\nnames={tmp_path}/classes.names\n")
    cfg = cfg_gen.write_cfg(2, tmp_cfg_dir)
    model, hist = train(epochs=1, batch_size=2, gradient_accumulations=2, model_def=cfg, data_config=str(tmp_path / "custom.data"),
                        n_cpu=0, img_size=96, multiscale_training=False, checkpoint_dir=str(tmp_path / "ckpt"), max_batches=4,
                        precision=precision, box_loss=box_loss)
    assert len(hist) == 4 and all(np.isfinite(hist))
    assert set(model.yolo_layers[0].metrics) == {"loss", "x", "y", "w", "h", "conf", "cls", "cls_acc", "recall50", "recall75",
                                                 "precision", "conf_obj", "conf_noobj", "grid_size"}
    ck = torch.load(str(tmp_path / "ckpt" / "yolov3_ckpt_0.pth"))
    assert len(ck) == 438
    m2 = Darknet(cfg, precision="fp32").to("cuda")
    m2.load_state_dict(ck)
    res = evaluate(m2, str(tmp_path / "valid.txt"), 0.5, 0.001, 0.5, 96, 2)   # API contract: tuple of 5 or None
    assert res is None or len(res) == 5


def test_train_multiscale_on_the_gpu(tmp_path, tmp_cfg_dir):
    """train() with the reference's DEFAULT multi-scale schedule (utils/datasets.py:78-79,131-133: every tenth batch a new side
    from [S - 96, S + 96] in steps of 32; train.py:40 turns it on even when the flag is passed as a string): the bf16 engine keeps
    a per-(batch, size) context, so a change of size mid-run must re-plan activations, statistics buffers and canvases.  12
    batches at base size 160: at least two different sizes are trained on, every loss is finite, the checkpoint loads."""
    import random
    from PIL import Image
    from amyloid_yolo_paper_amd.train import train
    rng = np.random.Generator(np.random.PCG64(19))
    img_dir, lab_dir = tmp_path / "images", tmp_path / "labels"
    img_dir.mkdir()
    lab_dir.mkdir()
    paths = []
    for i in range(8):
        p = img_dir / f"t{i}.png"
        Image.fromarray(synth.synth_tile(70 + i, 160)).save(p)
        rows = [(int(rng.integers(0, 2)), *rng.uniform(0.2, 0.8, 2), *rng.uniform(0.1, 0.4, 2)) for _ in range(int(rng.integers(1, 4)))]
        (lab_dir / f"t{i}.txt").write_text("\n".join("%d %.6f %.6f %.6f %.6f" % r for r in rows) + "\n")
        paths.append(str(p))
    (tmp_path / "train.txt").write_text("\n".join(paths) + "\n")
    (tmp_path / "classes.names").write_text("CAA\nCored\n")
    (tmp_path / "custom.data").write_text(f"classes= 2\ntrain={tmp_path}/train.txt\nvalid={tmp_path}/none.txt\nnames={tmp_path}/classes.names\n")
    cfg = cfg_gen.write_cfg(2, tmp_cfg_dir)
    random.seed(3)   # the schedule draws from Python's global generator, like the reference
    sizes = []
    from amyloid_yolo_paper_amd import train_engine_bf16 as eng
    real = eng._context

    def spy(model, B, S, dev):
        sizes.append(S)
        return real(model, B, S, dev)

    eng._context = spy
    try:
        model, hist = train(epochs=3, batch_size=2, gradient_accumulations=2, model_def=cfg, data_config=str(tmp_path / "custom.data"),
                            n_cpu=0, img_size=160, multiscale_training="True", checkpoint_dir=str(tmp_path / "ckpt"), precision="bf16")
    finally:
        eng._context = real
    assert len(hist) == 12 and all(np.isfinite(hist)), hist
    seen = sorted(set(sizes))
    assert len(seen) >= 2 and all(64 <= s <= 256 and s % 32 == 0 for s in seen), seen
    assert len(model._train_ctx) <= 2                      # the engine keeps the two most recent shapes
    assert len(torch.load(str(tmp_path / "ckpt" / "yolov3_ckpt_2.pth"))) == 438


_BT_NAMES = ["iou_scores", "class_mask", "obj_mask", "noobj_mask", "tx", "ty", "tw", "th", "tcls", "tconf"]


@pytest.mark.parametrize("case", gc.TRAIN_CASES, ids=lambda c: c[0])
def test_build_targets_device_vs_golden(golden_dir, case):
    """utils.build_targets (ay_build_targets) against the reference's own outputs (tests/golden/bt_*.npz): masks exact,
    floats within 1e-6 relative."""
    from amyloid_yolo_paper_amd import utils as U
    name, Cc, S, B, seed = case
    z = np.load(os.path.join(golden_dir, "bt_" + name + ".npz"))
    G = S // 8
    rng = np.random.Generator(np.random.PCG64(seed + 100))
    pb = rng.uniform(0, G, (B, 3, G, G, 4)).astype(np.float32)
    pc = rng.uniform(0, 1, (B, 3, G, G, Cc)).astype(np.float32)
    tg = gc.train_targets(B, Cc, S, seed)
    out = U.build_targets(torch.from_numpy(pb), torch.from_numpy(pc), torch.from_numpy(tg), torch.from_numpy(z["anchors"]), 0.5)
    assert len(out) == 10
    for n, v in zip(_BT_NAMES, out):
        v = v.numpy()
        if n in ("obj_mask", "noobj_mask"):
            assert v.dtype == bool
            np.testing.assert_array_equal(v, z[n])
        else:
            np.testing.assert_allclose(v, z[n], rtol=1e-6, atol=1e-6)


def test_build_targets_device_duplicates_and_empty():
    """duplicate (sample, anchor, cell) targets: last writer wins, classes accumulate; no targets: all-zero / all-noobj."""
    from amyloid_yolo_paper_amd import utils as U
    B, A, Cc, G = 3, 3, 3, 8
    rng = np.random.Generator(np.random.PCG64(9))
    pb = rng.uniform(0, G, (B, A, G, G, 4)).astype(np.float32)
    pc = rng.uniform(0, 1, (B, A, G, G, Cc)).astype(np.float32)
    tg = np.array([[0, 1, .31, .33, .3, .4], [0, 2, .32, .34, .31, .39], [1, 0, .7, .2, .1, .15], [1, 2, .71, .21, .6, .7],
                   [2, 1, .5, .5, .9, .9], [2, 1, .12, .88, .05, .07], [0, 0, .9, .9, .2, .2]], np.float32)
    anc = (np.array([(10, 13), (16, 30), (33, 23)], np.float32) / 8.0).astype(np.float32)
    ref = bo.build_targets(pb, pc, tg, anc, 0.5)
    out = U.build_targets(torch.from_numpy(pb).cuda(), torch.from_numpy(pc).cuda(), torch.from_numpy(tg).cuda(), torch.from_numpy(anc), 0.5)
    for n, v, r in zip(_BT_NAMES, out, ref):
        assert v.is_cuda
        v = v.cpu().numpy()
        if r.dtype == bool:
            np.testing.assert_array_equal(v, r, err_msg=n)
        else:
            np.testing.assert_allclose(v, r, rtol=1e-6, atol=1e-6, err_msg=n)
    assert out[8].cpu().numpy().sum(-1).max() == 2.0  # a multi-hot cell exists in this case
    out = U.build_targets(torch.from_numpy(pb), torch.from_numpy(pc), torch.zeros(0, 6), torch.from_numpy(anc), 0.5)
    assert not out[2].any() and out[3].all() and all(float(out[i].abs().sum()) == 0.0 for i in (0, 1, 4, 5, 6, 7, 8, 9))


def test_yolo_giou_loss_kernel_vs_autograd():
    """GIoU box-loss variant (BASELINE configs[4]; no reference counterpart): loss and d loss / d head against torch
    autograd of the published GIoU formula composed with the reference's conf/cls terms (tolerance 1e-4 relative)."""
    import torch.nn.functional as F
    from oracle.darknet_oracle import giou_cxcywh
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    B, A, Cc, G, S = 3, 3, 3, 16, 128
    rng = np.random.Generator(np.random.PCG64(15))
    head = torch.from_numpy(rng.normal(0, 0.8, (B, A * (5 + Cc), G, G)).astype(np.float32))
    tg = synth.synth_targets(B, Cc, seed=31, max_per_tile=9, min_per_tile=4, wh_range=(0.05, 0.5), grid=G)
    anchors = [(10, 13), (16, 30), (33, 23)]
    h = head.clone().requires_grad_(True)
    p = h.view(B, A, 5 + Cc, G, G).permute(0, 1, 3, 4, 2)
    sx, sy, w, hh = torch.sigmoid(p[..., 0]), torch.sigmoid(p[..., 1]), p[..., 2], p[..., 3]
    conf, cls = torch.sigmoid(p[..., 4]), torch.sigmoid(p[..., 5:])
    _, boxes, aux = bo.decode(head.numpy(), anchors, Cc, S)
    sa = torch.from_numpy(np.asarray(aux["scaled_anchors"], np.float32))
    bt = bo.build_targets(boxes, aux["cls"], tg, aux["scaled_anchors"], 0.5)
    iou_scores, class_mask, obj, noobj, tx, ty, tw, th, tcls, tconf = [torch.from_numpy(np.ascontiguousarray(v)) for v in bt]
    gi = torch.arange(G, dtype=torch.float32).view(1, 1, 1, G).expand(B, A, G, G)[obj]
    gj = torch.arange(G, dtype=torch.float32).view(1, 1, G, 1).expand(B, A, G, G)[obj]
    aw = sa[:, 0].view(1, A, 1, 1).expand(B, A, G, G)[obj]
    ah = sa[:, 1].view(1, A, 1, 1).expand(B, A, G, G)[obj]
    pb = torch.stack((sx[obj] + gi, sy[obj] + gj, torch.exp(w[obj]) * aw, torch.exp(hh[obj]) * ah), 1)
    tb = torch.stack((tx[obj] + gi, ty[obj] + gj, torch.exp(tw[obj]) * aw, torch.exp(th[obj]) * ah), 1)
    loss = ((1.0 - giou_cxcywh(pb, tb)).mean() + F.binary_cross_entropy(conf[obj], tconf[obj])
            + 100 * F.binary_cross_entropy(conf[noobj], tconf[noobj]) + F.binary_cross_entropy(cls[obj], tcls[obj]))
    loss.backward()
    hd, td = head.to(dev), torch.from_numpy(tg).to(dev)
    dhead = torch.empty_like(hd)
    sums = torch.empty(16, device=dev)
    ws = torch.empty(L.ay_yolo_loss_workspace_bytes(B, A, Cc, G), device=dev, dtype=torch.uint8)
    an = (C.c_float * 6)(*[float(v) for a in anchors for v in a])
    check(L.ay_yolo_loss_giou_fwd_bwd(ptr(hd), ptr(td), tg.shape[0], B, A, Cc, G, S, an, C.c_float(0.5), C.c_float(1.0), ptr(dhead),
                                      ptr(sums), ptr(ws), ws.numel(), _lib.stream_ptr()))
    s = sums.cpu().numpy().astype(np.float64)
    assert s[1] == 0 and s[2] == 0 and s[3] == 0
    got = s[0] / s[7] + s[4] / s[7] + 100 * s[5] / s[8] + s[6] / (s[7] * Cc)
    assert abs(got - loss.item()) <= 1e-4 * abs(loss.item()), (got, loss.item())
    g = h.grad.numpy()
    d = dhead.cpu().numpy()
    assert np.abs(d - g).max() <= 1e-4 * np.abs(g).max() + 1e-8, float(np.abs(d - g).max() / np.abs(g).max())
    # the box-coordinate channels really carry gradient (not just conf/cls)
    gv = g.reshape(B, A, 5 + Cc, G, G)[:, :, :4]
    assert np.abs(gv).max() > 1e-4


def test_train_step_giou_end_to_end(tmp_cfg_dir):
    """model.box_loss = 'giou' through model(x, targets) + backward (fp32 path) against the oracle network with the same
    option and torch autograd: loss 1e-4, head-filter gradients 2e-4 of their scale."""
    from oracle.darknet_oracle import OracleDarknet
    C_, S, B = 3, 96, 2
    m = _model(C_, tmp_cfg_dir)
    m.train()
    m.box_loss = "giou"
    cfg = cfg_gen.write_cfg(C_, tmp_cfg_dir)
    o = OracleDarknet(cfg)
    o.set_params(synth.synth_params(parse_config.parse_model_config(cfg), seed=7))
    o.box_loss = "giou"
    o.require_grad()
    tg = torch.from_numpy(gc.train_targets(B, C_, S, 23))
    x = torch.from_numpy(gc.model_inputs(S, B, 12))
    loss, _ = m(x, tg)
    loss.backward()
    lo, _ = o.forward(x, tg, train_bn=True)
    lo.backward()
    close(loss.item(), lo.item(), 1e-4, "giou loss")
    for li in (81, 93, 105):
        g = m.module_list[li][0].weight.grad.cpu().numpy()
        ref = o.params[li]["weight"].grad.numpy()
        assert np.abs(g - ref).max() <= 2e-4 * np.abs(ref).max(), (li, float(np.abs(g - ref).max() / np.abs(ref).max()))
    assert m.yolo_layers[0].metrics["y"] == 0.0 and m.yolo_layers[0].metrics["x"] > 0.0
