"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same seeded
inputs and against the reference-generated fixtures in tests/golden.

Bars: integer/index results bit-exact; fp32 box/IoU floats within 1e-4 relative (|a-b| <= 1e-4*max(1,|b|));
bf16 MFMA convolutions within bf16 rounding of an oracle that applies the same rounding points
(stated per test)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_cases as gc
from amyloid_yolo_paper_amd import _lib, cfg_gen, parse_config, synth
from amyloid_yolo_paper_amd import utils as ay
from amyloid_yolo_paper_amd._lib import ConvDesc, check, ptr
from amyloid_yolo_paper_amd.models import Darknet
from oracle import boxes_oracle as bo
from oracle.darknet_oracle import OracleDarknet

pytestmark = pytest.mark.gpu
TOL = 1e-4


def close(a, b, tol=TOL, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    assert err.max(initial=0.0) <= tol, (what, float(err.max()), int(err.argmax()))


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "run on the GPU box"
    return torch.device("cuda", 0)


# ----------------------------------------------------------------------------------------- box math
def test_box_math_bit_exact(golden_dir, dev):
    z = load(golden_dir, "kat")
    b1, b2 = gc.iou_inputs()
    t1, t2 = torch.from_numpy(b1), torch.from_numpy(b2)
    np.testing.assert_array_equal(ay.bbox_iou(t1, t2).numpy(), z["iou_rand"])
    np.testing.assert_array_equal(ay.bbox_iou(t1, t2, x1y1x2y2=False).numpy(), z["iou_rand_c"])
    np.testing.assert_array_equal(ay.bbox_iou(t1[:1], t2).numpy(), z["iou_bcast"])
    np.testing.assert_array_equal(ay.xywh2xyxy(t1).numpy(), z["xyxy"])
    kat = ay.bbox_iou(torch.tensor([[100., 100, 200, 200]]), torch.tensor([[150., 150, 200, 200], [201, 201, 300, 300], [100, 100, 200, 200]]))
    np.testing.assert_array_equal(kat.numpy(), z["iou"])
    np.testing.assert_array_equal(ay.bbox_wh_iou(torch.tensor([3., 4.]), torch.tensor([[3., 4], [6, 2], [1, 1]])).numpy(), z["whiou"])
    # pairwise + GIoU against the oracle (GIoU itself is pinned by closed-form vectors: test_giou_closed_form_vectors_hip)
    pw = ay.bbox_iou_pairwise(t1[:40], t2[:50]).numpy()
    ref = np.stack([bo.bbox_iou(b1[i:i + 1], b2[:50]) for i in range(40)])
    np.testing.assert_array_equal(pw, ref)
    close(ay.bbox_iou(t1, t2, giou=True).numpy(), bo.bbox_giou(b1, b2), 1e-6)
    rb = ay.rescale_boxes(torch.tensor([[10., 20, 200, 300, .9, .8, 1], [50, 60, 70, 80, .5, .5, 0]]), 416, (1536, 1024))
    close(rb.numpy(), z["rescale"], 1e-6)


# ----------------------------------------------------------------------------------------- decode
def test_decode_kat_and_heads(golden_dir, dev):
    L = _lib.lib()
    z = load(golden_dir, "kat")
    p = torch.zeros(1, 21, 2, 2)
    p[0, 0, 1, 0], p[0, 9, 0, 1], p[0, 18, 1, 1] = 1.0, 0.5, 2.0
    out = torch.empty(1, 12, 7, device=dev)
    an = (C.c_float * 6)(10, 13, 16, 30, 33, 23)
    pd = p.to(dev)
    check(L.ay_yolo_decode(ptr(pd), 0, ptr(out), 1, 3, 2, 2, 64, an, 12, 0, _lib.stream_ptr()))
    close(out.cpu().numpy(), z["decode"], 1e-6)
    # real head tensors of the S=96, C=3 fixture, all three scales into one output tensor
    zz = load(golden_dir, "model_c3_s96_b2")
    anchors = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]
    N = zz["out"].shape[1]
    out = torch.zeros(2, N, 8, device=dev)
    row = 0
    for j in range(3):
        h = torch.from_numpy(zz[f"head{j}"]).to(dev)
        G = h.shape[2]
        an = (C.c_float * 6)(*[float(v) for a in anchors[j] for v in a])
        check(L.ay_yolo_decode(ptr(h), 0, ptr(out), 2, 3, 3, G, 96, an, N, row, _lib.stream_ptr()))
        row += 3 * G * G
    close(out.cpu().numpy(), zz["out"], 1e-5)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("case", [(256, 3, 3, 40, 2), (512, 3, 2, 13, 3), (80, 3, 3, 9, 2), (128, 3, 80, 20, 1), (64, 2, 1, 33, 2)],
                         ids=lambda c: "cin%d_A%d_C%d_G%d_B%d" % c)
def test_head_decode_fused_equals_head_then_decode(dev, case, dtype):
    """ay_head_decode_fwd_* (the detection head's linear 1x1 convolution with the decode of models.py:127-172 in its epilogue; what the
    native plan issues for layers 81+82 / 93+94 / 105+106) against the two launches it replaces, ay_conv_fwd_* (out_f32) +
    ay_yolo_decode: the same bits, for 5 + C = 8 (16-byte row stores), 7, 6 and 85 (scalar stores, several 32-channel groups), both
    staging depths (cin % 64), ragged grids, rows of other heads left untouched."""
    cin, A, Cn, G, B = case
    L = _lib.lib()
    st = _lib.stream_ptr()
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    K = 5 + Cn
    cout = A * K
    cpad = (cout + 31) // 32 * 32
    g = torch.Generator().manual_seed(cin + 13 * K + G)
    x = torch.randn(B, cin, G, G, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) * (1.5 / np.sqrt(cin))
    bias = torch.randn(cout, generator=g)
    xd, wd = x.to(dev), w.to(dev)
    xb = torch.empty(B, cin // 16, G, G, 16, device=dev, dtype=tdt)
    check(getattr(L, f"ay_nchw_f32_to_blocked_{dtype}")(ptr(xd), ptr(xb), B, cin, G, G, st))
    packed = torch.empty(L.ay_packed_weight_bytes(cpad, cin, 1), device=dev, dtype=torch.uint8)
    check(getattr(L, f"ay_pack_conv_weights_{dtype}")(ptr(wd), ptr(packed), cout, cpad, cin, 1, st))
    sc, sh = torch.zeros(cpad, device=dev), torch.zeros(cpad, device=dev)
    sc[:cout], sh[:cout] = 1.0, bias.to(dev)
    anchors = [(10.0, 13.0), (16.0, 30.0), (33.0, 23.0), (30.0, 61.0)][:A]
    anc = (C.c_float * (2 * A))(*[v for a in anchors for v in a])
    img = 32 * G
    n_total = A * G * G + 7
    d = ConvDesc(B, cin, cout, G, G, G, G, 1, 1, 0, 1, cpad)
    head = torch.empty(B, cpad // 16, G, G, 16, device=dev, dtype=torch.float32)
    want = torch.full((B, n_total, K), -7.0, device=dev)
    check(getattr(L, f"ay_conv_fwd_{dtype}")(C.byref(d), ptr(xb), ptr(packed), ptr(sc), ptr(sh), None, ptr(head), st), "head conv")
    check(L.ay_yolo_decode(ptr(head), 1, ptr(want), B, A, Cn, G, img, anc, n_total, 5, st), "decode")
    got = torch.full((B, n_total, K), -7.0, device=dev)
    check(getattr(L, f"ay_head_decode_fwd_{dtype}")(C.byref(d), ptr(xb), ptr(packed), ptr(sc), ptr(sh), A, Cn, img, anc, ptr(got), n_total, 5, st),
          "fused head")
    assert torch.equal(got, want)
    assert bool((got[:, :5] == -7.0).all()) and bool((got[:, 5 + A * G * G:] == -7.0).all()) and bool(torch.isfinite(got[:, 5:5 + A * G * G]).all())


# ----------------------------------------------------------------------------------------- NMS
@pytest.mark.parametrize("name", [c[0] for c in gc.NMS_CASES])
def test_nms_golden(golden_dir, dev, name):
    z = load(golden_dir, "nms_" + name)
    pred, conf_t, nms_t = gc.nms_case_inputs(name)
    t = torch.from_numpy(pred.copy())
    res = ay.non_max_suppression(t, conf_t, nms_t)            # host tensor in, host rows out
    np.testing.assert_array_equal(t.numpy()[0, :8], z["corners0"])  # in-place corner conversion, bit-exact
    o_rows, o_keep, _ = bo.non_max_suppression(pred.copy(), conf_t, nms_t)
    for b in range(pred.shape[0]):
        n = int(z[f"n{b}"])
        if n == 0:
            assert res[b] is None
            continue
        np.testing.assert_array_equal(res.keep_idx[b], z[f"keep{b}"])      # bit-exact box indices (reference)
        np.testing.assert_array_equal(res.keep_idx[b], o_keep[b])          # and the oracle agrees
        got = res[b].numpy()
        close(got, z[f"rows{b}"], 1e-5, name)
        np.testing.assert_array_equal(got[:, 4:], z[f"rows{b}"][:, 4:])    # conf / cls columns untouched


@pytest.mark.parametrize("case", [(2048, [900, 1024, 17, 0], 6, 51), (1500, [640, 1000], 13, 52), (4096, [1023, 1], 80, 53), (600, [600], 5, 54),
                                  (8192, [4096, 1025, 2500, 300], 3, 55), (8192, [4097, 3000], 6, 56), (6000, [2047, 2048, 2049], 13, 57)],
                         ids=lambda c: "N%d_C%d" % (c[0], c[2]))
def test_nms_class_partitions_vs_oracle(dev, case):
    """The LDS path of `nms_merge_kernel` scans four class partitions (class & 3) on four wavefronts.  With more than four classes a
    partition holds several classes (the same-class test inside the scan keeps them apart), with 5 or 13 they are uneven, with 80 a
    partition interleaves twenty; images at exactly 1 024 candidates (the path's limit), a single candidate, none; `max_det` smaller
    than the number of heads (the first max_det heads in score order are kept, the count says how many there were): indices, rows and
    counts against the CPU oracle.  Images with 1 025 .. 4 096 candidates take `nms_merge_mid_kernel` (the same scan, all in LDS, alive
    words one per lane), 4 097 and more the workspace scan: mixed in one batch here."""
    N, cands, Cn, seed = case
    pred = gc.nms_prediction(N, cands, Cn, seed, conf_thres=0.5)
    o_rows, o_keep, _ = bo.non_max_suppression(pred.copy(), 0.5, 0.4)
    res = ay.non_max_suppression(torch.from_numpy(pred.copy()).to(dev), 0.5, 0.4)
    for b in range(len(cands)):
        if o_rows[b] is None:
            assert res[b] is None
            continue
        np.testing.assert_array_equal(res.keep_idx[b], o_keep[b])
        close(res[b].cpu().numpy(), o_rows[b], 1e-5)
        assert int(res.cand_count[b]) == cands[b]
    # truncation: the fused device call with room for 8 heads per image
    from amyloid_yolo_paper_amd.utils import nms_device
    rows, keep, count, cand = nms_device(torch.from_numpy(pred.copy()).to(dev), 0.5, 0.4, 8, slot=3)
    rows, keep, count = rows.cpu().numpy(), keep.cpu().numpy(), count.cpu().numpy()
    for b in range(len(cands)):
        n_all = 0 if o_rows[b] is None else len(o_keep[b])
        assert int(count[b]) == n_all
        k = min(n_all, 8)
        if k:
            np.testing.assert_array_equal(keep[b, :k], o_keep[b][:k])
            close(rows[b, :k], o_rows[b][:k], 1e-5)


def test_nms_device_tensor_and_large(dev):
    """device-resident input; 20k candidates exercises the workspace (non-LDS) sort path."""
    pred = gc.nms_prediction(30000, [20000], 3, 41, conf_thres=0.3)
    # (score ties exist at this size: both sides break them towards the lower original row, by definition)
    o_rows, o_keep, _ = bo.non_max_suppression(pred.copy(), 0.3, 0.45)
    t = torch.from_numpy(pred.copy()).to(dev)
    res = ay.non_max_suppression(t, 0.3, 0.45)
    assert res[0].is_cuda
    np.testing.assert_array_equal(res.keep_idx[0], o_keep[0])
    close(res[0].cpu().numpy(), o_rows[0], 1e-5)
    assert int(res.cand_count[0]) == 20000


# ----------------------------------------------------------------------------------------- conv kernels
def _bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


CONV_CASES = [
    # cin, cout, k, stride, H, leaky, residual, out_f32
    (32, 64, 3, 1, 40, True, True, False),      # BN=64 tile, ragged 40 (not a multiple of 32)
    (64, 128, 3, 1, 32, True, False, False),    # BN=128
    (128, 256, 3, 1, 13, True, True, False),    # 416/32 grid: heavy masking
    (32, 64, 3, 2, 64, True, False, False),     # stride 2, BN=64
    (128, 256, 3, 2, 26, True, False, False),   # stride 2, BN=128, ragged
    (64, 32, 1, 1, 52, True, False, False),     # 1x1 BN=32, NK=4
    (256, 128, 1, 1, 26, True, False, False),   # 1x1 BN=128
    (1024, 24, 1, 1, 13, False, False, True),   # linear head, f32 out, cout pad 24->32
    (256, 255, 1, 1, 8, False, False, True),    # COCO-sized head, 255->256
    (48, 96, 1, 1, 16, True, False, False),     # cin not a multiple of 64 (NK=1 path)
    (512, 1024, 3, 1, 8, True, True, False),    # deep K loop
    (512, 256, 1, 1, 20, True, False, False),   # 1x1 with the 256-channel tile (BN=256), ragged
    (1024, 512, 1, 1, 13, True, False, False),  # 1x1 BN=256, two channel groups, 13x13
    (256, 256, 1, 1, 32, False, False, False),  # 1x1 BN=256, linear
    # canvas tiling (several small images per pixel tile, one-pixel gutters): odd batches leave the last canvas row half empty
    (128, 256, 3, 1, 13, True, True, False, 7),
    (256, 128, 1, 1, 13, True, False, False, 5),
    (512, 256, 1, 1, 13, True, False, False, 9),   # BN=256 1x1
    (64, 64, 3, 1, 5, True, True, False, 11),      # five images across a tile (gx = 5), BN=64
    (128, 128, 3, 1, 15, True, False, False, 4),   # largest image that still fits twice (2 x 16 = 32 columns)
    (128, 256, 3, 1, 26, True, True, False, 5),    # one image across, stacked with one-row gutters: 16x32 tile, deferred stores
    (64, 128, 3, 1, 26, True, False, False, 3),    # the same without residual
    (256, 128, 1, 1, 26, True, False, False, 5),   # 1x1 on the stacked canvas
    (128, 256, 3, 1, 31, True, True, False, 3),    # an image as wide as a tile with its gutter
    (64, 128, 3, 1, 52, True, True, False, 3),     # canvas wider than a tile: three 53-column cells over five tiles
    (128, 64, 1, 1, 52, True, False, False, 5),
    (32, 128, 3, 1, 104, True, False, False, 2),
    (128, 256, 3, 2, 26, True, False, False, 5),   # stride 2 on the canvas: 13x13 outputs, input cells twice as large
    (64, 128, 3, 2, 52, True, False, False, 3),    # BN=128, 26x26 outputs
    (32, 64, 3, 2, 16, True, False, False, 9),     # BN=64, 8x8 outputs
    # 16x16x32-MFMA kernel (3x3 s1, cout % 128 == 0, cin % 32 == 0, >= 16 rows, no canvas): tap pairs + the two-stage straddle
    (128, 256, 3, 1, 40, True, True, False, 2),    # ragged 40: tile rows 16+16+8, columns 32+8; residual (deferred stores)
    (64, 128, 3, 1, 48, True, False, False, 3),    # two stages only (one straddle), three images
    (256, 512, 3, 1, 16, True, True, False, 2),    # exactly one tile row; 4 channel groups; 16 stages
    (128, 128, 3, 1, 20, False, True, False, 1),   # linear, batch 1
    (32, 128, 3, 1, 33, True, False, False, 2),    # a single stage pair, 33 = one pixel into the second tile column / third row
    # 32-channel tile of the ring kernel (stem through the MFMA kernels, data gradients into 32 channels)
    (16, 32, 3, 1, 40, True, False, False, 2),
    (64, 32, 3, 1, 24, False, True, False, 3),     # with accumulation through the residual operand
    (64, 32, 1, 1, 20, False, True, False, 2),
]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv_bf16_kernel(dev, case, dtype):
    """16-bit MFMA block (bfloat16 | IEEE-half storage: the same kernel template) vs torch-CPU fp32 conv of the operands
    rounded to that type (+affine, leaky, residual), rounded once.  Tolerance: 1 ulp of the result (2^-7 relative for
    bf16, 2^-10 for half: a sum that lands next to a rounding boundary may round the other way) + 1e-3 (half: 2e-4) absolute for
    accumulation-order noise."""
    cin, cout, k, stride, H, leaky, has_res, out_f32 = case[:8]
    L = _lib.lib()
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    _bf16r = lambda t: t.to(tdt).to(torch.float32)   # rounding of the storage type under test
    to_blocked = getattr(L, f"ay_nchw_f32_to_blocked_{dtype}")
    st = _lib.stream_ptr()
    B = case[8] if len(case) > 8 else 2
    g = torch.Generator().manual_seed(cin * 7 + cout + k + H)
    x = _bf16r(torch.randn(B, cin, H, H, generator=g))
    w = torch.randn(cout, cin, k, k, generator=g) * (1.0 / np.sqrt(cin * k * k))
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    pad = (k - 1) // 2
    Ho = (H + 2 * pad - k) // stride + 1
    res = _bf16r(torch.randn(B, cout, Ho, Ho, generator=g)) if has_res else None
    ref = F.conv2d(x, _bf16r(w), None, stride, pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if leaky:
        ref = F.leaky_relu(ref, 0.1)
    if has_res:
        ref = ref + res
    cpad = (cout + 31) // 32 * 32
    xb = torch.empty(B, cin // 16, H, H, 16, device=dev, dtype=tdt)
    xd, wd = x.to(dev), w.to(dev)  # keep device operands alive across the C calls
    check(to_blocked(ptr(xd), ptr(xb), B, cin, H, H, st))
    packed = torch.empty(L.ay_packed_weight_bytes(cpad, cin, k), device=dev, dtype=torch.uint8)
    check(getattr(L, f"ay_pack_conv_weights_{dtype}")(ptr(wd), ptr(packed), cout, cpad, cin, k, st))
    sc = torch.zeros(cpad, device=dev)
    sh = torch.zeros(cpad, device=dev)
    sc[:cout], sh[:cout] = scale.to(dev), shift.to(dev)
    rb = None
    if has_res:
        rb = torch.empty(B, cpad // 16, Ho, Ho, 16, device=dev, dtype=tdt)
        rd = res.to(dev)
        check(to_blocked(ptr(rd), ptr(rb), B, cout, Ho, Ho, st))
    ob = torch.full((B, cpad // 16, Ho, Ho, 16), float("nan"), device=dev, dtype=torch.float32 if out_f32 else tdt)
    d = ConvDesc(B, cin, cout, H, H, Ho, Ho, k, stride, int(leaky), int(out_f32), cpad)
    check(getattr(L, f"ay_conv_fwd_{dtype}")(C.byref(d), ptr(xb), ptr(packed), ptr(sc), ptr(sh), ptr(rb), ptr(ob), st), "conv")
    got = torch.empty(B, cout, Ho, Ho, device=dev)
    fn = L.ay_blocked_f32_to_nchw_f32 if out_f32 else getattr(L, f"ay_blocked_{dtype}_to_nchw_f32")
    check(fn(ptr(ob), ptr(got), B, cout, Ho, Ho, st))
    got = got.cpu()
    assert torch.isfinite(got).all()
    if not out_f32:
        ref = _bf16r(ref)
    err = (got - ref).abs()
    bound = ref.abs() * 2.0 ** -7 + 1e-3 if dtype == "bf16" else ref.abs() * 2.0 ** -10 + 2e-4
    assert bool((err <= bound).all()), float((err - bound).max())
    if cpad > cout:  # padded channels come out as exact zeros (scale = shift = 0 there)
        full = torch.empty(B, cpad, Ho, Ho, device=dev)
        check(fn(ptr(ob), ptr(full), B, cpad, Ho, Ho, st))
        assert float(full[:, cout:].abs().max()) == 0.0


@pytest.mark.parametrize("case", [(128, 256, 3, 1, 40), (256, 128, 1, 1, 13), (128, 256, 3, 2, 26)], ids=lambda c: "x".join(str(v) for v in c))
def test_half_storage_saturates_instead_of_overflowing(dev, case):
    """precision="fp16": a stored value beyond the half range comes out as +-65504 (MODE.FP16_OVFL, set by every kernel that stores
    halves), not as +-inf that the next layer turns into NaN; true infinities and NaNs in the INPUT of a conversion pass through."""
    L = _lib.lib()
    st = _lib.stream_ptr()
    # the layout converter (fp32 NCHW -> blocked half)
    v = torch.tensor([1.0, 65504.0, 65520.0, 1e6, -3e38, float("inf"), float("-inf"), float("nan")] * 2).view(1, 16, 1, 1).to(dev)
    vb = torch.empty(1, 1, 1, 1, 16, device=dev, dtype=torch.float16)
    check(L.ay_nchw_f32_to_blocked_f16(ptr(v), ptr(vb), 1, 16, 1, 1, st))
    got = vb.view(-1)[:8].float().cpu()
    assert got[:5].tolist() == [1.0, 65504.0, 65504.0, 65504.0, -65504.0], got
    assert got[5] == float("inf") and got[6] == float("-inf") and bool(torch.isnan(got[7]))
    # a convolution block whose fp32 results lie far outside the half range (scale 1e6): m16 / 1x1 ring / stride-2 ring epilogues
    cin, cout, k, stride, H = case
    B = 2
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, cin, H, H, generator=g).to(torch.float16).float()
    w = torch.randn(cout, cin, k, k, generator=g) * (1.0 / np.sqrt(cin * k * k))
    pad = (k - 1) // 2
    Ho = (H + 2 * pad - k) // stride + 1
    ref = F.leaky_relu(F.conv2d(x, w.to(torch.float16).float(), None, stride, pad) * 1e6, 0.1).clamp(-65504.0, 65504.0)
    xb = torch.empty(B, cin // 16, H, H, 16, device=dev, dtype=torch.float16)
    xd, wd = x.to(dev), w.to(dev)
    check(L.ay_nchw_f32_to_blocked_f16(ptr(xd), ptr(xb), B, cin, H, H, st))
    packed = torch.empty(L.ay_packed_weight_bytes(cout, cin, k), device=dev, dtype=torch.uint8)
    check(L.ay_pack_conv_weights_f16(ptr(wd), ptr(packed), cout, cout, cin, k, st))
    sc = torch.full((cout,), 1e6, device=dev)
    sh = torch.zeros(cout, device=dev)
    ob = torch.empty(B, cout // 16, Ho, Ho, 16, device=dev, dtype=torch.float16)
    d = ConvDesc(B, cin, cout, H, H, Ho, Ho, k, stride, 1, 0, cout)
    check(L.ay_conv_fwd_f16(C.byref(d), ptr(xb), ptr(packed), ptr(sc), ptr(sh), None, ptr(ob), st), "conv")
    got = torch.empty(B, cout, Ho, Ho, device=dev)
    check(L.ay_blocked_f16_to_nchw_f32(ptr(ob), ptr(got), B, cout, Ho, Ho, st))
    got = got.cpu()
    assert torch.isfinite(got).all() and float(got.max()) == 65504.0 and float(got.min()) < -6e4
    sat = ref.abs() >= 65504.0
    assert bool(sat.any()) and torch.equal(got[sat], ref[sat])            # every overflowing value is the signed maximum
    inside = ref.abs() < 6e4
    assert bool(((got - ref).abs()[inside] <= ref.abs()[inside] * 2.0 ** -9 + 1.0).all())


F32_CASES = [
    # cin1, cin2, up1, cout, k, stride, H, leaky, has_res, B      (cin = cin1 + cin2)
    (3, 0, 0, 32, 3, 1, 70, True, False, 2),       # stem: 3 channels in a stage of 8 (masked filter tail), ragged tiles
    (32, 0, 0, 64, 3, 2, 70, True, False, 2),      # stride 2, even input
    (64, 0, 0, 128, 3, 2, 13, True, False, 3),     # stride 2, odd input (13 -> 7)
    (64, 0, 0, 32, 1, 1, 40, True, False, 2),      # 1x1, 32 of the tile's 64 output channels
    (32, 0, 0, 64, 3, 1, 33, True, True, 1),       # shortcut, one pixel into the second tile column / fifth tile row
    (128, 0, 0, 256, 3, 1, 13, True, True, 5),     # 4 channel groups, 13x13, residual, odd batch
    (256, 0, 0, 24, 1, 1, 13, False, False, 2),    # linear head with bias, 24 of 64 channels
    (128, 256, 1, 128, 1, 1, 26, True, False, 2),  # route [upsampled x2 | direct] folded into the loader
    (128, 0, 1, 64, 1, 1, 16, True, False, 2),     # a lazily upsampled single source
    (24, 40, 0, 72, 3, 1, 20, True, False, 1),     # route without upsampling, 72 output channels (two groups, the second ragged)
    (20, 0, 0, 16, 3, 1, 9, False, False, 1),      # cin % 8 != 0 in the LAST stage of a multi-stage loop
]


@pytest.mark.parametrize("case", F32_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv_f32_mfma_kernel(dev, case):
    """ay_conv_fwd_f32 -- the parity path's block on exact-fp32 MFMA (csrc/ay_conv_f32_mfma.hip) -- against torch-CPU fp32
    conv2d of the same operands with the reference's epilogue order (affine, LeakyReLU, + shortcut), route / upsample folded as
    models.py:86-96,244-245.  fp32 products and sums in another order: 2e-6 of the result's scale."""
    cin1, cin2, up1, cout, k, stride, H, leaky, has_res, B = case
    L = _lib.lib()
    st = _lib.stream_ptr()
    g = torch.Generator().manual_seed(cin1 * 3 + cin2 + cout + k + H)
    cin = cin1 + cin2
    h1 = H >> up1
    x1 = torch.randn(B, cin1, h1, h1, generator=g)
    x2 = torch.randn(B, cin2, H, H, generator=g) if cin2 else None
    w = torch.randn(cout, cin, k, k, generator=g) * (1.0 / np.sqrt(cin * k * k))
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    pad = (k - 1) // 2
    Ho = (H + 2 * pad - k) // stride + 1
    res = torch.randn(B, cout, Ho, Ho, generator=g) if has_res else None
    xin = x1.repeat_interleave(2, 2).repeat_interleave(2, 3) if up1 else x1
    if cin2:
        xin = torch.cat([xin, x2], 1)
    ref = F.conv2d(xin, w, None, stride, pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if leaky:
        ref = F.leaky_relu(ref, 0.1)
    if has_res:
        ref = ref + res
    d = ConvDesc(B, cin, cout, H, H, Ho, Ho, k, stride, int(leaky), 0, cout)
    x1d, x2d, wd, scd, shd = x1.to(dev), None if x2 is None else x2.to(dev), w.to(dev), scale.to(dev), shift.to(dev)
    rd = None if res is None else res.to(dev)
    out = torch.full((B, cout, Ho, Ho), float("nan"), device=dev)
    check(L.ay_conv_fwd_f32(C.byref(d), ptr(x1d), cin1, up1, ptr(x2d), ptr(wd), ptr(scd), ptr(shd), ptr(rd), ptr(out), st), "ay_conv_fwd_f32")
    got = out.cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    assert float(err.max()) <= 2e-6 * max(1.0, float(ref.abs().max())) * np.sqrt(cin * k * k / 16.0 + 1.0), float(err.max())


def test_stem_and_concat(dev):
    L = _lib.lib()
    st = _lib.stream_ptr()
    g = torch.Generator().manual_seed(5)
    B, H = 2, 70
    x = torch.rand(B, 3, H, H, generator=g)
    w = torch.randn(32, 3, 3, 3, generator=g) * 0.2
    scale, shift = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1
    ref = _bf16r(F.leaky_relu(F.conv2d(x, w, None, 1, 1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), 0.1))
    ob = torch.empty(B, 2, H, H, 16, device=dev, dtype=torch.bfloat16)
    xd, wd, scd, shd = x.to(dev), w.to(dev), scale.to(dev), shift.to(dev)
    check(L.ay_stem_conv_fwd(ptr(xd), ptr(wd), ptr(scd), ptr(shd), ptr(ob), B, H, H, 1, st))
    got = torch.empty(B, 32, H, H, device=dev)
    check(L.ay_blocked_bf16_to_nchw_f32(ptr(ob), ptr(got), B, 32, H, H, st))
    err = (got.cpu() - ref).abs()
    assert bool((err <= ref.abs() * 2.0 ** -7 + 1e-5).all())
    # route + upsample gather is a pure copy: bit-exact
    a = _bf16r(torch.randn(B, 32, 6, 6, generator=g))
    b_ = _bf16r(torch.randn(B, 48, 12, 12, generator=g))
    ab = torch.empty(B, 2, 6, 6, 16, device=dev, dtype=torch.bfloat16)
    bb = torch.empty(B, 3, 12, 12, 16, device=dev, dtype=torch.bfloat16)
    ad, bd = a.to(dev), b_.to(dev)
    check(L.ay_nchw_f32_to_blocked_bf16(ptr(ad), ptr(ab), B, 32, 6, 6, st))
    check(L.ay_nchw_f32_to_blocked_bf16(ptr(bd), ptr(bb), B, 48, 12, 12, st))
    ob = torch.empty(B, 5, 12, 12, 16, device=dev, dtype=torch.bfloat16)
    check(L.ay_concat_upsample_bf16(ptr(ab), 32, 1, ptr(bb), 48, ptr(ob), B, 12, 12, st))
    got = torch.empty(B, 80, 12, 12, device=dev)
    check(L.ay_blocked_bf16_to_nchw_f32(ptr(ob), ptr(got), B, 80, 12, 12, st))
    ref = torch.cat([F.interpolate(a, scale_factor=2, mode="nearest"), b_], 1)
    assert torch.equal(got.cpu(), ref)


@pytest.mark.parametrize("H", [72, 70, 136], ids=lambda h: f"s{h}")
def test_stem_fused_kernel(dev, H):
    """fused layer 0 + layer 1 vs torch-CPU with the same rounding points (bf16 image/filters, fp32 accumulate, bf16
    intermediate, bf16 output).  Ragged sizes (72 -> 36: not a multiple of the tile; 136: several tiles per row); 70 is not a
    multiple of 4 and takes the round-1 kernel (4-byte tile DMA), the others the pipelined v2 kernel (16-byte descriptor DMA)."""
    L = _lib.lib()
    st = _lib.stream_ptr()
    g = torch.Generator().manual_seed(11)
    B = 2
    x = torch.rand(B, 3, H, H, generator=g)
    w0 = torch.randn(32, 3, 3, 3, generator=g) * 0.3
    s0, t0 = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.2
    w1 = torch.randn(64, 32, 3, 3, generator=g) * (1.0 / np.sqrt(288))
    s1, t1 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    mid = _bf16r(F.leaky_relu(F.conv2d(_bf16r(x), _bf16r(w0), None, 1, 1) * s0.view(1, -1, 1, 1) + t0.view(1, -1, 1, 1), 0.1))
    ref = _bf16r(F.leaky_relu(F.conv2d(mid, _bf16r(w1), None, 2, 1) * s1.view(1, -1, 1, 1) + t1.view(1, -1, 1, 1), 0.1))
    w0p = torch.zeros(32, 32)
    w0p[:, :27] = w0.reshape(32, 27)
    xd, w0d, w1d = x.to(dev), w0p.to(torch.bfloat16).to(dev), w1.to(dev)
    s0d, t0d, s1d, t1d = s0.to(dev), t0.to(dev), s1.to(dev), t1.to(dev)
    packed = torch.empty(L.ay_packed_weight_bytes(64, 32, 3), device=dev, dtype=torch.uint8)
    check(L.ay_pack_conv_weights_bf16(ptr(w1d), ptr(packed), 64, 64, 32, 3, st))
    Ho = H // 2
    ob = torch.full((B, 4, Ho, Ho, 16), float("nan"), device=dev, dtype=torch.bfloat16)
    check(L.ay_stem_s2_fused_fwd(ptr(xd), ptr(w0d), ptr(s0d), ptr(t0d), 1, ptr(packed), ptr(s1d), ptr(t1d), 1, ptr(ob), B, H, H, st))
    got = torch.empty(B, 64, Ho, Ho, device=dev)
    check(L.ay_blocked_bf16_to_nchw_f32(ptr(ob), ptr(got), B, 64, Ho, Ho, st))
    got = got.cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    bound = ref.abs() * 2.0 ** -7 + 4e-3   # an intermediate that rounds the other way moves the sum by ~1e-3
    assert float((err > bound).float().mean()) <= 1e-3 and float(err.max()) <= 0.05, (float((err > bound).float().mean()), float(err.max()))


# ----------------------------------------------------------------------------------------- whole model
_models = {}


def build_models(C_, cfg_dir, dev, precision):
    key = (C_, precision)
    if key not in _models:
        cfg = cfg_gen.write_cfg(C_, cfg_dir)
        defs = parse_config.parse_model_config(cfg)
        params = synth.synth_params(defs, seed=7)
        wpath = os.path.join(cfg_dir, f"synth_c{C_}.weights")
        if not os.path.exists(wpath):
            synth.write_darknet_weights(wpath, defs, params, seen=12345)
        m = Darknet(cfg, precision=precision).to(dev).eval()
        m.load_darknet_weights(wpath)              # through the Darknet .weights format boundary
        o = OracleDarknet(cfg)
        o.set_params(params)
        _models[key] = (m, o)
    return _models[key]


@pytest.mark.parametrize("case", gc.MODEL_CASES, ids=lambda c: c[0])
def test_model_fp32_vs_reference_fixtures(golden_dir, tmp_cfg_dir, dev, case):
    """precision='fp32' HIP path == the reference's CPU path: boxes within 1e-4, NMS indices bit-exact."""
    name, C_, S, B, start = case
    z = load(golden_dir, "model_" + name)
    m, _ = build_models(C_, tmp_cfg_dir, dev, "fp32")
    m.keep_layer_outputs = True
    x = torch.from_numpy(gc.model_inputs(S, B, start))
    out = m(x)
    assert not out.is_cuda and out.shape == (B, m.num_boxes(S), 5 + C_)
    if "out" in z:
        close(out.numpy(), z["out"], TOL, "boxes")
    else:
        close(out.numpy()[:, z["out_rows"]], z["out_sel"], TOL, "boxes")
    for k, li in enumerate(z["layer_idx"]):
        li = int(li)
        if li not in m.layer_outputs:
            continue  # conv fused into the following shortcut: its own output is never materialised
        f = m.layer_output_nchw(li).cpu().numpy().reshape(-1)
        close(f[z["samp_idx"][k]], z["samp_val"][k], TOL, f"layer {li}")
    res = ay.non_max_suppression(out, 0.5, 0.4)
    for b in range(B):
        n = int(z[f"nms_n{b}"])
        assert (0 if res[b] is None else res[b].shape[0]) == n
        if n:
            np.testing.assert_array_equal(res.keep_idx[b], z[f"nms_keep{b}"])   # bit-exact box indices after NMS
            close(res[b].numpy(), z[f"nms_rows{b}"], TOL, "nms rows")
    m.keep_layer_outputs = False


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", [c for c in gc.MODEL_CASES if c[2] <= 416], ids=lambda c: c[0])
def test_model_bf16_vs_bf16_oracle(tmp_cfg_dir, dev, case, dtype):
    """The 16-bit MFMA paths (bfloat16 and IEEE-half storage) vs the oracle run with the same rounding points
    (mode='bf16' / 'fp16').  Per layer: at least 99.8 % of the stored activations within 2 ulps of the storage type (2^-6
    relative for bf16, 2^-9 for fp16) + 0.03 (fp16: 0.03 / 8) absolute of the oracle's.  (Accumulation order moves a sum across
    a rounding boundary now and then; on the residual stream such a 1-ulp difference rides the identity path through all
    later blocks of the stage, so a small fraction of elements sits 1-2 ulps off.)  Every bound of the half path is the
    bfloat16 bound divided by 8 = the ratio of the two rounding steps."""
    name, C_, S, B, start = case
    k = 1.0 if dtype == "bf16" else 0.125
    m, o = build_models(C_, tmp_cfg_dir, dev, dtype)
    m.keep_layer_outputs = True
    x = torch.from_numpy(gc.model_inputs(S, B, start))
    out = m(x).numpy()
    with torch.no_grad():
        ref = o.forward(x, mode=dtype, collect=True).numpy()
    worst = 0.0
    for li, t in sorted(m.layer_outputs.items()):
        if m._graph[li]["type"] not in ("convolutional", "shortcut", "route"):
            continue
        got = m.layer_output_nchw(li).cpu()
        want = o.layer_outputs[li]
        err = (got - want).abs()
        is_head = m._graph[li]["type"] == "convolutional" and not m._graph[li]["bn"]
        if is_head:
            # linear fp32 heads: the synthetic objectness filters carry a x4-x20 gain (synth.HEAD_CAL), which
            # multiplies the 1-2 ulp input differences; bound the logits loosely and the decoded boxes below
            assert float(err.max()) <= 0.5 * k and float((err > (want.abs() * 2.0 ** -6 + 0.05) * k).float().mean()) <= 5e-2, (li, float(err.max()))
            continue
        bound = (want.abs() * 2.0 ** -6 + 0.03) * k
        frac_bad = float((err > bound).float().mean())
        worst = max(worst, frac_bad)
        assert frac_bad <= 2e-3, (li, frac_bad, float(err.max()))
    # decoded boxes: conf/cls are sigmoids of those logits; coordinates relative to the box scale
    dconf = np.abs(out[..., 4:] - ref[..., 4:])
    assert np.quantile(dconf, 0.99) <= 2e-2 * k and dconf.max() <= 0.12 * k, (float(np.quantile(dconf, 0.99)), float(dconf.max()))
    box_scale = np.maximum(1.0, ref[..., 2:4].max(-1, keepdims=True))
    rel = np.abs(out[..., :4] - ref[..., :4]) / box_scale
    assert np.quantile(rel, 0.999) <= 5e-2 * k, float(np.quantile(rel, 0.999))
    m.keep_layer_outputs = False


@pytest.mark.parametrize("hw_size", [((64, 64), 64), ((37, 53), 64), ((53, 37), 96), ((600, 800), 416), ((100, 80), 416),
                                      ((1536, 1536), 1024), ((1, 7), 32)], ids=str)
def test_ingest_tiles_u8(hw_size):
    """ay_ingest_tiles_u8 == ToTensor + pad_to_square + nearest resize of the reference (ATen CPU ops), bit for bit."""
    from amyloid_yolo_paper_amd import datasets as D
    from oracle import ingest_oracle as io
    (h, w), size = hw_size
    rng = np.random.Generator(np.random.PCG64(h * 1000 + w))
    tiles = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
    got = D.ingest_tiles_device(tiles, size).cpu()
    for b in range(2):
        ref = io.ingest(tiles[b], size)
        assert torch.equal(got[b], ref), (hw_size, float((got[b] - ref).abs().max()))
    # host composition of this package (what ImageFolder does) agrees too
    host = D.resize(D.pad_to_square(D.to_tensor(tiles[0]))[0], size)
    assert torch.equal(got[0], host)


def test_detect_end_to_end(tmp_cfg_dir, tmp_path, dev):
    """detect(): image files of mixed sizes -> rescaled boxes, through the .weights boundary, with the GPU ingest and with
    the host transforms (identical results), and against the oracle pipeline (fp32: boxes within 1e-4, same kept rows)."""
    from PIL import Image
    from amyloid_yolo_paper_amd.detect import detect
    from oracle import ingest_oracle as io
    C_, S = 2, 128
    _, o = build_models(C_, tmp_cfg_dir, dev, "fp32")
    cfg = cfg_gen.write_cfg(C_, tmp_cfg_dir)
    wpath = os.path.join(tmp_cfg_dir, f"synth_c{C_}.weights")
    folder = tmp_path / "imgs"
    folder.mkdir()
    sizes = [(128, 128), (100, 160), (160, 100), (128, 128), (200, 200)]
    tiles = []
    for i, (h, w) in enumerate(sizes):
        t = np.ascontiguousarray(synth.synth_tile(40 + i, 256)[:h, :w])  # uint8 HWC
        Image.fromarray(t).save(str(folder / f"t{i}.png"))
        tiles.append(t)
    kw = dict(image_folder=str(folder), model_def=cfg, weights_path=wpath, conf_thres=0.5, nms_thres=0.4, batch_size=3, img_size=S,
              precision="fp32", verbose=False)
    paths, res_dev, _ = detect(device_ingest=True, **kw)
    paths2, res_host, _ = detect(device_ingest=False, **kw)
    assert paths == paths2 and len(paths) == len(sizes)
    n_boxes = 0
    for i, (a, b) in enumerate(zip(res_dev, res_host)):
        assert (a is None) == (b is None)
        x = io.ingest(tiles[i], S).unsqueeze(0)
        with torch.no_grad():
            ref = bo.non_max_suppression(o.forward(x).numpy().copy(), 0.5, 0.4)[0][0]
        assert (a is None) == (ref is None), i
        if a is None:
            continue
        assert torch.equal(a, b), i
        ref = bo.rescale_boxes(np.array(ref, np.float32), S, sizes[i])
        close(a.numpy(), ref, TOL, f"image {i}")
        n_boxes += a.shape[0]
    assert n_boxes > 0


@pytest.mark.parametrize("case", [(64, 40, 2), (128, 32, 2), (128, 13, 1), (64, 72, 1)], ids=str)
def test_resblock_fused_kernel(dev, case):
    """ay_resblock_fwd_bf16 (1x1 -> 3x3 -> +x in one kernel) against the two ay_conv_fwd_bf16 calls it replaces (same rounding
    points): BIT-identical where the 3x3 of the two-call path is the 32x32x16 kernel too (same K order: C = 64); for C = 128 the
    two-call 3x3 runs on the 16x16x32 kernel, whose fp32 sums are taken in another order, so single elements may land on the
    other side of a bf16 rounding boundary (at most one ulp, at most 1e-3 of the elements).  Both within 1 bf16 ulp (+1e-3)
    of the torch-CPU composition."""
    Cc, H, B = case
    CM = Cc // 2
    L = _lib.lib()
    st = _lib.stream_ptr()
    g = torch.Generator().manual_seed(Cc * 3 + H)
    x = _bf16r(torch.randn(B, Cc, H, H, generator=g))
    w1 = torch.randn(CM, Cc, 1, 1, generator=g) * (1.0 / np.sqrt(Cc))
    w2 = torch.randn(Cc, CM, 3, 3, generator=g) * (1.0 / np.sqrt(CM * 9))
    s1, t1 = torch.rand(CM, generator=g) + 0.5, torch.randn(CM, generator=g) * 0.1
    s2, t2 = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    mid = _bf16r(F.leaky_relu(F.conv2d(x, _bf16r(w1)) * s1.view(1, -1, 1, 1) + t1.view(1, -1, 1, 1), 0.1))
    ref = _bf16r(F.leaky_relu(F.conv2d(mid, _bf16r(w2), None, 1, 1) * s2.view(1, -1, 1, 1) + t2.view(1, -1, 1, 1), 0.1) + x)
    xd, w1d, w2d = x.to(dev), w1.to(dev), w2.to(dev)
    xb = torch.empty(B, Cc // 16, H, H, 16, device=dev, dtype=torch.bfloat16)
    check(L.ay_nchw_f32_to_blocked_bf16(ptr(xd), ptr(xb), B, Cc, H, H, st))
    p1 = torch.empty(L.ay_packed_weight_bytes(CM, Cc, 1), device=dev, dtype=torch.uint8)
    p2 = torch.empty(L.ay_packed_weight_bytes(Cc, CM, 3), device=dev, dtype=torch.uint8)
    check(L.ay_pack_conv_weights_bf16(ptr(w1d), ptr(p1), CM, CM, Cc, 1, st))
    check(L.ay_pack_conv_weights_bf16(ptr(w2d), ptr(p2), Cc, Cc, CM, 3, st))
    s1d, t1d, s2d, t2d = s1.to(dev), t1.to(dev), s2.to(dev), t2.to(dev)
    # two-call path
    mb = torch.empty(B, CM // 16, H, H, 16, device=dev, dtype=torch.bfloat16)
    o2 = torch.full((B, Cc // 16, H, H, 16), float("nan"), device=dev, dtype=torch.bfloat16)
    d1 = ConvDesc(B, Cc, CM, H, H, H, H, 1, 1, 1, 0, CM)
    d2 = ConvDesc(B, CM, Cc, H, H, H, H, 3, 1, 1, 0, Cc)
    check(L.ay_conv_fwd_bf16(C.byref(d1), ptr(xb), ptr(p1), ptr(s1d), ptr(t1d), None, ptr(mb), st), "conv1")
    check(L.ay_conv_fwd_bf16(C.byref(d2), ptr(mb), ptr(p2), ptr(s2d), ptr(t2d), ptr(xb), ptr(o2), st), "conv2")
    # fused
    assert L.ay_resblock_supported(Cc) == 1 and L.ay_resblock_supported(256) == 0
    of = torch.full((B, Cc // 16, H, H, 16), float("nan"), device=dev, dtype=torch.bfloat16)
    check(L.ay_resblock_fwd_bf16(ptr(xb), ptr(p1), ptr(s1d), ptr(t1d), 1, ptr(p2), ptr(s2d), ptr(t2d), 1, ptr(of), B, Cc, H, H, st),
          "resblock")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(of.float()).all())
    if Cc == 64:
        assert torch.equal(of.view(torch.int16), o2.view(torch.int16)), int((of.view(torch.int16) != o2.view(torch.int16)).sum())
    else:
        a_, b_ = of.float(), o2.float()
        diff = (a_ - b_).abs()
        assert bool((diff <= b_.abs() * 2.0 ** -7 + 1e-6).all()) and float((diff > 0).float().mean()) <= 1e-3, \
            (float(diff.max()), float((diff > 0).float().mean()))
    got = torch.empty(B, Cc, H, H, device=dev)
    check(L.ay_blocked_bf16_to_nchw_f32(ptr(of), ptr(got), B, Cc, H, H, st))
    err = (got.cpu() - ref).abs()
    bound = ref.abs() * 2.0 ** -6 + 2e-3   # two rounded layers: the intermediate may round the other way too
    assert float((err > bound).float().mean()) <= 1e-3, (float((err > bound).float().mean()), float(err.max()))


@pytest.mark.parametrize("thr", [0.5, 0.75])
def test_eval_statistics_device_vs_reference(golden_dir, dev, thr):
    """stats.get_batch_statistics (ay_match_detections) + ap_per_class against the reference's outputs: TP flags exact,
    precision / recall / AP / F1 to 1e-12."""
    from amyloid_yolo_paper_amd import stats
    z = load(golden_dir, "stats_cases")
    tag = f"t{int(thr * 100)}"
    outputs, targets = gc.stats_inputs()
    t_out = [None if o is None else torch.from_numpy(o) for o in outputs]
    metrics = stats.get_batch_statistics(t_out, torch.from_numpy(targets), thr)
    assert len(metrics) == int(z[f"{tag}_n"])
    for k, (tp, scores, labels) in enumerate(metrics):
        np.testing.assert_array_equal(tp, z[f"{tag}_tp{k}"])
        np.testing.assert_array_equal(scores.numpy(), z[f"{tag}_scores{k}"])
        np.testing.assert_array_equal(labels.numpy(), z[f"{tag}_labels{k}"])
    tp, scores, labels = [np.concatenate([np.asarray(v) for v in x], 0) for x in zip(*metrics)]
    p, r, ap, f1, cls = stats.ap_per_class(tp, scores, labels, targets[:, 1].tolist())
    for got, name in ((p, "p"), (r, "r"), (ap, "ap"), (f1, "f1")):
        np.testing.assert_allclose(got, z[f"{tag}_{name}"], rtol=1e-12, atol=0)
    np.testing.assert_array_equal(cls, z[f"{tag}_cls"])
    # against the oracle on a bigger random case (ties in IoU included: duplicated targets)
    rng = np.random.Generator(np.random.PCG64(3))
    B = 9
    tg, outs = [], []
    for b in range(B):
        nt = int(rng.integers(20, 120))
        xy = rng.uniform(0, 900, (nt, 2))
        tb = np.concatenate([xy, xy + rng.uniform(10, 90, (nt, 2))], 1)
        tb[nt // 2] = tb[0]                                       # exact duplicate target: equal IoUs, first index wins
        cls_ = rng.integers(0, 2, nt)
        tg += [[b, cls_[k], *tb[k]] for k in range(nt)]
        nd = int(rng.integers(50, 300))
        pick = rng.integers(0, nt, nd)
        rows = np.concatenate([tb[pick] + rng.normal(0, 6, (nd, 4)), rng.uniform(0.5, 1, (nd, 2)), cls_[pick][:, None]], 1)
        outs.append(rows.astype(np.float32))
    tg = np.asarray(tg, np.float32)
    ref = bo.get_batch_statistics(outs, tg, 0.5)
    got = stats.get_batch_statistics([torch.from_numpy(o) for o in outs], torch.from_numpy(tg), 0.5)
    for (a, _, _), (b_, _, _) in zip(got, ref):
        np.testing.assert_array_equal(a, b_)


def test_forward_is_deterministic_under_dynamic_dealing(tmp_cfg_dir, dev):
    """The ring kernel deals items to workgroups through atomic counters (which workgroup computes a tile varies from
    launch to launch); every tile is computed exactly once by the same arithmetic, so outputs repeat bit for bit, and
    stale-buffer reuse would show as a mismatch after the buffers are poisoned in between."""
    m, _ = build_models(3, tmp_cfg_dir, dev, "bf16")
    x = torch.from_numpy(gc.model_inputs(416, 3, 20))
    ref = m(x).clone()
    for rep in range(4):
        for t in m._act_bufs.get(("bf16", 3, 416), {}).values():
            if torch.is_tensor(t) and t.dtype == torch.bfloat16:
                t.fill_(float("nan"))                      # an item that is skipped leaves NaNs behind
        out = m(x)
        assert torch.equal(out, ref), rep
    assert bool(torch.isfinite(ref).all())


@pytest.mark.parametrize("case", [(256, 512, 256, 16), (128, 256, 128, 26), (64, 64, 128, 8)], ids=str)
def test_conv1x1_cat_kernel(dev, case):
    """ay_conv1x1_cat_fwd_bf16 (route [upsampled x2 | direct] folded into the 1x1 loader) is BIT-identical to
    ay_concat_upsample_bf16 followed by ay_conv_fwd_bf16 (same operands, same K order)."""
    c1, c2, cout, H = case
    L = _lib.lib()
    st = _lib.stream_ptr()
    B = 2
    g = torch.Generator().manual_seed(c1 + c2 + H)
    a_half = _bf16r(torch.randn(B, c1, H // 2, H // 2, generator=g))
    b_full = _bf16r(torch.randn(B, c2, H, H, generator=g))
    w = torch.randn(cout, c1 + c2, 1, 1, generator=g) * (1.0 / np.sqrt(c1 + c2))
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ad, bd, wd, sc, sh = a_half.to(dev), b_full.to(dev), w.to(dev), scale.to(dev), shift.to(dev)
    ab = torch.empty(B, c1 // 16, H // 2, H // 2, 16, device=dev, dtype=torch.bfloat16)
    bb = torch.empty(B, c2 // 16, H, H, 16, device=dev, dtype=torch.bfloat16)
    check(L.ay_nchw_f32_to_blocked_bf16(ptr(ad), ptr(ab), B, c1, H // 2, H // 2, st))
    check(L.ay_nchw_f32_to_blocked_bf16(ptr(bd), ptr(bb), B, c2, H, H, st))
    packed = torch.empty(L.ay_packed_weight_bytes(cout, c1 + c2, 1), device=dev, dtype=torch.uint8)
    check(L.ay_pack_conv_weights_bf16(ptr(wd), ptr(packed), cout, cout, c1 + c2, 1, st))
    cat = torch.empty(B, (c1 + c2) // 16, H, H, 16, device=dev, dtype=torch.bfloat16)
    check(L.ay_concat_upsample_bf16(ptr(ab), c1, 1, ptr(bb), c2, ptr(cat), B, H, H, st))
    d = ConvDesc(B, c1 + c2, cout, H, H, H, H, 1, 1, 1, 0, cout)
    o_ref = torch.full((B, cout // 16, H, H, 16), float("nan"), device=dev, dtype=torch.bfloat16)
    o_cat = torch.full_like(o_ref, float("nan"))
    check(L.ay_conv_fwd_bf16(C.byref(d), ptr(cat), ptr(packed), ptr(sc), ptr(sh), None, ptr(o_ref), st))
    check(L.ay_conv1x1_cat_fwd_bf16(C.byref(d), ptr(ab), c1, ptr(bb), ptr(packed), ptr(sc), ptr(sh), ptr(o_cat), st))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(o_cat.float()).all())
    assert torch.equal(o_cat.view(torch.int16), o_ref.view(torch.int16))
    # and against torch: upsample + cat + conv on the CPU
    ref = F.conv2d(torch.cat([F.interpolate(a_half, scale_factor=2, mode="nearest"), b_full], 1), _bf16r(w))
    ref = _bf16r(F.leaky_relu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), 0.1))
    got = torch.empty(B, cout, H, H, device=dev)
    check(L.ay_blocked_bf16_to_nchw_f32(ptr(o_cat), ptr(got), B, cout, H, H, st))
    err = (got.cpu() - ref).abs()
    assert bool((err <= ref.abs() * 2.0 ** -7 + 1e-3).all()), float(err.max())



@pytest.mark.parametrize("opts", [dict(), dict(fuse_blocks=False, fold_routes=False), dict(stem_mode="fp32"), dict(precision="fp16"),
                                  dict(precision="fp16", fuse_blocks=False, fold_routes=False, stem_mode="fp32")], ids=str)
def test_native_plan_equals_per_layer_walk(tmp_cfg_dir, dev, opts):
    """ay_plan_forward (graph lowered once, values in one arena with lifetime reuse, the network issued from C) gives the
    same rows, bit for bit, as the per-layer ctypes walk: same kernels, same arguments.  Repeats show that reusing the
    arena across batches and across differently shaped plans leaves nothing behind."""
    opts = dict(opts)
    precision = opts.pop("precision", "bf16")   # the half-precision plan issues the _f16 entry points
    m, _ = build_models(3, tmp_cfg_dir, dev, precision)
    saved = {k: getattr(m, k) for k in ("fuse_blocks", "fold_routes", "stem_mode", "use_plan")}
    try:
        for k, v in opts.items():
            setattr(m, k, v)
        for S, B, start in ((416, 3, 20), (128, 2, 5), (416, 3, 31)):
            x = torch.from_numpy(gc.model_inputs(S, B, start))
            m.use_plan = False
            ref = m(x).clone()
            m.use_plan = True
            for rep in range(2):
                out = m(x)
                assert torch.equal(out, ref), (S, rep)
        prep = m._prepare(dev)
        plan = m._plan(3, 416, prep, dev)
        per_layer = sum(t.numel() * t.element_size() for k, t in m._act_bufs[(precision, 3, 416)].items() if isinstance(k, int))
        assert plan.workspace.numel() < 0.3 * per_layer          # the per-layer walk keeps every layer output alive
        # timed variant: one positive duration per op, the 3x3 layers dominating
        L = _lib.lib()
        xd = torch.from_numpy(gc.model_inputs(416, 3, 20)).to(dev)
        out = torch.empty(3, m.num_boxes(416), 8, device=dev)
        ms = (C.c_float * len(plan.ops))()
        check(L.ay_plan_forward_timed(plan.handle, ptr(xd), ptr(plan.workspace), ptr(out), ms, _lib.stream_ptr()), "timed")
        assert all(v > 0 for v in ms) and len(ms) == len(plan.ops)
        m.use_plan = False
        assert torch.equal(out.cpu(), m(torch.from_numpy(gc.model_inputs(416, 3, 20))))
        # stream-ordered profiling of selected layers across several forwards
        m.use_plan = True
        convs = {getattr(o, "_layer", None) for o in plan.ops} - {None}
        pick = set(sorted(convs)[:5])
        m.plan_profile_begin(3, 416, pick)
        for _ in range(3):
            m.forward_device(torch.from_numpy(gc.model_inputs(416, 3, 20)))
        per_op, n_fwd = m.plan_profile_end(3, 416)
        assert n_fwd == 3 and len(per_op) == len(plan.ops)
        assert all((ms > 0) == (layer in pick) for layer, kind, ms in per_op)
    finally:
        for k, v in saved.items():
            setattr(m, k, v)


@pytest.mark.parametrize("case", [(70, 100, 32, 32, 1), (70, 100, 32, 24, 1), (133, 97, 32, 32, 2), (64, 64, 16, 40, 2)], ids=str)
def test_region_tiles_device(dev, case):
    """ay_ingest_region_tiles_u8 (§8f N4) == the oracle's padded dzsave grid + N1 chain, bit for bit (integer mean, one fp32
    division), for the whole raster as one region and streamed strip by strip through RegionTileStream."""
    from oracle.ingest_oracle import region_tiles
    from amyloid_yolo_paper_amd.wsi import RegionTileStream
    H, W, tile, S, shrink = case
    r = np.random.default_rng(H + W).integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    want, (ty, tx) = region_tiles(r, tile, S, shrink)
    L = _lib.lib()
    rd = torch.from_numpy(r).to(dev)
    out = torch.empty(ty * tx, 3, S, S, device=dev)
    check(L.ay_ingest_region_tiles_u8(ptr(rd), H, W, W * 3, shrink, tile, ty, tx, S, ptr(out), _lib.stream_ptr()), "region")
    assert torch.equal(out.cpu(), want)
    got, coords = [], []
    stream = RegionTileStream(r, tile, S, shrink)
    assert (stream.tiles_y, stream.tiles_x) == (ty, tx)
    for tiles, cs in stream:
        got.append(tiles.cpu())
        coords += cs
    assert coords == [(j, i) for j in range(ty) for i in range(tx)]
    assert torch.equal(torch.cat(got), want)


def test_detect_region_equals_per_tile_detection(tmp_cfg_dir, dev):
    """detect_region over a raster == model + non_max_suppression on each oracle-cut tile, boxes moved to slide coordinates."""
    from oracle.ingest_oracle import region_tiles
    from amyloid_yolo_paper_amd.wsi import detect_region
    from amyloid_yolo_paper_amd.utils import non_max_suppression
    m, _ = build_models(3, tmp_cfg_dir, dev, "bf16")
    S, tile = 128, 192
    tiles = (gc.model_inputs(S, 6, 40) * 255).astype(np.uint8).transpose(0, 2, 3, 1)         # six synthetic tiles
    big = np.kron(np.ones((1, 1, 1), np.uint8), np.concatenate([np.concatenate(list(tiles[:3]), 1), np.concatenate(list(tiles[3:]), 1)], 0))
    raster = np.repeat(np.repeat(big, 2, 0), 2, 1)[: 2 * S + 77, : 3 * 2 * S - 50]           # 256-px content, ragged edges
    res = detect_region(m, raster, tile=tile, img_size=S, conf_thres=0.5, nms_thres=0.4, batch_size=4)
    want, (ty, tx) = region_tiles(raster, tile, S)
    det = non_max_suppression(m(want), 0.5, 0.4)
    expect = {}
    for t, d in enumerate(det):
        if d is not None:
            d = d.clone()
            d[:, :4] *= tile / S
            d[:, [0, 2]] += (t % tx) * tile
            d[:, [1, 3]] += (t // tx) * tile
            expect[(t // tx, t % tx)] = d
    assert len(expect) > 0 and {(a, b_) for a, b_, _ in res} == set(expect)
    for a, b_, d in res:
        assert torch.equal(d, expect[(a, b_)])


def test_merge_detections_device(golden_dir, dev):
    """ay_merge_detections (SURVEY 8f N3 on the device: core.py:366-423 + 326-364, one wavefront per image) against
    (a) the reference's own outputs as sets of rows on the fixture cases whose result does not depend on the set's iteration
    order, and (b) bit for bit, rows AND order, the CPU restatement with the kernel's explicit pair order
    (oracle.merge_detections_ordered) on the fixtures and on batches of clustered boxes with long merge chains, exact
    duplicates, empty images and a full 1024-row image."""
    from amyloid_yolo_paper_amd.postprocess import merge_detections_device
    z = load(golden_dir, "merge_cases")
    cases = dict(gc.merge_inputs())
    rng = np.random.Generator(np.random.PCG64(123))
    for k, (n, ncl, spread) in enumerate([(300, 12, 60.0), (1024, 40, 25.0), (64, 3, 5.0), (0, 1, 1.0)]):
        centers = rng.uniform(50, 1450, (ncl, 2))
        xy = centers[rng.integers(0, ncl, n)] + rng.normal(0, spread, (n, 2))
        wh = rng.uniform(6, 90, (n, 2))
        det = np.concatenate([xy, xy + wh, rng.uniform(0.5, 1, (n, 2)), rng.integers(0, 3, (n, 1))], 1).astype(np.float32)
        if n >= 64:
            det[5] = det[2]            # exact duplicates collapse like rows of a Python set
            det[40] = det[2]
        cases[f"clusters{k}"] = det
    names = list(cases)
    M = 1024
    rows = torch.zeros(len(names), M, 7)
    count = torch.zeros(len(names), dtype=torch.int32)
    for b, name in enumerate(names):
        d = cases[name]
        rows[b, :len(d)] = torch.from_numpy(d)
        count[b] = len(d)
    out, out_count = merge_detections_device(rows.to(dev).contiguous(), count.to(dev))
    out, out_count = out.cpu().numpy(), out_count.cpu().numpy()
    for b, name in enumerate(names):
        want = bo.merge_detections_ordered(cases[name])
        got = out[b, :out_count[b]].astype(np.float64)
        assert got.shape == want.shape, (name, got.shape, want.shape)
        np.testing.assert_array_equal(got, want, err_msg=name)
        if name in z and name != "random40":
            assert set(map(tuple, got.tolist())) == set(map(tuple, z[name].reshape(-1, 7).tolist())), name
    assert out_count[names.index("clusters1")] < 1024 and out_count[names.index("clusters3")] == 0
    assert not out[names.index("clusters3")].any()      # rows past an image's count read as zeros, not as uninitialised memory
    # the list form detect(merge_boxes=True, merge_on_device=True) goes through: the same rows, None for an image without detections
    from amyloid_yolo_paper_amd.postprocess import merge_detections_batch_device
    dets = [None if len(cases[nm]) == 0 else torch.from_numpy(cases[nm]) for nm in names]
    got_list = merge_detections_batch_device(dets)
    for b, nm in enumerate(names):
        if len(cases[nm]) == 0:
            assert got_list[b] is None
        else:
            np.testing.assert_array_equal(got_list[b].numpy(), out[b, :out_count[b]], err_msg=nm)


def test_giou_closed_form_vectors_hip(golden_dir):
    """ay_box_iou / ay_box_iou_pairwise in GIoU mode against the exact rationals of the published definition
    (tests/golden/giou_kat.json; the reference has no GIoU)."""
    import json
    doc = json.load(open(os.path.join(golden_dir, "giou_kat.json")))
    b1 = torch.tensor([c["box1"] for c in doc["cases"]], dtype=torch.float32)
    b2 = torch.tensor([c["box2"] for c in doc["cases"]], dtype=torch.float32)
    want = np.array([c["giou"][0] / c["giou"][1] for c in doc["cases"]])
    np.testing.assert_allclose(ay.bbox_iou(b1, b2, giou=True).cpu().numpy(), want, rtol=0, atol=2e-7)
    pw = ay.bbox_iou_pairwise(b1, b2, giou=True).cpu().numpy()
    np.testing.assert_allclose(np.diag(pw), want, rtol=0, atol=2e-7)
    np.testing.assert_allclose(pw, ay.bbox_iou_pairwise(b2, b1, giou=True).cpu().numpy().T, rtol=0, atol=2e-7)


def test_fp16_is_an_inference_storage_type(tmp_cfg_dir, dev):
    """BASELINE.json configs[4] names an "fp16 MFMA path": `Darknet(precision="fp16")` runs the same plan and kernels on IEEE
    half storage (v_mfma_*_f16).  Inference only: a training call is refused with a clear error (bf16 keeps fp32's exponent
    range and is the training type), and bf16 stays the constructor default so existing scripts do not change behaviour."""
    assert Darknet(cfg_gen.write_cfg(3)).precision == "bf16"
    m, _ = build_models(3, tmp_cfg_dir, dev, "fp16")
    x = torch.from_numpy(gc.model_inputs(64, 2, 0))
    out = m(x)
    assert out.shape == (2, m.num_boxes(64), 8) and bool(torch.isfinite(out).all())
    assert all(t.dtype in (torch.float16, torch.float32) for k, t in m._act_bufs[("fp16", 2, 64)].items())
    m.train()
    with pytest.raises(_lib.AyError, match="inference storage type"):
        m(x, torch.tensor([[0, 1, 0.5, 0.5, 0.2, 0.2]]))
    with pytest.raises(_lib.AyError, match="inference storage type"):
        m.train_step_device(x, torch.tensor([[0, 1, 0.5, 0.5, 0.2, 0.2]]))
