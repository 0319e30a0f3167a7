"""Seeded synthetic inputs for tests, smoke and bench.

Nothing here comes from the reference tree: pretrained weights there are
Git-LFS stubs, so the measured configuration uses random weights of
the same architecture written in the Darknet ``.weights`` layout
(``models.py:257-336`` of the reference defines that layout; App. C.2).

* ``synth_tiles``   uint8 HWC histology-like tiles, ``PCG64(1000+i)``.
* ``synth_params``  conv N(0,.02), BN gamma N(1,.02), beta 0 (the reference's
  ``weights_init_normal``, ``utils/utils.py:27-33``), BN running statistics set
  by propagating second moments through the graph so an eval-mode forward stays
  O(1) instead of collapsing/exploding over 75 layers, and head objectness bias
  shifted so that ~0.5-1 % of boxes pass ``conf >= 0.5``.
* ``synth_targets`` ``[nT,6]`` rows (sample, class, cx, cy, w, h), unique cells.
"""
import math

import numpy as np

LEAKY_M2 = 0.5 * (1.0 + 0.01)          # E[leaky(z)^2], z ~ N(0,1)
LEAKY_MEAN = 0.9 / math.sqrt(2.0 * math.pi)  # E[leaky(z)]


# (seed, classes) -> {head layer: (objectness gain per anchor, objectness bias per anchor)};
# printed by ``python -m oracle.calibrate_heads`` (offline, CPU oracle, three 1024^2 tiles).
HEAD_CAL = {
    (7, 2): {81: ([4.6348, 5.233, 5.1379], [-1.1341, -4.2461, -0.5355]),
             93: ([9.3465, 9.6287, 12.2964], [-7.1228, -8.9625, -1.7906]),
             105: ([10.8881, 21.8673, 15.2098], [-7.9731, -2.2181, -5.7313])},
    (7, 3): {81: ([4.6348, 4.1302, 2.3477], [-1.1341, -2.444, -5.1912]),
             93: ([5.8969, 7.4149, 9.3489], [-3.7614, -4.6462, -2.1114]),
             105: ([7.0444, 7.0258, 3.9525], [-5.798, -0.1279, -8.4073])},
}


def conv_table(module_defs):
    """(layer index, cin, cout, k, stride, bn, leaky) for every convolutional block.

    ``module_defs`` is the parsed cfg *including* the leading [net] block.
    """
    net = module_defs[0]
    filters = [int(net["channels"])]
    table = []
    for i, d in enumerate(module_defs[1:]):
        t = d["type"]
        if t == "convolutional":
            cout = int(d["filters"])
            table.append(dict(index=i, cin=filters[-1], cout=cout, k=int(d["size"]), stride=int(d["stride"]),
                              bn=bool(int(d["batch_normalize"])), leaky=d["activation"] == "leaky"))
            f = cout
        elif t == "route":
            f = sum(filters[1:][int(x)] for x in d["layers"].split(","))
        elif t == "shortcut":
            f = filters[1:][int(d["from"])]
        else:  # upsample / yolo / maxpool keep the channel count
            f = filters[-1]
        filters.append(f)
    return table


def synth_params(module_defs, seed=7, conf_bias=-2.4, input_m2=0.7, input_mean=0.8, head_cal="auto"):
    """dict: layer index -> {'weight', ('bias') | ('gamma','beta','mean','var')} (float32).

    ``head_cal``: {head layer idx: (gain[A], bias[A])} applied to the objectness channel of each
    anchor (``"auto"`` looks HEAD_CAL up by (seed, classes); ``None`` leaves ``conf_bias``)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    net = module_defs[0]
    stats = []  # (mean, second moment) of every layer output, analytic estimate
    filters = [int(net["channels"])]
    cur = (input_mean, input_m2)
    params = {}
    yolo_classes = [int(d["classes"]) for d in module_defs[1:] if d["type"] == "yolo"]
    num_classes = yolo_classes[0] if yolo_classes else 0
    if head_cal == "auto":
        head_cal = HEAD_CAL.get((seed, num_classes))
    for i, d in enumerate(module_defs[1:]):
        t = d["type"]
        if t == "convolutional":
            cin, cout, k = filters[-1], int(d["filters"]), int(d["size"])
            fan_in = cin * k * k
            if int(d["batch_normalize"]):
                w = (rng.standard_normal((cout, cin, k, k), dtype=np.float32) * np.float32(0.02))
                gamma = (1.0 + 0.02 * rng.standard_normal(cout, dtype=np.float32)).astype(np.float32)
                var = np.full(cout, fan_in * 0.02 ** 2 * cur[1], dtype=np.float32)
                params[i] = dict(weight=w, gamma=gamma, beta=np.zeros(cout, np.float32),
                                 mean=np.zeros(cout, np.float32), var=var)
                cur = (LEAKY_MEAN, LEAKY_M2) if d["activation"] == "leaky" else (0.0, 1.0)
            else:
                std = 1.0 / math.sqrt(fan_in * cur[1])
                w = (rng.standard_normal((cout, cin, k, k), dtype=np.float32) * np.float32(std))
                b = np.zeros(cout, np.float32)
                if num_classes and cout % (5 + num_classes) == 0:
                    b[4::5 + num_classes] = conf_bias
                    if head_cal and i in head_cal:
                        gain, bias = head_cal[i]
                        for a, (g, bb) in enumerate(zip(gain, bias)):
                            w[a * (5 + num_classes) + 4] *= np.float32(g)
                            b[a * (5 + num_classes) + 4] = bb
                params[i] = dict(weight=w, bias=b)
                cur = (0.0, 1.0)
            f = cout
        elif t == "route":
            idx = [int(x) for x in d["layers"].split(",")]
            chans = [filters[1:][j] for j in idx]
            srcs = [stats[j] for j in idx]
            tot = float(sum(chans))
            cur = (sum(c * s[0] for c, s in zip(chans, srcs)) / tot, sum(c * s[1] for c, s in zip(chans, srcs)) / tot)
            f = int(tot)
        elif t == "shortcut":
            a, b = stats[-1], stats[int(d["from"])]
            cur = (a[0] + b[0], a[1] + b[1] + 2.0 * a[0] * b[0])
            f = filters[1:][int(d["from"])]
        else:
            f = filters[-1]
        stats.append(cur)
        filters.append(f)
    return params


def write_darknet_weights(path, module_defs, params, seen=0):
    """Darknet binary: int32[5] header ([3]=seen), then per conv: BN(beta,gamma,mean,var)|bias, W."""
    header = np.array([0, 0, 0, seen, 0], dtype=np.int32)
    with open(path, "wb") as fh:
        header.tofile(fh)
        for i, d in enumerate(module_defs[1:]):
            if d["type"] != "convolutional":
                continue
            p = params[i]
            if int(d["batch_normalize"]):
                for k in ("beta", "gamma", "mean", "var"):
                    np.ascontiguousarray(p[k], np.float32).tofile(fh)
            else:
                np.ascontiguousarray(p["bias"], np.float32).tofile(fh)
            np.ascontiguousarray(p["weight"], np.float32).tofile(fh)


def synth_tile(index, size=1024):
    """One uint8 [size,size,3] tile: bright noisy background + a few dark Gaussian blobs."""
    rng = np.random.Generator(np.random.PCG64(1000 + index))
    img = rng.integers(180, 256, size=(size, size, 3), dtype=np.int64).astype(np.float32)
    n_blobs = int(rng.integers(0, 13))
    ys = np.arange(size, dtype=np.float32)
    for _ in range(n_blobs):
        cy, cx = rng.uniform(0, size, 2)
        sigma = rng.uniform(8.0, 60.0) * size / 1024.0
        depth = rng.uniform(60.0, 170.0)
        tint = rng.uniform(0.6, 1.0, 3).astype(np.float32)
        gy = np.exp(-0.5 * ((ys - cy) / sigma) ** 2).astype(np.float32)
        gx = np.exp(-0.5 * ((ys - cx) / sigma) ** 2).astype(np.float32)
        img -= depth * gy[:, None, None] * gx[None, :, None] * tint[None, None, :]
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_tiles(count, size=1024, start=0):
    """``[count,3,size,size]`` float32 in [0,1] exactly as ToTensor() would give (uint8/255, CHW)."""
    out = np.empty((count, 3, size, size), np.float32)
    for i in range(count):
        out[i] = synth_tile(start + i, size).transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    return out


def synth_targets(batch, num_classes, seed=11, max_per_tile=13, min_per_tile=1, wh_range=(0.02, 0.15), grid=None):
    """[nT,6] float32 rows (sample_idx, class, cx, cy, w, h), normalised; cells unique per tile.

    ``grid``: finest grid size used for the uniqueness check (defaults to 128 = 1024/8).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    grid = grid or 128
    rows = []
    for b in range(batch):
        n = int(rng.integers(min_per_tile, max_per_tile + 1))
        used = set()
        while n > 0:
            cx, cy = rng.uniform(0.05, 0.95, 2)
            w, h = rng.uniform(wh_range[0], wh_range[1], 2)
            # unique cell on the *coarsest* grid too (grid/4), so no (b, anchor, cell) collides
            cell = (int(cx * (grid // 4)), int(cy * (grid // 4)))
            if cell in used:
                continue
            used.add(cell)
            rows.append((b, int(rng.integers(0, num_classes)), cx, cy, w, h))
            n -= 1
    return np.asarray(rows, dtype=np.float32).reshape(-1, 6)
