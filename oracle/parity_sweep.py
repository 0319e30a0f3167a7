"""TEST INFRASTRUCTURE ONLY (CPU oracle experiment; VERDICT r3 "do this" #2a): per-layer-range storage sweep.

Which layers of the 16-bit MFMA path need the 11-bit significand of IEEE half for the NMS indices to match the reference's?
DESIGN.md section 2 only tried whole-network mixes (activations / filters in one type each).  This runs the CPU oracle on bench.py's
own 16 synthetic 1024x1024 tiles with the storage type chosen per layer index (`OracleDarknet.forward(layer_modes=...)`: filters of
the layer and its stored output rounded to that type, fp32 accumulate, heads stay fp32) and reports `keep_match` etc. against the fp32
oracle (= the reference's CPU path on tests/golden) with oracle/parity.py -- the same metric bench.py prints as `parity`.

    python oracle/parity_sweep.py [--tiles 16] [--out profiles/r04_parity_layer_sweep.txt]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from amyloid_yolo_paper_amd import cfg_gen, parse_config, synth  # noqa: E402
from oracle import boxes_oracle as bo, parity  # noqa: E402
from oracle.darknet_oracle import OracleDarknet  # noqa: E402

# backbone stage boundaries: 0 stem | 1-4 | 5-11 | 12-36 | 37-61 | 62-74 | neck+heads 75-105
SPLITS = [
    ("all bf16", lambda i: "bf16"),
    ("all fp16", lambda i: "fp16"),
    ("bf16 backbone 0-74, fp16 neck+heads 75-105", lambda i: "bf16" if i <= 74 else "fp16"),
    ("fp16 backbone 0-74, bf16 neck+heads 75-105", lambda i: "fp16" if i <= 74 else "bf16"),
    ("bf16 layers 0-11 (stem + 512^2/256^2 stages: the HBM-bound ones), fp16 12-105", lambda i: "bf16" if i <= 11 else "fp16"),
    ("bf16 layers 0-36, fp16 37-105", lambda i: "bf16" if i <= 36 else "fp16"),
    ("bf16 layers 0-61, fp16 62-105", lambda i: "bf16" if i <= 61 else "fp16"),
    ("fp16 everywhere except the 1x1 layers (bf16)", None),   # filled in below: needs the layer table
    ("fp16 everywhere except the 3x3 stride-1 layers (bf16: the MFMA-bound family)", None),
    ("fp16 layers 0-11, bf16 12-105", lambda i: "fp16" if i <= 11 else "bf16"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=16)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(int(os.environ.get("AY_CPU_THREADS", str(os.cpu_count() or 1))))
    cfg = cfg_gen.write_cfg(a.classes)
    defs = parse_config.parse_model_config(cfg)
    m = OracleDarknet(cfg)
    m.set_params(synth.synth_params(defs, seed=7))
    kinds = {}
    for i, d in enumerate(m.defs):
        if d["type"] == "convolutional":
            kinds[i] = (int(d["size"]), int(d["stride"]))
        elif d["type"] == "shortcut":
            kinds[i] = kinds.get(i - 1, (0, 0))   # the sum is stored by the 3x3's epilogue
    splits = list(SPLITS)
    splits[7] = (splits[7][0], lambda i: "bf16" if kinds.get(i) == (1, 1) else "fp16")
    splits[8] = (splits[8][0], lambda i: "bf16" if kinds.get(i) == (3, 1) and i > 0 else "fp16")
    tiles = torch.from_numpy(synth.synth_tiles(a.tiles, a.size, start=0))
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    emit(f"# per-layer-range storage sweep, CPU oracle, {a.tiles} bench tiles {a.size}^2, C={a.classes}, conf 0.5 nms 0.4 (oracle/parity_sweep.py)")
    ref = []
    t0 = time.time()
    with torch.no_grad():
        for i in range(a.tiles):
            out = m.forward(tiles[i:i + 1]).numpy()
            rows, keep, _ = bo.non_max_suppression(out, 0.5, 0.4)
            ref.append((keep[0], rows[0]))
        emit(f"# fp32 reference: {sum(len(k) for k, _ in ref)} kept indices ({time.time() - t0:.0f} s)")
        emit(f"{'storage plan':<86} keep_match extra_heads max_box_rel max_dconf")
        for name, fn in splits:
            items = []
            for i in range(a.tiles):
                out = m.forward(tiles[i:i + 1], layer_modes=fn).numpy()
                rows, keep, _ = bo.non_max_suppression(out, 0.5, 0.4)
                items.append(parity.detection_agreement(ref[i][0], ref[i][1], keep[0], rows[0]))
            s = parity.summarize(items)
            emit(f"{name:<86} {s['keep_match']:>10.4f} {s['extra_heads']:>11.4f} {s['max_box_rel']:>11.4f} {s['max_dconf']:>9.5f}")
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
