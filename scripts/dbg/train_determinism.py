"""Diagnosis of tests/test_gpu_train_bf16.py::test_bf16_train_step_tracks_fp32_step (red in GPUTEST_r01.json).

(i)  two bf16 forward+backward passes on the same inputs: are loss and gradients bit-identical?  which parameters differ?
(ii) the first-step loss / head gradients are written to an .npz so that a second process (AY_CANVAS=0) can be diffed
     against the default:  python scripts/dbg/train_determinism.py out.npz [ref.npz]
(iii) the 8-Adam-step sequence of the test, repeated: how far does the end loss move from run to run?
"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import torch

from amyloid_yolo_paper_amd import cfg_gen, parse_config, synth
from amyloid_yolo_paper_amd.models import Darknet

C_, S, B = 3, 256, 4
d = "/tmp/cfgd"
os.makedirs(d, exist_ok=True)
cfg = cfg_gen.write_cfg(C_, d)
defs = parse_config.parse_model_config(cfg)
wpath = os.path.join(d, "w.weights")
synth.write_darknet_weights(wpath, defs, synth.synth_params(defs, seed=7), seen=0)
x = torch.from_numpy(synth.synth_tiles(B, S, 10))
tg = torch.from_numpy(synth.synth_targets(B, C_, seed=21, max_per_tile=6, min_per_tile=3, wh_range=(0.05, 0.4), grid=S // 8))
prec = os.environ.get("PREC", "bf16")


def fresh():
    m = Darknet(cfg, precision=prec).to("cuda")
    m.load_darknet_weights(wpath)
    m.train()
    return m


def one_step(m):
    loss, _ = m(x, tg)
    loss.backward()
    g = {n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters()}
    for p in m.parameters():
        p.grad = None
    return float(loss.item()), g


m = fresh()
l1, g1 = one_step(m)
m = fresh()
l2, g2 = one_step(m)
print(f"[{prec}] loss run1 {l1!r} run2 {l2!r} identical={l1 == l2}")
ndiff = 0
for n in g1:
    a, b = g1[n], g2[n]
    if not torch.equal(a, b):
        ndiff += 1
        rel = float((a - b).norm() / (a.norm() + 1e-30))
        if ndiff <= 12 or rel > 1e-3:
            print(f"  differs: {n:44s} relL2 {rel:.3e} max|d| {float((a - b).abs().max()):.3e}")
print(f"[{prec}] parameters with run-to-run different gradients: {ndiff} of {len(g1)}")

out = sys.argv[1] if len(sys.argv) > 1 else None
keys = ["module_list.105.conv_105.weight", "module_list.93.conv_93.weight", "module_list.81.conv_81.weight",
        "module_list.104.conv_104.weight", "module_list.80.conv_80.weight", "module_list.73.conv_73.weight", "module_list.0.conv_0.weight",
        "module_list.1.conv_1.weight", "module_list.5.conv_5.weight", "module_list.62.conv_62.weight"]
if out:
    np.savez(out, loss=np.float64(l1), **{k.replace(".", "_"): g1[k].numpy() for k in keys})
if len(sys.argv) > 2:
    ref = np.load(sys.argv[2])
    print(f"vs {sys.argv[2]}: loss {l1!r} vs {float(ref['loss'])!r}")
    for k in keys:
        a, b = g1[k].numpy().ravel(), ref[k.replace(".", "_")].ravel()
        cos = float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        print(f"  {k:40s} cos {cos:.6f} relL2 {np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30):.3e}")

# (iii) the test's 8-step sequence, three times
for rep in range(int(os.environ.get("REPS", "3"))):
    m = fresh()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    ls = []
    for _ in range(8):
        loss, _ = m(x, tg)
        loss.backward()
        opt.step()
        opt.zero_grad()
        ls.append(float(loss.item()))
    print(f"[{prec}] 8 Adam steps rep {rep}: " + " ".join(f"{v:.2f}" for v in ls))
