"""Offline recipe for the objectness calibration constants in amyloid_yolo_paper_amd/synth.py
(HEAD_CAL).  TEST INFRASTRUCTURE: runs the CPU oracle on three synthetic 1024x1024 tiles with
seed-7 weights and prints, per (head, anchor), the gain that gives the objectness logit unit
spatial spread and the bias that puts its 99.2 % quantile at 0 (=> ~0.8 % of boxes pass
conf >= 0.5, the regime SURVEY.md §8d asks the synthetic weights to be in).

    python -m oracle.calibrate_heads
"""
import numpy as np
import torch

from amyloid_yolo_paper_amd import cfg_gen, parse_config, synth
from oracle.darknet_oracle import OracleDarknet


def main():
    for C in (2, 3):
        cfg = cfg_gen.write_cfg(C, "/tmp/ay_golden")
        defs = parse_config.parse_model_config(cfg)
        params = synth.synth_params(defs, seed=7, head_cal=None, conf_bias=0.0)
        m = OracleDarknet(cfg)
        m.set_params(params)
        heads = [i - 1 for i, d in enumerate(defs[1:]) if d["type"] == "yolo"]
        logits = {h: [] for h in heads}
        for tile in (6, 0, 1):
            x = torch.from_numpy(synth.synth_tiles(1, 1024, tile))
            with torch.no_grad():
                m.forward(x, collect=True)
            for h in heads:
                t = m.layer_outputs[h]
                G = t.shape[2]
                logits[h].append(t.view(1, 3, 5 + C, G, G)[0, :, 4].reshape(3, -1).numpy())
        table = {}
        for h in heads:
            l = np.concatenate(logits[h], 1)
            gain = 1.0 / l.std(1)
            q = np.quantile(l * gain[:, None], 0.992, axis=1)
            table[h] = ([round(float(g), 4) for g in gain], [round(float(-v), 4) for v in q])
        print(f"    {C}: {table},")


if __name__ == "__main__":
    main()
