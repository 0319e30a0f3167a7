"""§8f N4 oracle (oracle/ingest_oracle.region_tiles): the dzsave 'google' tile grid restated on the CPU."""
import numpy as np
import torch

from oracle.ingest_oracle import ingest, region_tiles


def test_region_tiles_are_padded_grid_crops():
    rng = np.random.default_rng(3)
    r = rng.integers(0, 256, size=(70, 100, 3), dtype=np.uint8)
    t, (ty, tx) = region_tiles(r, 32, 32)
    assert (ty, tx) == (3, 4) and t.shape == (12, 3, 32, 32)
    # interior tile: the crop itself / 255
    np.testing.assert_array_equal(t[5].numpy(), r[32:64, 32:64].transpose(2, 0, 1).astype(np.float32) / np.float32(255))
    # bottom-right tile: 6 valid rows, 4 valid columns, the rest background white
    last = t[11].numpy()
    np.testing.assert_array_equal(last[:, :6, :4], r[64:70, 96:100].transpose(2, 0, 1).astype(np.float32) / np.float32(255))
    assert (last[:, 6:, :] == 1.0).all() and (last[:, :, 4:] == 1.0).all()
    # the resize to the network size is the N1 chain on the padded tile
    t2, _ = region_tiles(r, 32, 24)
    pad = np.full((32, 32, 3), 255, np.uint8)
    pad[:6, :4] = r[64:70, 96:100]
    assert torch.equal(t2[11], ingest(pad, 24))


def test_region_tiles_halving_rounds_half_up():
    r = np.zeros((5, 6, 3), np.uint8)            # odd height: the last row is dropped (floor), as a 0.5 resize does
    r[0, 0], r[0, 1], r[1, 0], r[1, 1] = 1, 2, 2, 1   # mean 1.5 -> 2
    r[2:4, 2:4] = 255
    t, (ty, tx) = region_tiles(r, 4, 4, shrink=2)
    assert (ty, tx) == (1, 1)
    img = (t[0] * 255).round().numpy().astype(int)
    assert img[0, 0, 0] == 2 and img[0, 1, 1] == 255 and img[0, 0, 2] == 0
    assert (img[:, 2:, :] == 255).all() and (img[:, :, 3:] == 255).all()   # beyond the 2x3 halved image: background
