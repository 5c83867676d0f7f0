#!/usr/bin/env python
"""tests/golden/pnotree.npz: the sdf_pnotree conditioning side on the REAL reference (build container only; needs /root/reference).

* ``utils.nmat_to_pianotree_repr`` (utils.py:132-171) on seeded note matrices, incl. a step with more notes than slots and
  durations beyond the 32-step cap;
* ``dl_modules.PianoTreeEncoder`` (pianotree_enc.py) with seeded synthetic weights on the resulting grids, combined as
  ``Polyffusion_SDF._encode_pnotree`` does (models/model_sdf.py:138-151): four 2-bar segments -> [B, 1, 2048].
Arrays only."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from polyffusion_amd.weights import synth_pianotree_encoder_state  # noqa: E402
from tools.make_goldens import OUT, import_reference, save  # noqa: E402


def seeded_nmat(seed, n_notes, n_step=128, crowded_step=None):
    rng = np.random.Generator(np.random.PCG64(seed))
    o = rng.integers(0, n_step, n_notes)
    if crowded_step is not None:
        o[:25] = crowded_step                   # 25 notes on one step: more than the 18 free slots
    p = rng.integers(30, 100, n_notes)
    d = rng.integers(1, 40, n_notes)            # some beyond the cap of 32
    order = np.lexsort((d, p, o))
    return np.stack([o, p, d], 1)[order].astype(np.int64)


@torch.no_grad()
def main():
    import_reference()
    import utils as ref_utils
    from dl_modules import PianoTreeEncoder
    g = {}
    nmats = [seeded_nmat(51, 300), seeded_nmat(52, 120, crowded_step=7), seeded_nmat(53, 0), seeded_nmat(54, 600)]
    grids = []
    for i, nm in enumerate(nmats):
        g[f"nmat{i}"] = nm
        grids.append(ref_utils.nmat_to_pianotree_repr(nm, n_step=128))
        g[f"grid{i}"] = grids[-1]
    enc = PianoTreeEncoder().eval()
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in synth_pianotree_encoder_state(0).items()})
    pnotree = torch.from_numpy(np.stack(grids))                      # [4, 128, 20, 6]
    zs = [enc(seg)[0].mean for seg in pnotree.split(32, 1)]          # model_sdf.py:142-147
    g["z"] = torch.cat(zs, dim=-1).unsqueeze(1).numpy()              # [4, 1, 2048]
    dist, emb, lengths = enc(pnotree[:, :32])
    g["lengths_seg0"] = lengths.numpy()
    g["emb_seg0_sample"] = emb[0, :2].numpy()
    os.makedirs(OUT, exist_ok=True)
    save("pnotree.npz", **g)
    print("z", g["z"].shape, float(np.abs(g["z"]).max()))


if __name__ == "__main__":
    main()
