#!/usr/bin/env python
"""tests/golden/datasample.npz: the reference's DataSample on seeded synthetic songs (build container only).
Imports the REAL /root/reference/polyffusion/data/datasample.py (its MIDI front end, which needs muspy / pretty_midi /
mir_eval, is replaced by an empty module: DataSample itself never calls into it); only arrays are written."""
import os
import sys
import tempfile
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/polyffusion"
sys.path.insert(0, REPO)
from polyffusion_amd import synth  # noqa: E402


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs the reference mounted at /root/reference")
    os.chdir(tempfile.mkdtemp(prefix="pf_golden_"))

    def stub(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m

    stub("pretty_midi")
    tv = stub("torchvision")
    tv.models = stub("torchvision.models")
    tv.transforms = stub("torchvision.transforms")
    stub("labml", monit=types.SimpleNamespace(iterate=lambda n, it: it, enum=lambda n, it: enumerate(it)))
    stub("omegaconf", OmegaConf=object)
    sys.path.insert(0, REF)
    import data as data_pkg  # noqa: F401  (namespace package of the reference)
    stub("data.midi_to_data", get_data_for_single_midi=None)
    from data.datasample import DataSample
    out = {}
    for name, seed, bars in (("a", 31, 20), ("b", 32, 9), ("c", 33, 40)):
        ds = DataSample(synth.song_data(seed, bars))
        p2, _pn, ch, pm = ds.get_whole_song_data()
        out[f"{name}_seed"], out[f"{name}_bars"] = seed, bars
        out[f"{name}_prmat2c"], out[f"{name}_chord"], out[f"{name}_prmat"] = p2.numpy(), ch.numpy(), pm.numpy()
        out[f"{name}_pnotree"] = _pn.numpy().astype(np.int16)          # [S, 128, 20, 6] piano-tree grid (values < 131)
        s0 = ds[1]
        out[f"{name}_item1_prmat2c"], out[f"{name}_item1_chord"], out[f"{name}_item1_prmat"] = s0[0], s0[2], s0[3]
    path = os.path.join(REPO, "tests", "golden", "datasample.npz")
    np.savez_compressed(path, **out)
    print("datasample.npz", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
