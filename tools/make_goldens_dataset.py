#!/usr/bin/env python
"""tests/golden/dataset.npz: the reference's dataset loaders - DataSampleNpz (data/dataset.py) and DataSampleNpz_Musicalion
(data/dataset_musicalion.py) - run on the synthetic song files of tests/dataset_fixture.py (build container only: imports the REAL
/root/reference/polyffusion modules; third-party imports they never reach on this path are replaced by empty modules).  Only arrays
are written: for every song and track selection the whole-song tensors and two single items."""
import os
import sys
import tempfile
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/polyffusion"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import dataset_fixture as fx  # noqa: E402


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs the reference mounted at /root/reference")
    work = tempfile.mkdtemp(prefix="pf_golden_")
    os.chdir(work)

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
    import dirs
    # the Musicalion class takes its directory from dirs.py (a path relative to the working directory): lay the files out there
    fx.write_all(os.path.join(work, "pop"), os.path.join(work, dirs.MUSICALION_DATA_DIR), os.path.join(work, "split"))
    from data.dataset import DataSampleNpz
    from data.dataset_musicalion import DataSampleNpz_Musicalion
    out = {}

    def dump(tag, song, has_chord=True):
        p2, pn, ch, pm = song.get_whole_song_data()[:4] if has_chord else (*song.get_whole_song_data()[:2], None, song.get_whole_song_data()[-1])
        out[f"{tag}_len"] = len(song)
        out[f"{tag}_prmat2c"], out[f"{tag}_prmat"] = p2.numpy(), pm.numpy()
        out[f"{tag}_pnotree"] = pn.numpy().astype(np.int16)
        if ch is not None:
            out[f"{tag}_chord"] = ch.numpy()
        for i in (0, len(song) - 1):
            item = song[i]
            out[f"{tag}_item{i}_prmat2c"], out[f"{tag}_item{i}_prmat"] = item[0], item[-1]
            out[f"{tag}_item{i}_pnotree"] = np.asarray(item[1]).astype(np.int16)
            if has_chord:
                out[f"{tag}_item{i}_chord"] = item[2]

    for fn, (ds, seed, kind) in fx.SONGS.items():
        if ds == "pop909":
            for tracks in ([(0, 1, 2), (0,), (2, 0), (1,)] if kind != "single" else [(0, 1, 2)]):
                dump(f"{fn[:-4]}_t{''.join(map(str, tracks))}", DataSampleNpz(fn, list(tracks), data_dir=os.path.join(work, "pop")))
        else:
            song = DataSampleNpz_Musicalion(fn)
            res = song.get_whole_song_data()
            out[f"{fn[:-4]}_ntensors"] = len(res)
            tag = fn[:-4]
            out[f"{tag}_len"] = len(song)
            out[f"{tag}_prmat2c"], out[f"{tag}_pnotree"], out[f"{tag}_prmat"] = res[0].numpy(), res[1].numpy().astype(np.int16), res[-1].numpy()
            for i in (0, len(song) - 1):
                item = song[i]
                out[f"{tag}_item{i}_prmat2c"], out[f"{tag}_item{i}_pnotree"], out[f"{tag}_item{i}_prmat"] = item[0], np.asarray(item[1]).astype(np.int16), item[-1]
                out[f"{tag}_item{i}_n"] = len(item)
    path = os.path.join(REPO, "tests", "golden", "dataset.npz")
    np.savez_compressed(path, **out)
    print("dataset.npz", os.path.getsize(path) // 1024, "KiB;", len(out), "arrays")


if __name__ == "__main__":
    main()
