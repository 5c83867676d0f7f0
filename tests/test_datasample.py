"""Input side (SURVEY.md 8f, f4 subset): polyffusion_amd.datasample against the vectors the REAL reference's DataSample
produced on the same seeded songs (tests/golden/datasample.npz, tools/make_goldens_datasample.py).  Integer / 0-1 data:
bit-exact."""
import os

import numpy as np
import pytest
import torch

from polyffusion_amd import datasample, synth

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "datasample.npz"))


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_whole_song_matches_reference(name):
    ds = datasample.DataSample(synth.song_data(int(G[f"{name}_seed"]), int(G[f"{name}_bars"])))
    p2, pn, ch, pm = ds.get_whole_song_data()
    assert pn.dtype == __import__("torch").int64 and np.array_equal(pn.numpy(), G[f"{name}_pnotree"].astype(np.int64))   # piano-tree grid [S,128,20,6]
    for got, key in ((p2, "prmat2c"), (ch, "chord"), (pm, "prmat")):
        want = G[f"{name}_{key}"]
        assert got.numpy().dtype == want.dtype and np.array_equal(got.numpy(), want), key
    s1 = ds[1]
    assert np.array_equal(s1[0], G[f"{name}_item1_prmat2c"]) and np.array_equal(s1[2], G[f"{name}_item1_chord"])
    assert np.array_equal(s1[3], G[f"{name}_item1_prmat"])


def test_npz_round_trip_and_edges(tmp_path):
    d = synth.song_data(41, 12)
    path = str(tmp_path / "song.npz")
    np.savez(path, **d)
    a, b = datasample.DataSample(d).get_whole_song_data(), datasample.DataSample.from_npz(path).get_whole_song_data()
    assert all(np.array_equal(x.numpy(), y.numpy()) for x, y in ((a[0], b[0]), (a[2], b[2]), (a[3], b[3])))
    # a note that outlasts the segment is clipped, an onset beyond it ignored, a later row overwrites the duration
    nmat = np.array([[126, 60, 10], [200, 61, 4], [5, 62, 3], [5, 62, 7]])
    p2 = datasample.nmat_to_prmat2c(nmat, 128)
    assert p2[0, 126, 60] == 1 and p2[1, 127, 60] == 1 and p2[:, :, 61].sum() == 0
    assert datasample.nmat_to_prmat(nmat, 128)[5, 62] == 7
    # empty song: no usable downbeat -> empty tensors, not an exception
    e = dict(d)
    e["db_pos_filter"] = np.zeros_like(d["db_pos_filter"])
    assert len(datasample.DataSample(e)) == 0


def _pop909_file(tmp_path, seed=0):
    """A synthetic song in the POP909 .npz layout ref:data/dataset.py:70-84 reads: per-track note matrices and start tables."""
    import os, pickle
    rng = np.random.default_rng(seed)

    def track(n):
        on = np.sort(rng.integers(0, 512, n))
        nm = np.stack([on, rng.integers(30, 90, n), rng.integers(1, 8, n), np.full(n, 80), np.zeros(n, int)], 1)
        return nm, {b: int(np.searchsorted(on, b)) for b in range(0, 513)}

    ts = [track(60), track(30), track(90)]
    notes, st = np.empty(3, dtype=object), np.empty(3, dtype=object)
    for i, (a, b) in enumerate(ts):
        notes[i], st[i] = a, b
    db = np.arange(0, 512, 16)
    filt = np.ones(len(db), bool)
    filt[-8:] = False
    chord = np.zeros((128, 14), int)
    chord[:, 0], chord[:, 13] = rng.integers(0, 12, 128), rng.integers(0, 12, 128)
    os.makedirs(tmp_path / "data")
    np.savez(tmp_path / "data" / "7.npz", notes=notes, start_table=st, db_pos=db, db_pos_filter=filt, chord=chord)
    with open(tmp_path / "pop909.pickle", "wb") as f:
        pickle.dump((["1.npz"], ["7.npz"]), f)
    return ts, db, filt, chord


def test_pop909_song_file_tracks(tmp_path):
    """DataSampleNpz (ref:data/dataset.py:27-253; pinned against the reference loader further down): one track alone equals DataSample on
    that track's rows, several tracks give the union of their piano rolls, and the validation half of the split pickle is what
    --from_dataset indexes (ref:inference_sdf.py:95-105)."""
    from polyffusion_amd import datasample as ds
    ts, db, filt, chord = _pop909_file(tmp_path)
    per_track = []
    for i, (nm, table) in enumerate(ts):
        one, fn = ds.choose_song_from_val_dl("pop909", 0, (i,), str(tmp_path / "data"), str(tmp_path))
        assert fn == "7.npz"
        ref = ds.DataSample(dict(notes=nm, start_table=table, db_pos=db, db_pos_filter=filt, chord=chord)).get_whole_song_data()
        got = one.get_whole_song_data()
        assert all(torch.equal(a, b) for a, b in zip(got, ref))
        per_track.append(got[0])
    allt = ds.DataSampleNpz("7.npz", (0, 1, 2), str(tmp_path / "data")).get_whole_song_data()
    assert torch.equal(allt[0], torch.stack(per_track).amax(0)) and allt[2].shape == (3, 32, 36)
    asked = []
    s, _ = ds.choose_song_from_val_dl("pop909", None, (0,), str(tmp_path / "data"), str(tmp_path), ask=lambda q: asked.append(q) or "0")
    assert asked and len(s) == int(filt.sum())
    with pytest.raises(NotImplementedError):
        ds.choose_song_from_val_dl("lakh", 0, split_dir=str(tmp_path))


def test_split_pickle_admits_only_name_lists(tmp_path):
    import os, pickle
    from polyffusion_amd import datasample as ds
    with open(tmp_path / "bad.pickle", "wb") as f:
        pickle.dump((["a.npz"], [os.path.join]), f)           # a global inside: refused, never resolved
    with pytest.raises(pickle.UnpicklingError):
        ds.load_split(str(tmp_path / "bad.pickle"))
    with open(tmp_path / "odd.pickle", "wb") as f:
        pickle.dump({"train": []}, f)
    with pytest.raises(ValueError):
        ds.load_split(str(tmp_path / "odd.pickle"))
    # the reference's own split files load (they are lists of names): checked where the reference checkout exists
    ref = "/root/reference/data/train_split_pnt/pop909.pickle"
    if os.path.exists(ref):
        tr, va = ds.load_split(ref)
        assert len(tr) == 797 and len(va) == 89 and va[0] == "258.npz"


# ---------------------------------------------------------------------------------------------- dataset loaders, pinned
def _dataset_files(tmp_path):
    import dataset_fixture as fx
    fx.write_all(str(tmp_path / "pop"), str(tmp_path / "mus"), str(tmp_path / "split"))
    return fx


DG = np.load(os.path.join(os.path.dirname(__file__), "golden", "dataset.npz"))


def _same(got, want):
    got = got.numpy() if hasattr(got, "numpy") else np.asarray(got)
    return got.shape == want.shape and np.array_equal(got.astype(np.int64) if want.dtype == np.int16 else got, want.astype(np.int64) if want.dtype == np.int16 else want)


def test_pop909_loader_against_the_reference_loader_on_the_same_files(tmp_path):
    """tests/golden/dataset.npz holds what the REAL ref:data/dataset.py DataSampleNpz returned for the synthetic song files of
    tests/dataset_fixture.py (tools/make_goldens_dataset.py): three tracks in four selections (the rows of the chosen tracks one after
    the other, in the order given), a song whose start tables end inside the last segment (`notes[s_ind:]`), whose chords run out
    (zero rows appended), with a silent stretch (an empty segment) and an irregular downbeat grid, and a single-matrix file.  Bit-exact."""
    from polyffusion_amd import datasample as ds
    fx = _dataset_files(tmp_path)
    n = 0
    for fn, (kind_ds, seed, kind) in fx.SONGS.items():
        if kind_ds != "pop909":
            continue
        for tracks in ([(0, 1, 2), (0,), (2, 0), (1,)] if kind != "single" else [(0, 1, 2)]):
            tag = f"{fn[:-4]}_t{''.join(map(str, tracks))}"
            song = ds.DataSampleNpz(fn, tracks, str(tmp_path / "pop"))
            assert len(song) == int(DG[f"{tag}_len"])
            p2, pn, ch, pm = song.get_whole_song_data()
            assert p2.dtype == torch.float32 and pn.dtype == torch.int64 and ch.dtype == torch.float32 and pm.dtype == torch.float32
            assert _same(p2, DG[f"{tag}_prmat2c"]) and _same(pn, DG[f"{tag}_pnotree"]) and _same(ch, DG[f"{tag}_chord"]) and _same(pm, DG[f"{tag}_prmat"]), tag
            for i in (0, len(song) - 1):
                it = song[i]
                assert _same(it[0], DG[f"{tag}_item{i}_prmat2c"]) and _same(it[1], DG[f"{tag}_item{i}_pnotree"]), (tag, i)
                assert _same(it[2], DG[f"{tag}_item{i}_chord"]) and _same(it[3], DG[f"{tag}_item{i}_prmat"]), (tag, i)
            n += 1
    assert n == 9
    # --from_dataset indexes the validation half of the split (ref:inference_sdf.py:95-105)
    song, fn = ds.choose_song_from_val_dl("pop909", 1, (0, 1, 2), str(tmp_path / "pop"), str(tmp_path / "split"))
    assert fn == "pop_ragged.npz" and _same(song.get_whole_song_data()[0], DG["pop_ragged_t012_prmat2c"])


def test_musicalion_loader_against_the_reference_loader_on_the_same_file(tmp_path):
    """ref:data/dataset_musicalion.py DataSampleNpz_Musicalion on the synthetic file (start table ending inside the last segment, one
    filtered downbeat): same piano rolls and piano-tree grid, bit for bit; no chord track."""
    from polyffusion_amd import datasample as ds
    _dataset_files(tmp_path)
    song, fn = ds.choose_song_from_val_dl("musicalion", 0, data_dir=str(tmp_path / "mus"), split_dir=str(tmp_path / "split"))
    assert fn == "mus_a.npz" and len(song) == int(DG["mus_a_len"])
    res = song.get_whole_song_data()
    assert res[2] is None
    assert _same(res[0], DG["mus_a_prmat2c"]) and _same(res[1], DG["mus_a_pnotree"]) and _same(res[3], DG["mus_a_prmat"])
    for i in (0, len(song) - 1):
        it = song[i]
        assert _same(it[0], DG[f"mus_a_item{i}_prmat2c"]) and _same(it[1], DG[f"mus_a_item{i}_pnotree"]) and _same(it[3], DG[f"mus_a_item{i}_prmat"])
