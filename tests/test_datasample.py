"""Input side (SURVEY.md 8f, f4 subset): polyffusion_amd.datasample against the vectors the REAL reference's DataSample
produced on the same seeded songs (tests/golden/datasample.npz, tools/make_goldens_datasample.py).  Integer / 0-1 data:
bit-exact."""
import os

import numpy as np
import pytest

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
