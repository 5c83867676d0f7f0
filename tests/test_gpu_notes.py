"""Output step on the GPU (-m gpu): pf_prmat2c_durations and the host mirror of the reference's prmat2c_to_prmat /
prmat2c_to_midi_file against the oracle (bit-exact: integer durations, exact binary-fraction note times)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import notes_ref  # noqa: E402
from polyffusion_amd import _lib, midi, synth  # noqa: E402

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "notes.npz"))


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    _lib.require_gpu()


@pytest.mark.parametrize("n,steps,seed", [(3, 128, 11), (2, 64, 12), (1, 32, 13), (16, 128, 21), (1, 8, 22)])
@pytest.mark.parametrize("custom", [False, True])
def test_durations_bit_exact(n, steps, seed, custom):
    x = synth.prmat2c_image(seed, n, steps)
    got = midi.durations(torch.from_numpy(x).cuda(), is_custom_round=custom).cpu().numpy()
    assert np.array_equal(got, notes_ref.durations(x, custom))


def test_edges_empty_full_and_thresholds():
    z = np.zeros((2, 2, 128, 128), dtype=np.float32)
    assert midi.durations(z).sum().item() == 0                                   # empty image: no notes
    o = np.ones_like(z)
    d = midi.durations(o).cpu().numpy()                                          # everything on: every cell starts a note
    assert np.array_equal(d, np.broadcast_to(np.arange(128, 0, -1, dtype=np.int32)[None, :, None], d.shape))   # ... to the end
    t = np.zeros((1, 2, 4, 128), dtype=np.float32)
    t[0, 0, 0, :4] = [0.5, np.nextafter(np.float32(0.5), np.float32(1)), 1.5, -0.7]   # round-half-even: 0.5 -> 0
    t[0, 1, 1, :4] = 0.5                                                         # sustain exactly 0.5 does not continue
    d = midi.durations(t).cpu().numpy()
    assert d[0, 0, :4].tolist() == [0, 1, 1, 0]


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_prmat2c_to_prmat_matches_reference_golden(name):
    x = synth.prmat2c_image(int(G[f"{name}_seed"]), *[int(v) for v in G[f"{name}_shape"][[0, 2]]])
    got = midi.prmat2c_to_prmat(torch.from_numpy(x).cuda())
    assert got.dtype == np.int64 and np.array_equal(got, G[f"{name}_prmat"])


@pytest.mark.parametrize("tag", ["plain", "mask", "custom"])
def test_midi_file_holds_the_reference_notes(tmp_path, tag):
    name = "a"
    x = synth.prmat2c_image(int(G[f"{name}_seed"]), *[int(v) for v in G[f"{name}_shape"][[0, 2]]])
    mask = None
    if tag == "mask":
        mask = (np.random.Generator(np.random.PCG64(111)).random(x.shape) < 0.5).astype(np.float32)
    path = str(tmp_path / "gen.mid")
    midi.prmat2c_to_midi_file(torch.from_numpy(x).cuda(), path, labels=["C", "G", "Am"], is_custom_round=(tag == "custom"),
                              inp_mask=None if mask is None else torch.from_numpy(mask).cuda())
    tracks, lyrics, division, tempo = midi.read_smf(path)
    want = G[f"{name}_{tag}_notes"]                                              # rows: instrument, pitch, start, end, velocity
    assert len(tracks) == int(G[f"{name}_{tag}_ninstr"])
    for i, got in enumerate(tracks):
        rows = want[want[:, 0] == i]
        assert sorted(got) == sorted((int(p), round(s * 440), round(e * 440)) for _, p, s, e, _ in rows)
    assert lyrics == [("C", 0.0), ("G", 16.0), ("Am", 32.0)] and division == 220 and tempo == 500000
