"""Output step on CPU: the oracle restatement against the vectors the REAL reference produced
(tests/golden/notes.npz, tools/make_goldens_notes.py), and the host-side standard-MIDI-file writer."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import notes_ref  # noqa: E402
from polyffusion_amd import midi, synth  # noqa: E402

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "notes.npz"))


def _x(name):
    return synth.prmat2c_image(int(G[f"{name}_seed"]), *[int(v) for v in G[f"{name}_shape"][[0, 2]]])


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_oracle_durations_match_reference(name):
    x = _x(name)
    assert np.array_equal(notes_ref.prmat2c_to_prmat(x), G[f"{name}_prmat"])          # bit-exact (integers)


@pytest.mark.parametrize("name", ["a", "b", "c"])
@pytest.mark.parametrize("tag", ["plain", "mask", "custom"])
def test_oracle_note_lists_match_reference(name, tag):
    x = _x(name)
    mask = None
    if tag == "mask":
        n, _, steps, _ = x.shape
        seed = {"a": 11, "b": 12, "c": 13}[name] + 100
        mask = (np.random.Generator(np.random.PCG64(seed)).random((n, 2, steps, 128)) < 0.5).astype(np.float32)
    origin, inpainted = notes_ref.note_lists(x, inp_mask=mask, is_custom_round=(tag == "custom"))
    rows = np.array([(0, *nt, 80) for nt in origin] + [(1, *nt, 80) for nt in inpainted], dtype=np.float64).reshape(-1, 5)
    assert np.array_equal(rows, G[f"{name}_{tag}_notes"])                              # same notes, same order, exact times
    assert int(G[f"{name}_{tag}_ninstr"]) == (2 if mask is not None else 1)


def test_smf_writer_round_trip(tmp_path):
    x = _x("a")
    mask = (np.random.Generator(np.random.PCG64(7)).random(x.shape) < 0.5).astype(np.float32)
    origin, inpainted = notes_ref.note_lists(x, inp_mask=mask)
    path = str(tmp_path / "t.mid")
    midi.write_smf(path, [origin, inpainted], lyrics=[("A", 0.0), ("B", 16.0)])
    tracks, lyrics, division, tempo = midi.read_smf(path)
    assert division == 220 and tempo == 500000                      # pretty_midi's defaults: 220 ticks/beat, 120 bpm
    assert lyrics == [("A", 0.0), ("B", 16.0)]
    for want, got in zip((origin, inpainted), tracks):
        assert sorted(got) == sorted((p, round(s * 440), round(e * 440)) for p, s, e in want)   # 1/8 s = 55 ticks: exact


def test_smf_writer_empty_and_overlap(tmp_path):
    path = str(tmp_path / "e.mid")
    midi.write_smf(path, [[]])
    tracks, lyrics, _, _ = midi.read_smf(path)
    assert tracks == [[]] and lyrics == []
    # the same pitch re-struck while still sounding (the reference happily emits that): both notes survive the round trip
    notes = [(60, 0.0, 1.0), (60, 0.5, 1.5)]
    midi.write_smf(path, [notes])
    tracks, _, _, _ = midi.read_smf(path)
    assert sorted(tracks[0]) == [(60, 0, 440), (60, 220, 660)]


def test_check_prmat2c_integrity_vs_reference(golden):
    """utils.check_prmat2c_integrity (ref:utils.py:402-430) - the product's vectorised form against values the reference's own
    function returned for the seeded images (tools/make_goldens_notes.py)."""
    from polyffusion_amd import midi, synth
    g = golden("notes.npz")
    for name in ("a", "b", "c"):
        n, _, steps, _ = (int(v) for v in g[f"{name}_shape"])
        x = synth.prmat2c_image(int(g[f"{name}_seed"]), n, steps)
        assert abs(midi.check_prmat2c_integrity(x) - float(g[f"{name}_integrity"])) < 1e-12
        assert abs(midi.check_prmat2c_integrity(x, is_custom_round=True) - float(g[f"{name}_integrity_custom"])) < 1e-12


def test_check_prmat2c_integrity_negative_overshoot(golden):
    """`int(round(v)) == 0` is false below -0.5 (round(-0.7) == -1), so a negative overshoot in the previous cell makes the
    next sustain cell a continuation upstream; pinned on a noisy image by the reference's own function."""
    import numpy as np
    from polyffusion_amd import midi, synth
    g = golden("notes.npz")
    n, _, steps, _ = (int(v) for v in g["neg_shape"])
    seed = int(g["neg_seed"])
    x = synth.prmat2c_image(seed, n, steps)
    x = (x + 0.45 * np.random.Generator(np.random.PCG64(seed + 7)).standard_normal(x.shape)).astype(np.float32)
    assert (x < -0.5).sum() > 100
    assert abs(midi.check_prmat2c_integrity(x) - float(g["neg_integrity"])) < 1e-12
    assert abs(midi.check_prmat2c_integrity(x, is_custom_round=True) - float(g["neg_integrity_custom"])) < 1e-12
