"""Seeded synthetic inputs of the shapes the reference pipeline feeds the hot path.

There are no datasets or MIDI files on the GPU box, so benchmarks, parity tests and
golden vectors all draw from these numpy (PCG64) generators; the same seed gives the
same array everywhere.  Shapes follow SURVEY.md 8(d) "Value distributions / seeds".
"""
from __future__ import annotations

import numpy as np


def gaussian(shape, seed: int) -> np.ndarray:
    return np.random.Generator(np.random.PCG64(seed)).standard_normal(tuple(shape)).astype(np.float32)


def chords(batch: int, seed: int, n_step: int = 32) -> np.ndarray:
    """[B, 32, 36]: one-hot root (12) | Bernoulli(0.3) chroma (12) | one-hot bass (12)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.zeros((batch, n_step, 36), np.float32)
    root = rng.integers(0, 12, (batch, n_step))
    bass = rng.integers(0, 12, (batch, n_step))
    out[..., 12:24] = (rng.random((batch, n_step, 12)) < 0.3).astype(np.float32)
    np.put_along_axis(out[..., 0:12], root[..., None], 1.0, axis=-1)
    np.put_along_axis(out[..., 24:36], bass[..., None], 1.0, axis=-1)
    return out


def prmat(batch: int, seed: int, steps: int = 128, pitches: int = 128) -> np.ndarray:
    """[B, 128, 128] sparse integer durations: Bernoulli(0.02) * U{1..16}."""
    rng = np.random.Generator(np.random.PCG64(seed))
    on = rng.random((batch, steps, pitches)) < 0.02
    dur = rng.integers(1, 17, (batch, steps, pitches))
    return (on * dur).astype(np.float32)


def prmat2c_image(seed: int, n: int, steps: int = 128) -> np.ndarray:
    """A generated-sample-like ``[n, 2, steps, 128]`` onset/sustain image for the output step (notes / MIDI): sparse onsets,
    sustain runs, Gaussian jitter, and a sprinkle of the exact values the reference's rounding rules hinge on."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = np.zeros((n, 2, steps, 128), dtype=np.float32)
    x[:, 0] = (rng.random((n, steps, 128)) < 0.03) * (0.6 + 0.8 * rng.random((n, steps, 128)))
    x[:, 1] = (rng.random((n, steps, 128)) < 0.35) * (0.3 + 0.9 * rng.random((n, steps, 128)))
    x += 0.05 * rng.standard_normal(x.shape).astype(np.float32)
    flat = x.reshape(-1)
    idx = rng.choice(flat.size, 600, replace=False)
    flat[idx] = rng.choice(np.array([0.5, 1.5, -0.5, 0.95, 1.05, 1.0, 0.50000006, 0.49999997], dtype=np.float32), 600)
    return x


def song_data(seed: int, n_bars: int = 20) -> dict:
    """A quantised song in the dictionary format of the reference's ``get_data_for_single_midi`` / POP909 ``.npz`` files
    (notes [N,5], start_table, db_pos, db_pos_filter, chord [beats,14]); 4 bins per beat, 4 beats per bar."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n_bins = n_bars * 16
    rows = []
    for o in range(n_bins):
        for _ in range(int(rng.integers(0, 3))):
            rows.append((o, int(rng.integers(36, 96)), int(rng.integers(1, 24)), int(rng.integers(40, 110)), 0))
    notes = np.array(rows, dtype=np.int64).reshape(-1, 5)
    start_table, r = {}, 0
    for o in range(n_bins + 1):
        while r < len(notes) and notes[r, 0] < o:
            r += 1
        start_table[o] = r
    db_pos = np.arange(0, n_bins, 16, dtype=np.int64)
    filt = np.ones(len(db_pos), dtype=bool)
    filt[-2:] = False                                  # like the reference: the last downbeats cannot start a segment
    chord = np.zeros((n_bars * 4, 14), dtype=np.int64)
    chord[:, 0] = rng.integers(0, 12, n_bars * 4)
    chord[:, 1:13] = rng.random((n_bars * 4, 12)) < 0.3
    chord[:, 13] = rng.integers(0, 12, n_bars * 4)
    return {"notes": notes, "start_table": np.array(start_table, dtype=object), "db_pos": db_pos, "db_pos_filter": filt, "chord": chord}


def pnotree(n: int, seed: int, n_step: int = 128) -> np.ndarray:
    """Seeded piano-tree grids [n, n_step, 20, 6] (int64) of sparse random notes - the condition of the sdf_pnotree variant."""
    from .datasample import nmat_to_pianotree_repr
    out = []
    for i in range(n):
        rng = np.random.Generator(np.random.PCG64([seed, i]))
        k = int(rng.integers(60, 200))
        nm = np.stack([rng.integers(0, n_step, k), rng.integers(36, 96, k), rng.integers(1, 17, k)], 1)
        out.append(nmat_to_pianotree_repr(nm[np.lexsort((nm[:, 2], nm[:, 1], nm[:, 0]))], n_step))
    return np.stack(out)
