"""Synthetic song files in the layouts the reference's dataset loaders read (ref:data/dataset.py:70-84 POP909: per-track note matrices
and start tables; ref:data/dataset_musicalion.py:67-75: one note matrix, no chords).  Deterministic from the seed; used by
tools/make_goldens_dataset.py (which runs the REAL reference loaders on these files) and by tests/test_datasample.py."""
import os
import pickle

import numpy as np


def _track(rng, n, n_bins, table_upto):
    """n notes with sorted onsets in [0, n_bins): rows (onset, pitch, duration, velocity, program); start table bin -> first row."""
    on = np.sort(rng.integers(0, n_bins, n))
    nm = np.stack([on, rng.integers(30, 90, n), rng.integers(1, 12, n), np.full(n, 80), np.zeros(n, int)], 1).astype(np.int64)
    return nm, {int(b): int(np.searchsorted(on, b)) for b in range(0, table_upto + 1)}


def pop909_song(seed: int, kind: str = "full"):
    """-> dict of the arrays np.savez writes.  kind: 'full' (three tracks, tables past the last segment, 128 beats of chords),
    'ragged' (tables end INSIDE the last segment -> the reference's `notes[s_ind:]` branch; chords shorter than the last segment ->
    zero padding; a silent stretch -> an empty segment; an irregular downbeat grid), 'single' (one note matrix, 0-d start table)."""
    rng = np.random.default_rng(1000 + seed)
    if kind == "single":
        nm, table = _track(rng, 70, 384, 384)
        db = np.arange(0, 384, 16)
        filt = np.ones(len(db), bool)
        filt[-8:] = False
        chord = np.zeros((96, 14), int)
        chord[:, 0], chord[:, 13] = rng.integers(0, 12, 96), rng.integers(0, 12, 96)
        chord[:, 1:13] = rng.integers(0, 2, (96, 12))
        return dict(notes=nm, start_table=np.array(table, dtype=object), db_pos=db, db_pos_filter=filt, chord=chord)
    n_bins = 512
    upto = 512 if kind == "full" else 470
    sizes = (60, 30, 90) if kind == "full" else (40, 0, 55)
    ts = [_track(rng, n, n_bins, upto) for n in sizes]
    if kind == "ragged":                     # a silent stretch in every track: bins [128, 272) hold no onset
        ts2 = []
        for nm, _ in ts:
            nm = nm[(nm[:, 0] < 128) | (nm[:, 0] >= 272)]
            ts2.append((nm, {int(b): int(np.searchsorted(nm[:, 0], b)) for b in range(0, upto + 1)}))
        ts = ts2
    notes, st = np.empty(3, dtype=object), np.empty(3, dtype=object)
    for i, (a, b) in enumerate(ts):
        notes[i], st[i] = a, b
    db = np.arange(0, n_bins, 16) if kind == "full" else np.array([0, 16, 32, 44, 60, 76, 92, 108, 124, 140, 156, 172, 188, 204, 220, 236, 252, 268,
                                                                    284, 300, 316, 332, 348, 364])
    filt = np.ones(len(db), bool)
    if kind == "full":
        filt[-8:] = False
    else:
        filt[[3, 9]] = False
    n_ch = 128 if kind == "full" else 100     # ragged: segment at bin 364 wants chord rows 91 .. 122
    chord = np.zeros((n_ch, 14), int)
    chord[:, 0], chord[:, 13] = rng.integers(0, 12, n_ch), rng.integers(0, 12, n_ch)
    chord[:, 1:13] = rng.integers(0, 2, (n_ch, 12))
    return dict(notes=notes, start_table=st, db_pos=db, db_pos_filter=filt, chord=chord)


def musicalion_song(seed: int):
    rng = np.random.default_rng(2000 + seed)
    nm, table = _track(rng, 120, 400, 390)          # the table ends inside the last usable segment
    db = np.arange(0, 400, 16)
    filt = np.ones(len(db), bool)
    filt[-7:] = False
    filt[2] = False
    return dict(notes=nm, start_table=np.array(table, dtype=object), db_pos=db, db_pos_filter=filt)


SONGS = {"pop_full.npz": ("pop909", 1, "full"), "pop_ragged.npz": ("pop909", 2, "ragged"), "pop_single.npz": ("pop909", 3, "single"),
         "mus_a.npz": ("musicalion", 4, None)}


def write_all(pop_dir: str, mus_dir: str, split_dir: str):
    """The four songs as files + split pickles whose validation halves list them."""
    for d in (pop_dir, mus_dir, split_dir):
        os.makedirs(d, exist_ok=True)
    for fn, (ds, seed, kind) in SONGS.items():
        if ds == "pop909":
            np.savez(os.path.join(pop_dir, fn), **pop909_song(seed, kind))
        else:
            np.savez(os.path.join(mus_dir, fn), **musicalion_song(seed))
    with open(os.path.join(split_dir, "pop909.pickle"), "wb") as f:
        pickle.dump((["train_only.npz"], [fn for fn, v in SONGS.items() if v[0] == "pop909"]), f)
    with open(os.path.join(split_dir, "musicalion.pickle"), "wb") as f:
        pickle.dump((["train_only.npz"], [fn for fn, v in SONGS.items() if v[0] == "musicalion"]), f)
