"""Input side of the generation path (SURVEY.md 8f, f4 - the part that needs no MIDI parser or chord extractor):
a quantised song (note matrix + beat tables + chord track) -> the model's condition / inpainting tensors.

Mirror of ``data/datasample.py:DataSample`` (ref:data/datasample.py:29-216) and the conversions it uses from
``utils.py`` (``nmat_to_prmat2c`` ref:utils.py:220-238, ``nmat_to_prmat`` :212-217, ``chd_to_onehot`` :194-200), same
names and results, for the ``data`` dictionary ``get_data_for_single_midi`` / the POP909 ``.npz`` files hold:

    notes        [N, 5] int   (onset bin, pitch, duration in bins, velocity, program), sorted by onset
    start_table  {bin: first row of ``notes`` at or after that bin}
    db_pos       downbeat positions in bins, db_pos_filter  bool mask of the usable ones
    chord        [beats, 14] int  (root, 12 chroma flags, bass) - one row per beat (4 bins)

A segment is 8 bars = 32 beats = 128 bins.  Host-side integer work (it runs once per generation, before the sampler).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

SEG_LGTH = 32
N_BIN = 4
SEG_LGTH_BIN = SEG_LGTH * N_BIN


def chd_to_onehot(chd: np.ndarray) -> np.ndarray:
    """ref:utils.py:194-200 - [n, 14] (root, chroma[12], bass) -> [n, 36] float32 (root one-hot | chroma | bass one-hot)."""
    n = chd.shape[0]
    out = np.zeros((n, 36), dtype=np.float32)
    out[np.arange(n), chd[:, 0]] = 1
    out[:, 12:24] = chd[:, 1:13]
    out[np.arange(n), 24 + chd[:, -1]] = 1
    return out


def nmat_to_prmat(nmat: np.ndarray, n_step: int = 32) -> np.ndarray:
    """ref:utils.py:212-217 - (onset, pitch, duration) rows -> [n_step, 128] int64 durations at onsets (later rows win)."""
    pr = np.zeros((n_step, 128), dtype=np.int64)
    for o, p, d in nmat:
        if o < n_step:
            pr[o, p] = d
    return pr


def nmat_to_prmat2c(nmat: np.ndarray, n_step: int = 32) -> np.ndarray:
    """ref:utils.py:220-238 - (onset, pitch, duration) rows -> [2, n_step, 128] float32: channel 0 onsets, channel 1 the
    sustained steps after each onset (clipped to the segment)."""
    pr = np.zeros((2, n_step, 128), dtype=np.float32)
    for o, p, d in nmat:
        if o < n_step:
            pr[0, o, p] = 1.0
            pr[1, o + 1:min(o + d, n_step), p] = 1.0
    return pr


def nmat_to_pianotree_repr(nmat: np.ndarray, n_step: int = 32, max_note_count: int = 20, dur_pad_ind: int = 2, min_pitch: int = 0,
                           pitch_sos_ind: int = 128, pitch_eos_ind: int = 129, pitch_pad_ind: int = 130) -> np.ndarray:
    """ref:utils.py:132-171 - (onset, pitch, duration) rows -> [n_step, max_note_count, 6] int64: per step a start token, the notes in
    row order as (pitch, 5 binary digits of duration - 1, durations capped at 32), an end token, padding (pitch 130, digits 2).
    A step with more than ``max_note_count - 2`` notes keeps overwriting its last slot, like the reference."""
    out = np.ones((n_step, max_note_count, 6), dtype=np.int64) * dur_pad_ind
    out[:, :, 0] = pitch_pad_ind
    out[:, 0, 0] = pitch_sos_ind
    cur = np.ones(n_step, dtype=np.int64)
    for o, p, d in nmat:
        if o >= n_step:
            continue
        out[o, cur[o], 0] = p - min_pitch
        d = min(int(d), 32)
        out[o, cur[o], 1:] = [int(c) for c in np.binary_repr(d - 1, width=5)]
        if cur[o] < max_note_count - 1:
            cur[o] += 1
    out[np.arange(n_step), cur, 0] = pitch_eos_ind
    return out


class DataSample:
    """ref:data/datasample.py:29-216 (``__getitem__`` / ``get_whole_song_data``)."""

    def __init__(self, data) -> None:
        self.notes = np.asarray(data["notes"])
        st = data["start_table"]
        self.start_table = st.item() if isinstance(st, np.ndarray) else dict(st)
        db = np.asarray(data["db_pos"])
        self.db_pos_filter = np.asarray(data["db_pos_filter"])
        self.db_pos = db[self.db_pos_filter]
        if len(self.db_pos) != 0:
            self.last_db = self.db_pos[-1]
        self.chord = np.asarray(data["chord"]).astype(np.int32)

    @classmethod
    def from_npz(cls, path: str) -> "DataSample":
        with np.load(path, allow_pickle=True) as z:
            return cls({k: z[k] for k in ("notes", "start_table", "db_pos", "db_pos_filter", "chord")})

    def __len__(self) -> int:
        return len(self.db_pos)

    def note_mat_seg_at_db(self, db: int) -> np.ndarray:
        """Rows of ``notes`` whose onset lies in [db, db + 128) (ref:data/datasample.py:85-97, including its open-ended tail)."""
        s = self.start_table[db]
        if db + SEG_LGTH_BIN in self.start_table:
            return self.notes[s:self.start_table[db + SEG_LGTH_BIN]].copy()
        return self.notes[s:].copy()

    def _nmat(self, db: int) -> np.ndarray:
        seg = self.note_mat_seg_at_db(db)
        seg[:, 0] -= db
        return seg[:, :3].astype(np.int64)          # onset, pitch, duration

    def __getitem__(self, idx: int):
        db = int(self.db_pos[idx])
        nmat = self._nmat(db)
        chord = self.chord[db // N_BIN:db // N_BIN + SEG_LGTH]
        if chord.shape[0] < SEG_LGTH:
            chord = np.append(chord, np.zeros([SEG_LGTH - chord.shape[0], 14], dtype=np.int32), axis=0)
        return nmat_to_prmat2c(nmat, SEG_LGTH_BIN), nmat_to_pianotree_repr(nmat, SEG_LGTH_BIN), chord, nmat_to_prmat(nmat, SEG_LGTH_BIN)

    def get_whole_song_data(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """Consecutive non-overlapping 8-bar segments from the first usable downbeat on (ref:data/datasample.py:191-216):
        ``prmat2c [S, 2, 128, 128]``, ``pnotree [S, 128, 20, 6]`` int64, ``chord [S, 32, 36]`` one-hot, ``prmat [S, 128, 128]`` (float32)."""
        prmat2c, pnotree, chord, prmat = [], [], [], []
        idx = i = 0
        while i < len(self):
            p2, pt, c, pm = self[i]
            prmat2c.append(p2)
            pnotree.append(pt)
            chord.append(chd_to_onehot(c))
            prmat.append(pm)
            idx += SEG_LGTH_BIN
            while i < len(self) and self.db_pos[i] < idx:
                i += 1
        return (torch.from_numpy(np.array(prmat2c, dtype=np.float32)), torch.from_numpy(np.array(pnotree, dtype=np.int64)),
                torch.from_numpy(np.array(chord, dtype=np.float32)), torch.from_numpy(np.array(prmat, dtype=np.float32)))


# ---- validation-set song files (ref:data/dataset.py:27-253, data/dataset_musicalion.py:25-208) ----------------------------------------
# PINNED ON SYNTHETIC FILES OF THE DOCUMENTED LAYOUT (tests/golden/dataset.npz): the POP909 / Musicalion .npz collections are not part of the reference checkout (only the split lists under
# data/train_split_pnt/ are) and no fixture of the reference holds one of their files; the classes below restate the reference's
# loaders for the file layout those loaders read; tests/test_datasample.py holds them, bit for bit, to what the REAL reference loaders
# returned on synthetic files of that layout (tests/golden/dataset.npz, tools/make_goldens_dataset.py).
POP909_DATA_DIR = "data/POP909_4_bin_pnt_8bar"              # ref:dirs.py:8-9
MUSICALION_DATA_DIR = "data/musicalion_solo_piano_4_bin_pnt"
TRAIN_SPLIT_DIR = "data/train_split_pnt"                    # ref:dirs.py:5


class DataSampleNpz(DataSample):
    """ref:data/dataset.py:27-253 - one POP909 song: ``notes`` / ``start_table`` hold one entry per TRACK (melody, bridge, piano) and
    ``use_track`` says which of them make up the segment (rows of the chosen tracks one after the other, in track order - the
    reference does not re-sort them, :103-124).  A file with a single note matrix (0-d ``start_table``) is read like ``DataSample``."""

    def __init__(self, song_fn: str, use_track=(0, 1, 2), data_dir: str = POP909_DATA_DIR) -> None:
        import os
        self.song_fn, self.use_track = song_fn, list(use_track)
        with np.load(os.path.join(data_dir, song_fn), allow_pickle=True) as z:
            data = {k: z[k] for k in ("notes", "start_table", "db_pos", "db_pos_filter", "chord")}
        st = data["start_table"]
        self.multi_track = st.ndim > 0                       # ref:dataset.py:103 `len(self.start_table.shape) > 0`
        if self.multi_track:
            self.notes = [np.asarray(n) for n in data["notes"]]
            self.start_table = [dict(t) for t in st]
        else:
            self.notes, self.start_table = np.asarray(data["notes"]), st.item()
        db = np.asarray(data["db_pos"])
        self.db_pos_filter = np.asarray(data["db_pos_filter"])
        self.db_pos = db[self.db_pos_filter]
        if len(self.db_pos) != 0:
            self.last_db = self.db_pos[-1]
        self.chord = np.asarray(data["chord"]).astype(np.int32)

    def note_mat_seg_at_db(self, db: int) -> np.ndarray:
        if not self.multi_track:
            return DataSample.note_mat_seg_at_db(self, db)
        rows = []
        for t in self.use_track:
            notes, table = self.notes[t], self.start_table[t]
            s = table[db]
            seg = notes[s:table[db + SEG_LGTH_BIN]] if db + SEG_LGTH_BIN in table else notes[s:]
            rows.extend(np.array(seg))
        out = np.array(rows)
        return out if out.size else np.zeros([0, 5], dtype=np.int64)


class DataSampleNpz_Musicalion(DataSample):
    """ref:data/dataset_musicalion.py:25-208 - one Musicalion song: a single note matrix, NO chord track (``chord`` comes back ``None``,
    so only texture / piano-tree conditioned models can use it - the reference asserts ``cond_type != "chord"``, inference_sdf.py:620)."""

    def __init__(self, song_fn: str, data_dir: str = MUSICALION_DATA_DIR) -> None:
        import os
        self.song_fn = song_fn
        with np.load(os.path.join(data_dir, song_fn), allow_pickle=True) as z:
            self.notes, self.start_table = np.asarray(z["notes"]), z["start_table"].item()
            db, self.db_pos_filter = np.asarray(z["db_pos"]), np.asarray(z["db_pos_filter"])
        self.db_pos = db[self.db_pos_filter]
        if len(self.db_pos) != 0:
            self.last_db = self.db_pos[-1]
        self.chord = None

    def __getitem__(self, idx: int):
        db = int(self.db_pos[idx])
        nmat = self._nmat(db)
        return nmat_to_prmat2c(nmat, SEG_LGTH_BIN), nmat_to_pianotree_repr(nmat, SEG_LGTH_BIN), None, nmat_to_prmat(nmat, SEG_LGTH_BIN)

    def get_whole_song_data(self):
        prmat2c, pnotree, prmat = [], [], []
        idx = i = 0
        while i < len(self):
            p2, pt, _, pm = self[i]
            prmat2c.append(p2); pnotree.append(pt); prmat.append(pm)
            idx += SEG_LGTH_BIN
            while i < len(self) and self.db_pos[i] < idx:
                i += 1
        return (torch.from_numpy(np.array(prmat2c, dtype=np.float32)), torch.from_numpy(np.array(pnotree, dtype=np.int64)), None,
                torch.from_numpy(np.array(prmat, dtype=np.float32)))


def load_split(path: str):
    """ref:inference_sdf.py:96-98 - the (train files, validation files) pair of a split pickle.  The reference calls ``pickle.load``; a
    split is two lists of file names, so this loader admits exactly that (tuples / lists of str) and refuses every global."""
    import pickle

    class _NoGlobals(pickle.Unpickler):
        def find_class(self, module, name):
            raise pickle.UnpicklingError(f"split pickle refers to {module}.{name}: only lists of file names are expected")

    with open(path, "rb") as f:
        split = _NoGlobals(f).load()
    if not (isinstance(split, (tuple, list)) and len(split) == 2 and all(isinstance(s, (list, tuple)) and all(isinstance(n, str) for n in s) for s in split)):
        raise ValueError(f"{path}: expected a (train, validation) pair of file-name lists")
    return list(split[0]), list(split[1])


def choose_song_from_val_dl(dataset: str, index=None, use_track=(0, 1, 2), data_dir=None, split_dir: str = TRAIN_SPLIT_DIR, ask=input):
    """ref:inference_sdf.py:95-118 (``choose_song_from_val_dl`` / ``..._musicalion``): a song of the VALIDATION half of the split, by
    ``index`` or - like the reference - by asking on the terminal.  Returns (sample object, file name)."""
    import os
    if dataset not in ("pop909", "musicalion"):
        raise NotImplementedError(dataset)                       # ref:inference_sdf.py:590, 623
    val = load_split(os.path.join(split_dir, f"{dataset}.pickle"))[1]
    if index is None:
        print(val)
        index = int(ask(f"choose one from {dataset}:"))
    song_fn = val[int(index)]
    if dataset == "pop909":
        return DataSampleNpz(song_fn, use_track, data_dir or POP909_DATA_DIR), song_fn
    return DataSampleNpz_Musicalion(song_fn, data_dir or MUSICALION_DATA_DIR), song_fn
