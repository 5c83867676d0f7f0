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
