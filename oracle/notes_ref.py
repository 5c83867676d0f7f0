"""CPU restatement of the reference's OUTPUT step: piano-roll image -> notes.

TEST INFRASTRUCTURE ONLY - imported by tests/ (and nothing under polyffusion_amd/).  Pinned against the real
reference by tests/golden/notes.npz (tools/make_goldens_notes.py runs the reference's own functions).

Reference semantics restated here (ref = /root/reference/polyffusion/utils.py):
* prmat2c_to_prmat            ref:utils.py:240-269   durations at onsets, [N*ratio, n_step, 128] int64
* prmat2c_to_midi_file        ref:utils.py:433-485   the note list it hands to pretty_midi (pitch, start, end, velocity 80),
                                                     split into "origin" / "inpainted" by inp_mask, one bar = n_step/8 s
Both use Python ``int(round(v)) > 0`` on float32 values: round-half-to-even, so the predicate is exactly ``v > 0.5``
(0.5 rounds to 0, 1.5 to 2, negatives to <= 0); ``custom_round`` (ref:utils.py:395-399) is ``0.95 < v < 1.05`` and is
applied to the ONSET only (the sustain test always uses round).
"""
from __future__ import annotations

import numpy as np


def _on(v, custom=False):
    v = np.asarray(v, dtype=np.float32)
    return ((v > 0.95) & (v < 1.05)) if custom else (v > 0.5)


def durations(prmat2c: np.ndarray, is_custom_round: bool = False) -> np.ndarray:
    """[N, 2, S, 128] float -> [N, S, 128] int64: note length in steps at every onset, 0 elsewhere.
    A note runs while the SUSTAIN channel of the following steps is on (ref:utils.py:259-263 / 461-465)."""
    x = np.asarray(prmat2c, dtype=np.float32)
    n, _, s, k = x.shape
    onset, sus = _on(x[:, 0], is_custom_round), _on(x[:, 1])
    run = np.zeros((n, s + 1, k), dtype=np.int64)          # run[t] = length of the sustain run starting at step t
    for t in range(s - 1, -1, -1):
        run[:, t] = np.where(sus[:, t], run[:, t + 1] + 1, 0)
    return np.where(onset, 1 + run[:, 1:], 0)


def prmat2c_to_prmat(prmat2c: np.ndarray, n_step: int = 32) -> np.ndarray:
    """ref:utils.py:240-269 - [N, 2, 32*ratio, 128] -> [N*ratio, 32, 128] int64 (a pure reshape of the durations)."""
    d = durations(prmat2c)
    n, s, k = d.shape
    return d.reshape(n * (s // n_step), n_step, k)


def note_lists(prmat2c: np.ndarray, inp_mask=None, is_custom_round: bool = False):
    """ref:utils.py:433-476 - (origin, inpainted) lists of (pitch, start_s, end_s) in the reference's append order
    (bar, step, key); a bar lasts S/8 seconds and a note is clipped to the end of its bar."""
    d = durations(prmat2c, is_custom_round)
    n, s, _ = d.shape
    t_bar = int(s / 8)
    origin, inpainted = [], []
    for b, step, key in zip(*np.nonzero(d)):
        t = b * t_bar
        note = (int(key), t + step * 1 / 8, min(t + (step + int(d[b, step, key])) * 1 / 8, t + t_bar))
        if inp_mask is not None and inp_mask[b, 0, step, key] == 0.0:
            inpainted.append(note)
        else:
            origin.append(note)
    return origin, inpainted
