"""MIDI file -> the quantised-song dictionary ``DataSample`` reads: restatement of
``/root/reference/polyffusion/data/midi_to_data.py`` (``get_data_for_single_midi`` :219-241 and the helpers it calls).

    notes         get_note_matrix :19-46 + dedup_note_matrix :49-66   rows (onset, pitch, duration, velocity, program) at 4 bins per beat
    chord         extract_chords_from_midi_file + get_chord_matrix :88-120 -> polyffusion_amd.chord_extractor (pinned by the
                  reference's example.mid / example.out and its vendored mir_eval)
    db_pos(_filter) get_downbeat_pos_and_filter :151-197   barline positions and whether a full 8-bar window of equal bars follows
    start_table   get_start_table :200-214

PARITY UNPINNED for the note / barline side: the reference reads the file with ``muspy.read_midi`` + ``Music.adjust_resolution(4)`` +
``Music.infer_barlines_and_beats()`` (muspy 0.5.0 in its requirements.txt), muspy is not in this image and the reference holds no
fixture for these functions.  ``Music`` below restates muspy's published behaviour: notes paired first-in-first-out per (channel,
pitch) within a track, one track per (MIDI track, channel, program), every time attribute rescaled to the new resolution and rounded
separately (``time`` and ``duration``: Python ``round``, half to even), barlines every ``resolution * 4 * numerator / denominator``
from each time signature to the next (the last one to the end of the music).  The functions on top of it follow the reference line
by line and are property-tested through a write -> read round trip (tests/test_frontend.py).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np

from . import chord_extractor
from .chord_extractor import _read_tracks

ONE_BEAT = 0.5
SEG_LGTH = 32
BEAT = 4
BIN = 4
SEG_LGTH_BIN = SEG_LGTH * BIN


class Track:
    def __init__(self, program: int, is_drum: bool):
        self.program, self.is_drum = program, is_drum
        self.notes: List[List[int]] = []        # [time, pitch, duration, velocity]


class Music:
    """The subset of ``muspy.Music`` (0.5.0) the front end touches: ``read_midi``, ``adjust_resolution``,
    ``infer_barlines_and_beats`` (barlines only), ``get_end_time``."""

    def __init__(self, path: str):
        self.resolution, raw = _read_tracks(open(path, "rb").read())
        self.time_signatures: List[List[int]] = []     # [time, numerator, denominator]
        self._other_times: List[int] = []              # tempo / key-signature events count towards the end time
        self.tracks: List[Track] = []
        for ev in raw:
            by_key: Dict[tuple, Track] = {}
            active: Dict[tuple, list] = {}
            program = [0] * 16
            for tick, kind, args in ev:
                if kind == "meta":
                    mk, payload = args
                    if mk == 0x58:
                        self.time_signatures.append([tick, payload[0], 2 ** payload[1]])
                    elif mk in (0x51, 0x59):
                        self._other_times.append(tick)
                elif kind == 0xC0:
                    program[args[0]] = args[1]
                elif kind == 0x90 and args[2] > 0:
                    active.setdefault((args[0], args[1]), []).append((tick, args[2]))
                elif kind == 0x80 or (kind == 0x90 and args[2] == 0):
                    q = active.get((args[0], args[1]))
                    if not q:
                        continue
                    onset, vel = q.pop(0)                    # duplicate_note_mode = "fifo"
                    key = (args[0], program[args[0]])
                    if key not in by_key:
                        by_key[key] = Track(program[args[0]], args[0] == 9)
                        self.tracks.append(by_key[key])
                    by_key[key].notes.append([onset, args[1], tick - onset, vel])
        self.time_signatures.sort(key=lambda t: t[0])
        self.barlines: List[float] = []

    def adjust_resolution(self, target: int):
        f = target / self.resolution
        r = lambda v: int(round(v * f))
        for t in self.tracks:
            for n in t.notes:
                n[0], n[2] = r(n[0]), r(n[2])
        for ts in self.time_signatures:
            ts[0] = r(ts[0])
        self._other_times = [r(v) for v in self._other_times]
        self.resolution = target

    def get_end_time(self) -> int:
        ends = [n[0] + n[2] for t in self.tracks for n in t.notes] + [ts[0] for ts in self.time_signatures] + self._other_times
        return max(ends) if ends else 0

    def infer_barlines(self):
        self.barlines = []
        end = self.get_end_time()
        for i, (time, num, den) in enumerate(self.time_signatures):
            stop = self.time_signatures[i + 1][0] if i + 1 < len(self.time_signatures) else end
            step = self.resolution * 4 * num / den
            t = float(time)
            while t < stop:
                self.barlines.append(t)
                t += step


def get_note_matrix(music: Music) -> List[list]:
    """:19-46 - every note of every track with a positive duration, sorted by (onset, pitch, duration)."""
    notes = [[int(n[0]), n[1], int(n[2]), n[3], t.program] for t in music.tracks for n in t.notes if int(n[2]) > 0]
    notes.sort(key=lambda x: (x[0], x[1], x[2]))
    return notes


def dedup_note_matrix(notes: List[list]) -> List[list]:
    """:49-66 - drop a note whose (onset, pitch) equals the previous row's (the same note doubled on another track)."""
    out, last = [], None
    for i, n in enumerate(notes):
        if i == 0 or n[:2] != last[:2]:
            out.append(n)
        last = n
    return out


def get_downbeat_pos_and_filter(music: Music):
    """:151-197 - barline positions; a downbeat is usable when its bar is 2, 4 or 8 beats long and enough bars of the same length follow
    to fill 8 beats... (the reference's own rule: ``left = 8 * BIN - length`` bins of equal bars)."""
    music.infer_barlines()
    if any(not float(b).is_integer() for b in music.barlines):
        return None, None
    db_pos = [int(b) for b in music.barlines]
    diff = np.diff(db_pos).tolist()
    diff.append(diff[len(diff) - 1])
    flt = []
    for i in range(len(db_pos)):
        if diff[i] not in {2 * BIN, 4 * BIN, 8 * BIN}:
            flt.append(False)
            continue
        length, left, idx, bad = diff[i], 8 * BIN - diff[i], i + 1, False
        while left > 0 and idx < len(db_pos):
            if diff[idx] != length:
                bad = True
                break
            left -= length
            idx += 1
        flt.append(not bad)
    return db_pos, flt


def get_start_table(notes: List[list], db_pos: List[int]) -> Dict[int, int]:
    """:200-214 - downbeat bin -> first row of ``notes`` at or after it."""
    row, table = 0, {}
    for db in db_pos:
        while row < len(notes) and notes[row][0] < db:
            row += 1
        table[db] = row
    return table


def get_chord_matrix(chdfile_path: str) -> List[list]:
    """:88-120 (the lab file the extractor wrote)."""
    return chord_extractor.chord_matrix_from_labels(chord_extractor.read_chord_lab(chdfile_path), ONE_BEAT).tolist()


def get_data_for_single_midi(fpath: str, chdfile_path: Optional[str] = None) -> Optional[dict]:
    """:219-241.  ``chdfile_path``: where the extracted chord labels are written (the reference always writes them); None keeps them in
    memory."""
    music = Music(fpath)
    music.adjust_resolution(BIN)
    if len(music.time_signatures) == 0:
        music.time_signatures.append([0, 4, 4])
    note_mat = dedup_note_matrix(get_note_matrix(music))
    if chdfile_path is not None and os.path.dirname(chdfile_path):
        os.makedirs(os.path.dirname(chdfile_path), exist_ok=True)
    labels = chord_extractor.transcribe_midi(fpath, chdfile_path)
    chord = chord_extractor.chord_matrix_from_labels(labels, ONE_BEAT)
    db_pos, db_filter = get_downbeat_pos_and_filter(music)
    if db_pos is None:
        print("get downbeat error!")
        return None
    return {"notes": np.array(note_mat), "start_table": np.array(get_start_table(note_mat, db_pos)), "db_pos": np.array(db_pos),
            "db_pos_filter": np.array(db_filter), "chord": np.array(chord)}
