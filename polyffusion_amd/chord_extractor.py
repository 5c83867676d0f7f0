"""Chord labels from a MIDI file: the reference's rule-based extractor, restated (the input side).

``/root/reference/polyffusion/chord_extractor`` (called by ``data/midi_to_data.py:219-230`` through
``extract_chords_from_midi_file``) is template matching plus a dynamic program over half-beats:

* ``main.py:16-71``  ``transcribe_cb1000_midi`` / ``process_chord``: beats with their position in the bar (``MidiBeatExtractor``);
* ``midi_chord.py:21-113``  ``ChordRecognition.process_feature``: per-beat chroma (longest weighted overlap per pitch class)
  and bass chroma (lowest sounding pitch of each of 8 sub-beats);
* ``extractors/rule_based_channel_reweight.py:39-52``  ``midi_to_thickness_and_bass_weights``: one weight per melodic instrument
  from the "thickness" of its 100 Hz piano roll, the lowest instrument forced to 1;
* ``chord_class.py:59-137``  the chord vocabulary (N + 12 roots x 32 qualities + listed inversions) and its template score;
* ``midi_chord.py:115-190``  ``decode``: segments of up to 12 beats, scored, never crossing more than one downbeat;
* ``io_new/chordlab_io.py:22-27``  the ``start<TAB>end<TAB>label`` lines.

The reference reads the file with pretty_midi 0.2.10 (absent from this image): ``PrettyMIDI`` below restates exactly the parts of it
the extractor touches - tempo map from track 0, instruments keyed by (program, channel, track) in order of their first finished
note, a note-off closing EVERY open note of its key, ``get_end_time`` including controllers and pitch bends, ``get_beats`` /
``get_downbeats``, and ``Instrument.get_piano_roll`` (fs = 100) with its sustain-pedal and pitch-bend passes.  Pinned by the
reference's own example: ``chord_extractor/example.mid`` -> ``example.out`` (tests/golden/chord_example.*).

Host-side numpy: this is file preparation (a few hundred beats), not part of the step loop.
"""
from __future__ import annotations

import struct
from collections import OrderedDict, defaultdict
from typing import List, Tuple

import numpy as np

# ---------------------------------------------------------------------------------------------------------------------------
# pretty_midi 0.2.10, the subset the extractor uses
# ---------------------------------------------------------------------------------------------------------------------------


class Note:
    __slots__ = ("velocity", "pitch", "start", "end")

    def __init__(self, velocity, pitch, start, end):
        self.velocity, self.pitch, self.start, self.end = velocity, pitch, start, end


class Instrument:
    def __init__(self, program: int, is_drum: bool = False, name: str = ""):
        self.program, self.is_drum, self.name = program, is_drum, name
        self.notes: List[Note] = []
        self.pitch_bends: List[Tuple[int, float]] = []       # (pitch, time)
        self.control_changes: List[Tuple[int, int, float]] = []   # (number, value, time)

    def get_end_time(self) -> float:
        events = [n.end for n in self.notes] + [b[1] for b in self.pitch_bends] + [c[2] for c in self.control_changes]
        return max(events) if events else 0.0

    def get_piano_roll(self, fs: int = 100, pedal_threshold: int = 64) -> np.ndarray:
        """pretty_midi ``Instrument.get_piano_roll(fs=100)``: velocity sums per column, sustain pedal (CC 64) holding the running
        maximum while down, pitch bends shifting / interpolating the affected columns."""
        if not self.notes:
            return np.array([[]] * 128)
        end_time = self.get_end_time()
        roll = np.zeros((128, int(fs * end_time)))
        if self.is_drum:
            return roll
        for n in self.notes:
            roll[n.pitch, int(n.start * fs):int(n.end * fs)] += n.velocity
        if pedal_threshold is not None:
            t_on, down = 0, False
            for number, value, time in self.control_changes:
                if number != 64:
                    continue
                now = int(time * fs)
                cur = value >= pedal_threshold
                if not down and cur:
                    t_on, down = now, True
                elif down and not cur:
                    roll[:, t_on:now] = np.maximum.accumulate(roll[:, t_on:now], axis=1)
                    down = False
        bends = sorted(self.pitch_bends, key=lambda b: b[1])
        ends = bends[1:] + [(0, end_time)]
        for (pitch, t0), (_, t1) in zip(bends, ends):
            if abs(pitch) < 1:
                continue
            semis = 2.0 * pitch / 8192.0                                   # pitch_bend_to_semitones, range 2
            b_int = int(np.sign(semis) * np.floor(abs(semis)))
            b_dec = abs(semis - b_int)
            cols = np.r_[int(t0 * fs):int(t1 * fs)]
            bent = np.zeros(roll[:, cols].shape)
            if pitch >= 0:
                if b_int != 0:
                    bent[b_int:] = roll[:-b_int, cols]
                else:
                    bent = roll[:, cols]
                bent[1:] = (1 - b_dec) * bent[1:] + b_dec * bent[:-1]
            else:
                if b_int != 0:
                    bent[:b_int] = roll[-b_int:, cols]
                else:
                    bent = roll[:, cols]
                bent[:-1] = (1 - b_dec) * bent[:-1] + b_dec * bent[1:]
            roll[:, cols] = bent
        return roll


def _vlq(data: bytes, p: int) -> Tuple[int, int]:
    v = 0
    while True:
        b = data[p]
        p += 1
        v = (v << 7) | (b & 0x7F)
        if not b & 0x80:
            return v, p


def _read_tracks(data: bytes):
    """Standard MIDI file -> (ticks per beat, [track = list of (absolute tick, kind, args)])."""
    if data[:4] != b"MThd":
        raise ValueError("not a standard MIDI file")
    _, ntracks, division = struct.unpack(">HHH", data[8:14])
    if division & 0x8000:
        raise ValueError("SMPTE time division is not supported")
    pos, tracks = 8 + struct.unpack(">I", data[4:8])[0], []
    for _ in range(ntracks):
        if data[pos:pos + 4] != b"MTrk":
            raise ValueError("missing MTrk chunk")
        end = pos + 8 + struct.unpack(">I", data[pos + 4:pos + 8])[0]
        p, tick, running, ev = pos + 8, 0, None, []
        while p < end:
            dt, p = _vlq(data, p)
            tick += dt
            st = data[p]
            if st == 0xFF:
                kind = data[p + 1]
                ln, q = _vlq(data, p + 2)
                ev.append((tick, "meta", (kind, data[q:q + ln])))
                p = q + ln
            elif st in (0xF0, 0xF7):
                ln, q = _vlq(data, p + 1)
                p = q + ln
            else:
                if st & 0x80:
                    running = st
                    p += 1
                hi, ch = running & 0xF0, running & 0x0F
                n = 1 if hi in (0xC0, 0xD0) else 2
                ev.append((tick, hi, (ch,) + tuple(data[p:p + n])))
                p += n
        tracks.append(ev)
        pos = end
    return division, tracks


class PrettyMIDI:
    def __init__(self, path: str):
        self.resolution, tracks = _read_tracks(open(path, "rb").read())
        max_tick = max((ev[-1][0] for ev in tracks if ev), default=0) + 1
        # tempo map: set_tempo events of track 0 only (pretty_midi _load_tempo_changes)
        scales = [(0, 60.0 / (120.0 * self.resolution))]
        self.time_signature_changes: List[Tuple[int, int, float]] = []    # (numerator, denominator, time)
        meta0 = [e for e in (tracks[0] if tracks else []) if e[1] == "meta"]
        for tick, _, (kind, payload) in meta0:
            if kind == 0x51:
                scale = 60.0 / ((6e7 / int.from_bytes(payload, "big")) * self.resolution)
                if tick == 0:
                    scales = [(0, scale)]
                elif scale != scales[-1][1]:
                    scales.append((tick, scale))
        self._tick_scales = scales
        t2t = np.zeros(max_tick + 1)
        last = 0.0
        for (s0, sc), (s1, _) in zip(scales[:-1], scales[1:]):
            t2t[s0:s1 + 1] = last + sc * np.arange(s1 - s0 + 1)
            last = t2t[s1]
        s0, sc = scales[-1]
        t2t[s0:] = last + sc * np.arange(max_tick + 1 - s0)
        self._t2t = t2t
        self._meta_times = []
        for tick, _, (kind, payload) in meta0:
            if kind == 0x58:
                self.time_signature_changes.append((payload[0], 2 ** payload[1], float(t2t[tick])))
                self._meta_times.append(float(t2t[tick]))
            elif kind in (0x59, 0x05):            # key signatures and lyrics count towards the end time
                self._meta_times.append(float(t2t[tick]))
        # instruments (pretty_midi _load_instruments)
        imap: "OrderedDict[Tuple[int, int, int], Instrument]" = OrderedDict()
        stragglers = {}

        def get(program, channel, track, create):
            key = (program, channel, track)
            if key in imap:
                return imap[key]
            if not create and (channel, track) in stragglers:
                return stragglers[(channel, track)]
            if create:
                ins = Instrument(program, channel == 9)
                if (channel, track) in stragglers:
                    ins.control_changes = stragglers[(channel, track)].control_changes
                    ins.pitch_bends = stragglers[(channel, track)].pitch_bends
                imap[key] = ins
            else:
                ins = Instrument(program)
                stragglers[(channel, track)] = ins
            return ins

        for ti, ev in enumerate(tracks):
            open_notes = defaultdict(list)
            current = np.zeros(16, dtype=int)
            for tick, kind, args in ev:
                if kind == 0xC0:
                    current[args[0]] = args[1]
                elif kind == 0x90 and args[2] > 0:
                    open_notes[(args[0], args[1])].append((tick, args[2]))
                elif kind == 0x80 or (kind == 0x90 and args[2] == 0):
                    key = (args[0], args[1])
                    if key in open_notes:
                        close = [(s, v) for s, v in open_notes[key] if s != tick]
                        keep = [(s, v) for s, v in open_notes[key] if s == tick]
                        for s, v in close:
                            get(int(current[args[0]]), args[0], ti, True).notes.append(Note(v, args[1], float(t2t[s]), float(t2t[tick])))
                        if close and keep:
                            open_notes[key] = keep
                        else:
                            del open_notes[key]
                elif kind == 0xE0:
                    value = (args[2] << 7 | args[1]) - 8192
                    get(int(current[args[0]]), args[0], ti, False).pitch_bends.append((value, float(t2t[tick])))
                elif kind == 0xB0:
                    get(int(current[args[0]]), args[0], ti, False).control_changes.append((args[1], args[2], float(t2t[tick])))
        self.instruments = list(imap.values())

    def get_end_time(self) -> float:
        times = [i.get_end_time() for i in self.instruments] + self._meta_times
        return max(times) if times else 0.0

    def get_tempo_changes(self):
        times = np.array([self._t2t[t] for t, _ in self._tick_scales])
        tempi = np.array([60.0 / (s * self.resolution) for _, s in self._tick_scales])
        return times, tempi

    def get_beats(self, start_time: float = 0.0) -> np.ndarray:
        tct, tempi = self.get_tempo_changes()
        beats = [start_time]
        ti = 0
        while ti < tct.shape[0] - 1 and beats[-1] > tct[ti + 1]:
            ti += 1
        ts = sorted(self.time_signature_changes, key=lambda t: t[2])
        si = 0
        while si < len(ts) - 1 and beats[-1] >= ts[si + 1][2]:
            si += 1

        def bpm():
            if ts:
                return _qpm_to_bpm(tempi[ti], ts[si][0], ts[si][1])
            return tempi[ti]

        close = lambda a, b: a > b or np.isclose(a, b)
        end_time = self.get_end_time()
        while beats[-1] < end_time:
            b = bpm()
            nxt = beats[-1] + 60.0 / b
            if ti < tct.shape[0] - 1 and nxt > tct[ti + 1]:
                nxt, remaining = beats[-1], 1.0
                while ti < tct.shape[0] - 1 and nxt + remaining * 60.0 / b >= tct[ti + 1]:
                    over = (tct[ti + 1] - nxt) / (60.0 / b)
                    nxt += over * 60.0 / b
                    remaining -= over
                    ti += 1
                    b = bpm()
                nxt += remaining * 60.0 / b
            if ts and si == 0:
                cur = ts[si][2]
                if cur > beats[-1] and close(nxt, cur):
                    nxt = cur
            if si < len(ts) - 1:
                if close(nxt, ts[si + 1][2]):
                    nxt = ts[si + 1][2]
                    si += 1
                    b = bpm()
            beats.append(nxt)
        return np.array(beats[:-1])

    def get_downbeats(self, start_time: float = 0.0) -> np.ndarray:
        beats = self.get_beats(start_time)
        ts = sorted(self.time_signature_changes, key=lambda t: t[2])
        if not ts or ts[0][2] > start_time:
            ts.insert(0, (4, 4, start_time))

        def index(arr, value, default):
            idx = np.flatnonzero(np.isclose(arr, value))
            return idx[0] if idx.size > 0 else default

        def stride(num):
            return num // 3 if (num % 3 == 0 and num != 3) else num

        out, end_i = [], 0
        for a, b in zip(ts[:-1], ts[1:]):
            start_i = index(beats, a[2], 0)
            end_i = index(beats, b[2], start_i)
            out.append(beats[start_i:end_i:stride(a[0])])
        start_i = index(beats, ts[-1][2], end_i)
        out.append(beats[start_i::stride(ts[-1][0])])
        d = np.concatenate(out)
        return d[d >= start_time]


def _qpm_to_bpm(qpm: float, numerator: int, denominator: int) -> float:
    if denominator in (1, 2, 4, 8, 16, 32):
        if numerator == 3:
            return qpm * denominator / 4.0
        if numerator % 3 == 0:
            return qpm / 3.0 * denominator / 4.0
        return qpm * denominator / 4.0
    return qpm


# ---------------------------------------------------------------------------------------------------------------------------
# the extractor
# ---------------------------------------------------------------------------------------------------------------------------

QUALITIES = OrderedDict([          # chord_class.py:5-39 (semitone templates, root first)
    ("maj", [1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0]), ("min", [1, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0]),
    ("aug", [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]), ("dim", [1, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0]),
    ("sus4", [1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0]), ("sus4(b7)", [1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 1, 0]),
    ("sus4(b7,9)", [1, 0, 1, 0, 0, 1, 0, 1, 0, 0, 1, 0]), ("sus2", [1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0]),
    ("7", [1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0]), ("maj7", [1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 1]),
    ("min7", [1, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0]), ("minmaj7", [1, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1]),
    ("maj6", [1, 0, 0, 0, 1, 0, 0, 1, 0, 1, 0, 0]), ("min6", [1, 0, 0, 1, 0, 0, 0, 1, 0, 1, 0, 0]),
    ("9", [1, 0, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0]), ("maj9", [1, 0, 1, 0, 1, 0, 0, 1, 0, 0, 0, 1]),
    ("min9", [1, 0, 1, 1, 0, 0, 0, 1, 0, 0, 1, 0]), ("7(#9)", [1, 0, 0, 1, 1, 0, 0, 1, 0, 0, 1, 0]),
    ("maj6(9)", [1, 0, 1, 0, 1, 0, 0, 1, 0, 1, 0, 0]), ("min6(9)", [1, 0, 1, 1, 0, 0, 0, 1, 0, 1, 0, 0]),
    ("maj(9)", [1, 0, 1, 0, 1, 0, 0, 1, 0, 0, 0, 0]), ("min(9)", [1, 0, 1, 1, 0, 0, 0, 1, 0, 0, 0, 0]),
    ("maj(11)", [1, 0, 0, 0, 1, 1, 0, 1, 0, 0, 0, 1]), ("min(11)", [1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 0, 1]),
    ("11", [1, 0, 1, 0, 1, 1, 0, 1, 0, 0, 1, 0]), ("maj9(11)", [1, 0, 1, 0, 1, 1, 0, 1, 0, 0, 0, 1]),
    ("min11", [1, 0, 1, 1, 0, 1, 0, 1, 0, 0, 1, 0]), ("13", [1, 0, 1, 0, 1, 1, 0, 1, 0, 1, 1, 0]),
    ("maj13", [1, 0, 1, 0, 1, 1, 0, 1, 0, 1, 0, 1]), ("min13", [1, 0, 1, 1, 0, 1, 0, 1, 0, 1, 1, 0]),
    ("dim7", [1, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0]), ("hdim7", [1, 0, 0, 1, 0, 0, 1, 0, 0, 0, 1, 0]),
])
INVERSIONS = {"maj": ["3", "5"], "min": ["b3", "5"], "7": ["3", "5", "b7"], "maj7": ["3", "5", "7"], "min7": ["5", "b7"]}
NUM_TO_ABS_SCALE = ["C", "C#", "D", "Eb", "E", "F", "F#", "G", "Ab", "A", "Bb", "B"]
NUM_TO_INVERSION = ["1", "b2", "2", "b3", "3", "4", "b5", "5", "#5", "6", "b7", "7"]


class ChordClass:
    """chord_class.py:59-137 - vocabulary and vectorised template score (``batch_score``)."""

    def __init__(self):
        self.chord_list, chroma, bass = ["N"], [np.zeros(12, int)], [np.zeros(12, int)]
        unit = np.eye(12, dtype=int)[0]
        for i in range(12):
            for q, tpl in QUALITIES.items():
                t = np.roll(np.array(tpl), i)
                self.chord_list.append(f"{NUM_TO_ABS_SCALE[i]}:{q}")
                chroma.append(t)
                bass.append(np.roll(unit, i))
                for inv in INVERSIONS.get(q, ()):
                    self.chord_list.append(f"{NUM_TO_ABS_SCALE[i]}:{q}/{inv}")
                    chroma.append(t)
                    bass.append(np.roll(unit, i + NUM_TO_INVERSION.index(inv)))
        self.chroma_templates, self.bass_templates = np.array(chroma), np.array(bass)

    def batch_score(self, chromas: np.ndarray, basschromas: np.ndarray) -> np.ndarray:
        out = np.zeros((chromas.shape[0], len(self.chord_list)), dtype=np.float64)
        for i, c in enumerate(self.chord_list):
            if c == "N":
                out[:, i] = 0.2
                continue
            rc, rb = self.chroma_templates[i], self.bass_templates[i]
            out[:, i] = ((chromas[:, rc > 0].sum(axis=1) - chromas[:, rc == 0].sum(axis=1)) / (rc > 0).sum()
                         + 0.5 * basschromas[:, rb > 0].sum(axis=1) - (rc > 0).sum() * 0.1 - ("/" in c) * 0.05)
        return out


def is_percussive_channel(ins: Instrument) -> bool:   # extractors/midi_utilities.py:172-175
    return ins.is_drum or ins.program > 112


def thickness_and_bass_weights(midi: PrettyMIDI) -> np.ndarray:
    """extractors/rule_based_channel_reweight.py:6-52."""
    rolls = [i.get_piano_roll().T for i in midi.instruments if not is_percussive_channel(i)]

    def thickness(roll):
        chroma = np.zeros((roll.shape[0], 12))
        for n in range(12):
            chroma[:, n] = np.sum(roll[:, n::12], axis=1)
        t = (chroma > 0).sum(axis=1)
        return 0 if t.sum() == 0 else t[t > 0].mean()

    def bass_property(roll):
        r = np.argwhere(roll > 0)[:, 1]
        return (0.0, 1.0) if len(r) == 0 else (r.mean(), min(1.0, len(r) / len(roll)))

    thick = np.array([thickness(r) for r in rolls])
    bass = np.array([bass_property(r) for r in rolls])
    bass[bass[:, 1] < 0.2, 0] = 128
    w = 1 - np.exp(-(thick - 0.95))
    w /= w.max()
    w[np.argmin(bass[:, 0])] = 1.0
    return w


def beats_with_positions(midi: PrettyMIDI, extra_division: int = 2) -> np.ndarray:
    """main.py:33-50 / extractors/midi_utilities.py:12-35: [time, position in the bar (1 = downbeat)] per (sub-)beat."""
    beats = midi.get_beats()
    if extra_division > 1:
        interp = np.linspace(beats[:-1], beats[1:], extra_division + 1).T
        beats = np.append(interp[:, :-1].reshape(-1), interp[-1, -1])
    downbeats = midi.get_downbeats()
    j, pos, out = 0, -2, []
    for b in beats:
        if j < len(downbeats) and b == downbeats[j]:
            pos, j = 1, j + 1
        else:
            pos += 1
        assert pos > 0
        out.append([b, pos])
    assert j == len(downbeats)
    return np.array(out)


def recognize(midi: PrettyMIDI, half_beat_switch: bool = True) -> List[list]:
    """``process_chord(entry, extra_division=2)``: [[start, end, label], ...].

    Quirk kept: ``process_chord`` builds half-beat positions (main.py:33-50) and never uses them - ``ChordRecognition`` reads
    ``entry.beat``, which ``transcribe_cb1000_midi`` attached with ``MidiBeatExtractor``'s default ``div = 1`` (main.py:68): the
    frames of the whole recognition are BEATS.  (With half-beat frames the example splits into 124 segments instead of the 110 of
    ``example.out``.)"""
    beat = beats_with_positions(midi, 1)
    weights = thickness_and_bass_weights(midi)
    SUB = 8
    n = len(beat)
    onset = beat[:, 0].copy()
    offset, length = np.zeros(n), np.zeros(n)
    for i in range(n):
        offset[i] = beat[i, 0] + (beat[i, 0] - beat[i - 1, 0]) if i == n - 1 else beat[i + 1, 0]
        length[i] = beat[i + 1, 0] - beat[i, 0] if i < n - 1 else length[i - 1]
    chroma, bassc = np.zeros((n, 12)), np.zeros((n, 12))
    min_bass = np.full((n * SUB,), 259, dtype=int)

    def quantize(t):
        if t <= onset[0]:
            return 0.0
        if t >= offset[-1]:
            return n + 0.0
        b = np.searchsorted(onset, t, side="right") - 1
        return b + (t - onset[b]) / length[b]

    wi = 0
    for ins in midi.instruments:
        if is_percussive_channel(ins):
            continue
        for note in ins.notes:
            bs, be = quantize(note.start), quantize(note.end)
            lb, rb = int(np.floor(bs + 0.2)), int(np.ceil(be - 0.2))
            ls, rs = int(np.floor(bs * SUB + 0.2)), int(np.floor(be * SUB + 0.2))
            if rb < lb:
                rb = lb
            if rs > ls:
                min_bass[ls:rs] = np.minimum(min_bass[ls:rs], note.pitch)
            pc = note.pitch % 12
            for j in range(lb, rb):
                chroma[j][pc] = max(chroma[j][pc], (min(j + 1, be) - max(bs, j)) * weights[wi])
        wi += 1
    for i in range(SUB):
        terms = min_bass[i::SUB]
        ok = terms < 259
        bassc[ok, terms[ok] % 12] += 1.0 / SUB
    is_down = beat[:, 1] == 1
    is_halfdown = beat[:, 1] * 2 - 2 == beat[:, 1].max()
    is_even = beat[:, 1] % 2 == 1

    # decode (midi_chord.py:115-190)
    cc = ChordClass()
    MAXP, ncls = 12, len(cc.chord_list)
    bc, bb = np.zeros((n, MAXP, 12)), np.zeros((n, MAXP, 12))
    for i in range(n):
        for j in range(min(MAXP, i + 1)):
            bc[i, j] = chroma[i - j:i + 1].sum(axis=0)
            bb[i, j] = bassc[i - j:i + 1].sum(axis=0)
    score = cc.batch_score(bc.reshape(-1, 12), bb.reshape(-1, 12)).reshape(n, MAXP, ncls)
    obs = np.full((n, MAXP, ncls), -np.inf)
    for i in range(n):
        for j in range(min(MAXP, i + 1)):
            obs[i, j] = score[i, j] + j * 0.7 + is_halfdown[i - j] * 0.15 + is_even[i - j] * 0.2
    dp = np.full(n, -np.inf)
    prec, prei = np.zeros(n, dtype=int), np.zeros(n, dtype=int)
    for i in range(n):
        for j in range(MAXP):
            if i - j < 0:
                continue
            best = int(np.argmax(obs[i, j]))
            prev = 0.0 if i - j == 0 else dp[i - j - 1]
            if dp[i] < prev + obs[i, j, best]:
                dp[i], prec[i], prei[i] = prev + obs[i, j, best], best, i - j - 1
            if j > 0 and is_down[i - j + 1]:
                break
    out, cur = [], n - 1
    while cur >= 0:
        pi, pc_ = prei[cur], prec[cur]
        start = pi + 1 if half_beat_switch or is_even[pi + 1] else pi + 2
        end = cur if half_beat_switch or cur == n - 1 or is_even[cur + 1] else cur + 1
        out.append([onset[start], offset[end], cc.chord_list[pc_]])
        cur = pi
    return out[::-1]


def transcribe_midi(midi_path: str, output_path: str = None) -> List[list]:
    """``transcribe_cb1000_midi`` (main.py:60-71): chord lines of a MIDI file, optionally written in the lab format."""
    result = recognize(PrettyMIDI(midi_path))
    if output_path is not None:
        with open(output_path, "w") as f:
            for row in result:
                f.write("\t".join(str(v) for v in row) + "\n")
    return result


# ---------------------------------------------------------------------------------------------------------------------------
# chord label -> the 14-number beat rows the chord encoder reads
# ---------------------------------------------------------------------------------------------------------------------------

_PITCH = {"C": 0, "D": 2, "E": 4, "F": 5, "G": 7, "A": 9, "B": 11}
_DEGREES = dict(zip("1 2 3 4 5 6 7 8 9 10 11 12 13".split(), [0, 2, 4, 5, 7, 9, 11, 12, 14, 16, 17, 19, 21]))
_T7, _TM7, _Tm7 = [1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0], [1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 1], [1, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0]
# mir_eval.chord.QUALITIES (the copy vendored in the reference, mir_eval/chord.py:243-273): shorthand -> semitones relative to the root.
# Without reduce_extended_chords (the default, and what the reference calls) a 9th / 11th / 13th shorthand is its seventh chord.
_MIR_QUALITIES = {
    "maj": [1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0], "min": [1, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0], "aug": [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0],
    "dim": [1, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0], "sus4": [1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0], "sus2": [1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0],
    "7": _T7, "maj7": _TM7, "min7": _Tm7, "minmaj7": [1, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1], "maj6": [1, 0, 0, 0, 1, 0, 0, 1, 0, 1, 0, 0],
    "min6": [1, 0, 0, 1, 0, 0, 0, 1, 0, 1, 0, 0], "dim7": [1, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0], "hdim7": [1, 0, 0, 1, 0, 0, 1, 0, 0, 0, 1, 0],
    "maj9": _TM7, "min9": _Tm7, "9": _T7, "b9": _T7, "#9": _T7, "min11": _Tm7, "11": _T7, "#11": _T7, "maj13": _TM7, "min13": _Tm7, "13": _T7,
    "b13": _T7, "1": [1] + [0] * 11, "5": [1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0], "": [0] * 12,
}
# the label grammar of Harte et al. 2005 as mir_eval validates it (mir_eval/chord.py:338-357): root, optional ":" + shorthand and / or
# a parenthesised list of (possibly omitted "*") scale degrees, optional "/" + bass degree
_DEG = r"(?:b*|#*)(?:[1-9]|1[0-3]?)"
_LIST = rf"\(\*?{_DEG}(?:,\*?{_DEG})*\)"
_SHORT = "maj|min|dim|aug|1|5|sus2|sus4|maj6|min6|7|maj7|min7|dim7|hdim7|minmaj7|aug7|9|maj9|min9|11|maj11|min11|13|maj13|min13"
_LABEL = __import__("re").compile(rf"^(?:N|X|[A-G](?:b*|#*)(?:(?::(?:{_SHORT})(?:{_LIST})?)|(?::{_LIST}))?(?:/{_DEG})?)$")


class InvalidChordException(Exception):
    pass


def _degree_semitone(deg: str) -> int:   # mir_eval scale_degree_to_semitone: accidentals only in front
    off = 0
    if deg.startswith("#"):
        off, deg = deg.count("#"), deg.strip("#")
    elif deg.startswith("b"):
        off, deg = -deg.count("b"), deg.strip("b")
    if deg not in _DEGREES:
        raise InvalidChordException(f"Scale degree improperly formed: {deg}")
    return _DEGREES[deg] + off


def chord_encode(label: str):
    """``mir_eval.chord.encode(label)`` with its defaults (mir_eval/chord.py:360-520 of the reference's vendored copy): (root pitch
    class, 12 semitones relative to the root, bass interval relative to the root).  ``N`` -> (-1, zeros, -1), ``X`` -> (-1, -1s, -1).
    Degrees beyond the octave in a parenthesised list fall outside the 12-entry bitmap and are dropped, like there."""
    label = str(label)
    if label == "N":
        return -1, np.zeros(12, dtype=int), -1
    if label == "X":
        return -1, np.full(12, -1, dtype=int), -1
    if not _LABEL.match(label):
        raise InvalidChordException(f"Invalid chord label: {label}")
    bass = "1"
    if "/" in label:
        label, bass = label.split("/")
    degrees = set()
    if "(" in label:
        label, inner = label.split("(")
        if "*" in inner and ":" not in label:
            raise InvalidChordException("Intervals specifying omissions MUST have a quality.")
        degrees = {d.strip() for d in inner.strip(")").split(",")}
    quality = "" if degrees else "maj"
    root = label
    if ":" in label:
        root, qname = label.split(":")
        if qname:
            quality = qname.lower()
    if quality not in _MIR_QUALITIES:
        raise InvalidChordException(f"Unsupported chord quality shorthand: '{quality}'")
    pc = _PITCH[root[0]] + root[1:].count("#") - root[1:].count("b")
    bass_number = _degree_semitone(bass) % 12
    bitmap = np.array(_MIR_QUALITIES[quality], dtype=int)
    bitmap[0] = 1
    for deg in degrees:
        sign = -1 if deg.startswith("*") else 1
        idx = _degree_semitone(deg.strip("*"))
        if idx < 12:
            bitmap[idx % 12] += sign
    bitmap = (bitmap > 0).astype(int)
    bitmap[bass_number] = 1
    return pc % 12, bitmap, bass_number


def chord_matrix_from_labels(rows, one_beat: float = 0.5) -> np.ndarray:
    """``data/midi_to_data.py:88-120`` ``get_chord_matrix``: one 14-number row [root, absolute chroma (12), absolute bass] per beat
    until each label's (rounded) end beat."""
    beat_cnt, chords = 0, []
    for start, end, label in rows:
        root, bitmap, bass = chord_encode(label)
        line = [root] + list(np.roll(bitmap, root)) + [(bass + root) % 12]
        while beat_cnt < int(round(float(end) / one_beat)):
            beat_cnt += 1
            chords.append(line)
    return np.array(chords)


def read_chord_lab(path: str):
    rows = []
    for line in open(path):
        tok = line.strip().split("\t")
        if len(tok) == 3:
            rows.append([float(tok[0]), float(tok[1]), tok[2]])
    return rows
