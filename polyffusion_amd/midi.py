"""Output step of the generation path: generated onset/sustain images -> notes -> standard MIDI file.

Mirror of the reference's ``utils.prmat2c_to_prmat`` (ref:utils.py:240-269) and ``utils.prmat2c_to_midi_file``
(ref:utils.py:433-485), same names, arguments and note semantics.  The reference walks B x steps x 128 cells in Python
and delegates the file to pretty_midi; here the note extraction is one HIP kernel (``pf_prmat2c_durations``) on the image
where the sampler left it, and the file is written by a small SMF writer with pretty_midi's defaults (format 1,
220 ticks per beat, 120 bpm, 4/4, program 0 "Acoustic Grand Piano", note-off as note-on with velocity 0), so one 1/8 s
step is exactly 55 ticks.  No CPU fallback: without libpfhip.so / a GPU these functions raise.
"""
from __future__ import annotations

import struct
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

RESOLUTION = 220          # pretty_midi.PrettyMIDI() default ticks per quarter note
TEMPO_US = 500000         # 120 bpm
TICKS_PER_SECOND = RESOLUTION * 1_000_000 // TEMPO_US   # 440
VELOCITY = 80             # ref:utils.py:467

Note = Tuple[int, float, float]   # (pitch, start seconds, end seconds)


# ---------------------------------------------------------------------------------------------- device side
def durations(prmat2c, is_custom_round: bool = False) -> torch.Tensor:
    """``[N, 2, S, 128]`` image (torch tensor on the GPU, or anything convertible) -> ``[N, S, 128]`` int32 on the GPU:
    the length in steps of the note starting at each cell, 0 where there is no onset."""
    _lib.require_gpu()
    lib = _lib.load()
    x = torch.as_tensor(np.asarray(prmat2c) if not isinstance(prmat2c, torch.Tensor) else prmat2c)
    if x.dim() != 4 or x.shape[1] != 2 or x.shape[3] != 128:
        raise RuntimeError(f"prmat2c must be [N, 2, steps, 128], got {tuple(x.shape)}")
    x = x.detach().to(device="cuda", dtype=torch.float32).contiguous()
    dur = torch.empty(x.shape[0], x.shape[2], 128, dtype=torch.int32, device=x.device)
    _lib.check(lib.pf_prmat2c_durations(x.data_ptr(), x.shape[0], x.shape[2], int(bool(is_custom_round)), dur.data_ptr(),
                                        _lib.current_stream()), "pf_prmat2c_durations")
    return dur


def prmat2c_to_prmat(prmat2c, n_step: int = 32) -> np.ndarray:
    """ref:utils.py:240-269 - ``(N, 2, 32*ratio, 128)`` -> ``(N*ratio, 32, 128)`` int64 durations at onsets."""
    d = durations(prmat2c)
    n, s, k = d.shape
    return d.cpu().numpy().astype(np.int64).reshape(n * (s // n_step), n_step, k)


def note_lists(prmat2c, inp_mask=None, is_custom_round: bool = False) -> Tuple[List[Note], List[Note]]:
    """The two note lists ("origin", "inpainted") the reference hands to pretty_midi (ref:utils.py:447-476), in its
    append order (bar, step, key).  A bar lasts S/8 seconds; a note is clipped to the end of its bar."""
    d = durations(prmat2c, is_custom_round).cpu().numpy()
    n, s, _ = d.shape
    t_bar = int(s / 8)
    b, step, key = np.nonzero(d)
    start = b * t_bar + step / 8
    end = np.minimum(b * t_bar + (step + d[b, step, key]) / 8, b * t_bar + t_bar)
    inp = np.zeros(len(b), dtype=bool)
    if inp_mask is not None:
        m = inp_mask.detach().cpu().numpy() if isinstance(inp_mask, torch.Tensor) else np.asarray(inp_mask)
        inp = m[b, 0, step, key] == 0.0
    notes = list(zip(key.tolist(), start.tolist(), end.tolist()))
    return [nt for nt, i in zip(notes, inp) if not i], [nt for nt, i in zip(notes, inp) if i]


def prmat2c_to_midi_file(prmat2c, fpath: str, labels: Optional[Sequence[str]] = None, is_custom_round: bool = False,
                         inp_mask=None) -> None:
    """ref:utils.py:433-485 - write the generated bars as a MIDI file: one piano track, plus an "inpainted" track when
    ``inp_mask`` is given (cells whose mask is 0 were generated), plus one lyric per bar when ``labels`` is given."""
    origin, inpainted = note_lists(prmat2c, inp_mask, is_custom_round)
    steps = prmat2c.shape[2]
    t_bar = int(steps / 8)
    lyrics = [(str(lab), float(i * t_bar)) for i, lab in enumerate(labels)] if labels is not None else None
    write_smf(fpath, [origin, inpainted] if inp_mask is not None else [origin], lyrics)


# ---------------------------------------------------------------------------------------------- standard MIDI file
def _vlq(v: int) -> bytes:
    out = [v & 0x7F]
    v >>= 7
    while v:
        out.append(0x80 | (v & 0x7F))
        v >>= 7
    return bytes(reversed(out))


def _track(events: Iterable[Tuple[int, int, bytes]]) -> bytes:
    """events: (tick, order, message bytes).  Sorted by tick, then order (note-offs before note-ons on a shared tick)."""
    body, last = bytearray(), 0
    for tick, _order, msg in sorted(events, key=lambda e: (e[0], e[1])):
        body += _vlq(tick - last) + msg
        last = tick
    body += _vlq(1) + b"\xff\x2f\x00"
    return b"MTrk" + struct.pack(">I", len(body)) + bytes(body)


def _ticks(seconds: float) -> int:
    return int(round(seconds * TICKS_PER_SECOND))


def write_smf(path: str, tracks: Sequence[Sequence[Note]], lyrics: Optional[Sequence[Tuple[str, float]]] = None) -> None:
    """Format-1 file: a timing track (4/4, 120 bpm, lyrics) and one piano track per note list."""
    timing = [(0, 0, b"\xff\x58\x04\x04\x02\x18\x08"), (0, 1, b"\xff\x51\x03" + TEMPO_US.to_bytes(3, "big"))]
    for text, t in lyrics or ():
        raw = text.encode("utf-8")
        timing.append((_ticks(t), 2, b"\xff\x05" + _vlq(len(raw)) + raw))
    chunks = [_track(timing)]
    for i, notes in enumerate(tracks):
        ch = i if i < 9 else i + 1          # channel 9 is percussion
        if ch > 15:
            raise RuntimeError("write_smf: at most 15 melodic tracks")
        ev = [(0, 0, bytes([0xC0 | ch, 0]))]
        for pitch, start, end in notes:
            if not 0 <= int(pitch) <= 127:
                raise RuntimeError(f"write_smf: pitch {pitch} out of range")
            ev.append((_ticks(start), 2, bytes([0x90 | ch, int(pitch), VELOCITY])))
            ev.append((_ticks(end), 1, bytes([0x90 | ch, int(pitch), 0])))
        chunks.append(_track(ev))
    with open(path, "wb") as f:
        f.write(b"MThd" + struct.pack(">IHHH", 6, 1, len(chunks), RESOLUTION))
        for c in chunks:
            f.write(c)


def read_smf(path: str):
    """Minimal reader for the files ``write_smf`` produces (tests, round trips): returns
    ``(tracks, lyrics, division, tempo)`` with tracks = per instrument track a list of (pitch, start tick, end tick)."""
    data = open(path, "rb").read()
    if data[:4] != b"MThd":
        raise RuntimeError("not a standard MIDI file")
    _hlen, _fmt, ntrk, division = struct.unpack(">IHHH", data[4:14])
    pos, tracks, lyrics, tempo = 14, [], [], None

    def vlq(p):
        v = 0
        while True:
            b = data[p]
            p += 1
            v = (v << 7) | (b & 0x7F)
            if not b & 0x80:
                return v, p

    for ti in range(ntrk):
        if data[pos:pos + 4] != b"MTrk":
            raise RuntimeError("bad track chunk")
        end = pos + 8 + struct.unpack(">I", data[pos + 4:pos + 8])[0]
        p, tick, open_notes, notes = pos + 8, 0, {}, []
        while p < end:
            dt, p = vlq(p)
            tick += dt
            status = data[p]
            if status == 0xFF:
                kind = data[p + 1]
                ln, q = vlq(p + 2)
                payload = data[q:q + ln]
                p = q + ln
                if kind == 0x51:
                    tempo = int.from_bytes(payload, "big")
                elif kind == 0x05:
                    lyrics.append((payload.decode("utf-8"), tick / TICKS_PER_SECOND))
            elif status & 0xF0 == 0xC0:
                p += 2
            elif status & 0xF0 in (0x90, 0x80):
                pitch, vel = data[p + 1], data[p + 2]
                p += 3
                if status & 0xF0 == 0x90 and vel > 0:
                    open_notes.setdefault(pitch, []).append(tick)
                else:
                    notes.append((pitch, open_notes[pitch].pop(0), tick))     # first-in first-out, like pretty_midi
            else:
                raise RuntimeError(f"unexpected status byte {status:#x}")
        if ti > 0:
            tracks.append(notes)
        pos = end
    return tracks, lyrics, division, tempo


def check_prmat2c_integrity(prmat2c, is_custom_round: bool = False) -> float:
    """ref:utils.py:402-430 - fraction of "orphan" sustain cells: a cell whose sustain rounds to > 0 although neither the onset nor
    the sustain of the same key rounded to > 0 one step earlier (or it sits in step 0) counts as an error AND as a note; every
    onset that rounds to > 0 counts as a note; the result is errors / notes.  ``int(round(v)) > 0`` on a float32 is ``v > 0.5``
    (round-half-even), ``custom_round`` is ``0.95 < v < 1.05`` (ref:utils.py:395-399).  Vectorised; runs where the tensor lives."""
    x = torch.as_tensor(np.asarray(prmat2c) if not isinstance(prmat2c, torch.Tensor) else prmat2c).detach().float()
    on = ((x[:, 0] > 0.95) & (x[:, 0] < 1.05)) if is_custom_round else (x[:, 0] > 0.5)
    sus = ((x[:, 1] > 0.95) & (x[:, 1] < 1.05)) if is_custom_round else (x[:, 1] > 0.5)
    # "previous cell empty" is `int(round(v)) == 0`, i.e. -0.5 <= v <= 0.5 under round-half-even: a NEGATIVE overshoot below -0.5
    # rounds to -1 and counts as occupied upstream (custom_round only yields 0 / 1, so there it is just "not 1")
    occ = (lambda v, r: r) if is_custom_round else (lambda v, r: (v > 0.5) | (v < -0.5))
    prev = torch.zeros_like(sus)
    prev[:, 1:] = occ(x[:, 0], on)[:, :-1] | occ(x[:, 1], sus)[:, :-1]
    err = int((sus & ~prev).sum())
    total = err + int(on.sum())
    return float(err / total)
