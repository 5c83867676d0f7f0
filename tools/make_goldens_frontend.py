#!/usr/bin/env python
"""tests/golden/frontend.npz: chord-label encodings by the reference's VENDORED mir_eval (``mir_eval/chord.py`` ``encode``, what
``data/midi_to_data.py:88-120 get_chord_matrix`` calls) for the extractor's whole vocabulary plus hand-picked grammar cases, and the
beat-wise chord matrix of the reference's ``chord_extractor/example.out``.  (``tests/golden/chord_example.mid`` / ``.out`` are the
reference's own example input and expected output, copied as data.)  Build container only; needs /root/reference."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/polyffusion"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

EXTRA = ["C", "G:7/5", "D:maj(9)/3", "F#:min7/b7", "Bb:sus4(b7,9)", "E:(1,5)", "A:5", "Eb:maj(*3)", "C:1", "X", "Db:hdim7/b5", "B:min/5",
         "C:maj/9", "C:maj(#11)", "C:(*3)", "H:maj", "C:foo", "Cbb:min7(b1)", "C##:aug7", "G:maj11", "C:min(*b3,*5)/5", "A:(3)/6", "C:maj(13)",
         "D:7(b9,#11)/b7", "C/3", "c:maj", "C:Maj"]


def main():
    import mir_eval.chord as mc
    from polyffusion_amd.chord_extractor import ChordClass, read_chord_lab
    labels = ChordClass().chord_list + EXTRA
    enc = np.zeros((len(labels), 14), dtype=np.int64)      # root, bitmap[12], bass; invalid labels: root = -99
    for i, lab in enumerate(labels):
        try:
            r, b, s = mc.encode(lab)
            enc[i] = [r] + list(b) + [s]
        except Exception:
            enc[i, 0] = -99
    rows = read_chord_lab(os.path.join(REF, "chord_extractor", "example.out"))
    beat_cnt, chords = 0, []
    for _, end, lab in rows:          # data/midi_to_data.py:88-120, with the reference's mir_eval
        while beat_cnt < int(round(end / 0.5)):
            beat_cnt += 1
            r, b, s = mc.encode(lab)
            chords.append([r] + list(np.roll(b, r)) + [(s + r) % 12])
    out = os.path.join(REPO, "tests", "golden", "frontend.npz")
    np.savez_compressed(out, labels=np.array(labels), encodings=enc, example_chord_matrix=np.array(chords))
    print("frontend.npz", os.path.getsize(out) // 1024, "KiB;", len(labels), "labels,", int((enc[:, 0] == -99).sum()), "invalid;", len(chords), "beats")


if __name__ == "__main__":
    main()
