"""Batched evaluation drivers - mirror of the reference's ``polyffusion/expr.py`` (call stack 3.3).

Same five drivers, same way of driving ``Experiments`` (``expr.py:11-122``), including the reference's habit of passing the
image batch as the "condition" where ``uncond_scale == 0`` makes the condition irrelevant.  The one difference: the reference
pulls its batches from ``get_val_dataloader(16)`` (POP909 validation split; dataset loaders are outside the rebuilt path), here
the caller passes any iterable of ``(prmat2c, pnotree, chord, prmat)`` batches - e.g. ``song_batches()`` over quantised songs in
the reference's data-dictionary format (``datasample.DataSample``), or synthetic ones.
"""
from __future__ import annotations

import os
from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import midi
from .inference_sdf import Experiments

Batch = Tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor], Optional[torch.Tensor]]


def song_batches(song_npz_paths: Sequence[str], batch_size: int = 16, device="cuda") -> Iterator[Batch]:
    """8-bar segments of quantised songs, ``batch_size`` at a time, in the tuple layout of the reference's dataloader."""
    from .datasample import DataSample
    buf: List[Tuple[torch.Tensor, ...]] = []
    for path in song_npz_paths:
        p2c, _, chd, prmat = DataSample.from_npz(path).get_whole_song_data()
        for i in range(p2c.shape[0]):
            buf.append((p2c[i], chd[i], prmat[i]))
            if len(buf) == batch_size:
                yield _stack(buf, device)
                buf = []
    if buf:
        yield _stack(buf, device)


def _stack(buf, device) -> Batch:
    p2c, chd, prmat = (torch.stack([b[i] for b in buf]).to(device) for i in range(3))
    return p2c, None, chd, prmat


def _take(batches: Iterable[Batch], num: int) -> Iterator[Batch]:
    for i, batch in enumerate(batches):
        if i >= num:
            break
        yield batch


def prompt_generation(expr: Experiments, batches: Iterable[Batch], num: int, output_dir: str, check_integrity: bool = True):
    """expr.py:11-30 - unconditional generation, one batch-sized run per validation batch."""
    gen = [expr.predict(prmat2c, None, 0.0, False) for prmat2c, _, _, _ in _take(batches, num)]
    gen = torch.cat(gen)
    if check_integrity:
        print(midi.check_prmat2c_integrity(gen))
    os.makedirs(output_dir, exist_ok=True)
    midi.prmat2c_to_midi_file(gen, f"{output_dir}/uncond.mid")
    return gen


def acc_arrangement(expr: Experiments, batches: Iterable[Batch], num: int, output_dir: str):
    """expr.py:33-49 - accompaniment arrangement: inpaint everything below the melody."""
    gen = [expr.inpaint(prmat2c, "below", prmat2c, None, uncond_scale=0.0, no_output=True) for prmat2c, _, _, _ in _take(batches, num)]
    gen = torch.cat(gen)
    os.makedirs(output_dir, exist_ok=True)
    midi.prmat2c_to_midi_file(gen, f"{output_dir}/acc_arr.mid")
    return gen


def inpaint_bars(expr: Experiments, batches: Iterable[Batch], num: int, output_dir: str, bar_list=(2, 3, 4, 5)):
    """expr.py:52-73 - regenerate bars 2-5 of every segment; only the inpainted steps [32, 96) are written."""
    gen = []
    for prmat2c, _, _, _ in _take(batches, num):
        x0 = expr.inpaint(prmat2c, "bars", prmat2c, None, uncond_scale=0.0, bar_list=list(bar_list), no_output=True)
        gen.append(x0[:, :, 32:96, :])
    gen = torch.cat(gen)
    os.makedirs(output_dir, exist_ok=True)
    midi.prmat2c_to_midi_file(gen, f"{output_dir}/inp_bars.mid")
    return gen


def chd_conditioning(expr: Experiments, model, batches: Iterable[Batch], num: int, output_dir: str, uncond_scale: float = 1.0):
    """expr.py:76-96 - chord-conditioned generation; the chords go to ``chd[scale].npy`` next to the MIDI file."""
    gen, chd = [], []
    for _, _, chord, _ in _take(batches, num):
        gen.append(expr.generate(model._encode_chord(chord), None, uncond_scale, no_output=True))
        chd.append(chord)
    gen = torch.cat(gen)
    os.makedirs(output_dir, exist_ok=True)
    np.save(f"{output_dir}/chd[{uncond_scale}].npy", torch.stack(chd).cpu().numpy())
    midi.prmat2c_to_midi_file(gen, f"{output_dir}/chd_cond[{uncond_scale}].mid")
    return gen


def txt_conditioning(expr: Experiments, model, batches: Iterable[Batch], num: int, output_dir: str, uncond_scale: float = 1.0):
    """expr.py:99-122 - texture-conditioned generation; the originals are written beside the result."""
    gen, orig = [], []
    for prmat2c, _, _, prmat in _take(batches, num):
        gen.append(expr.generate(model._encode_txt(prmat), None, uncond_scale, no_output=True))
        orig.append(prmat2c)
    gen, orig = torch.cat(gen), torch.cat(orig)
    os.makedirs(output_dir, exist_ok=True)
    midi.prmat2c_to_midi_file(gen, f"{output_dir}/txt_cond[{uncond_scale}].mid")
    midi.prmat2c_to_midi_file(orig, f"{output_dir}/txt_orig[{uncond_scale}].mid")
    return gen


DRIVERS = {"uncond": prompt_generation, "inp_below": acc_arrangement, "inp_bars": inpaint_bars,
           "chd_cond": chd_conditioning, "txt_cond": txt_conditioning}
