#!/usr/bin/env python
"""Shard invariance at EQUAL per-GPU batch: the images of a global batch must not depend on how many GPUs made them.

    torchrun --nproc-per-node N tools/shard_invariance.py --per_rank 16 --out FILE     N ranks, 16 samples each (config 4's shape per GPU)
    python tools/shard_invariance.py --chunks N --per_rank 16 --out FILE               ONE process, the same 16 N samples in chunks of 16

Both load the sdf_txt model through ``load_model`` (rank 0 packs the synthetic weights, the blobs travel by one broadcast), encode the
global batch's textures, run the last ``--steps`` DDPM reverse steps of ``Experiments.predict`` on each block of 16 with the noise generator
keyed by the GLOBAL sample index (``sample_offset``), and write the [16 N, 2, 128, 128] images (rank 0 gathers).  With the same per-launch
batch every kernel picks the same tiles, so the two files must be bit-identical (tests/test_gpu_multirank.py)."""
import argparse
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyffusion_amd import dist as pfdist, synth  # noqa: E402
from polyffusion_amd.inference_sdf import Experiments, load_model  # noqa: E402
from polyffusion_amd.params import preset  # noqa: E402
from polyffusion_amd.sampler import SDFSampler  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--per_rank", type=int, default=16)
    ap.add_argument("--chunks", type=int, default=0, help="single process: number of blocks of --per_rank samples (= the rank count it stands in for)")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--model", default="sdf_txt")
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    rank, world, _ = pfdist.init_from_env()
    blocks = a.chunks if world == 1 else world
    assert blocks >= 1 and (world == 1 or a.chunks == 0)
    total = blocks * a.per_rank
    params = preset(a.model)
    model = load_model(params, types.SimpleNamespace(synthetic_weights=True, precision=a.precision, chkpt_path=None, chkpt_name=None), rank, world)
    model.ldm.eps_model.set_precision(a.precision)
    prmat = torch.from_numpy(synth.prmat(total, 4040)).cuda()          # the GLOBAL batch's textures; every block takes its rows
    mine = range(blocks) if world == 1 else [rank]
    outs = []
    for blk in mine:
        lo = blk * a.per_rank
        cond = model._encode_txt(prmat[lo: lo + a.per_rank].contiguous())
        s = SDFSampler(model.ldm, seed=a.seed, sample_offset=lo)
        ex = Experiments(a.model, dict(params, n_steps=a.steps), s)
        outs.append(ex.predict(cond, uncond_scale=1.0))
    local = torch.cat(outs, 0)
    full = local if world == 1 else pfdist.gather_rows(local, total, rank, world)
    if rank == 0:
        np.save(a.out, full.cpu().numpy())
        print(f"shard_invariance: {tuple(full.shape)} from {blocks} block(s) of {a.per_rank} on {world} rank(s) -> {a.out}", flush=True)
    pfdist.barrier()


if __name__ == "__main__":
    main()
