#!/usr/bin/env python
"""Experiment: does running the batch as S independent sub-batches on S HIP streams (kernels of different sub-batches in
different phases on the same CUs) beat one B=16 launch sequence?  Samples are independent, so this is legal on the path.
usage: python tools/exp_dual_stream.py [total_batch=16]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyffusion_amd.arch import UNetConfig  # noqa: E402
from polyffusion_amd.unet import UNetModel  # noqa: E402
from polyffusion_amd.weights import synth_unet_state  # noqa: E402


def make():
    m = UNetModel(in_channels=2, out_channels=2, channels=64, n_res_blocks=2, attention_levels=(2, 3),
                  channel_multipliers=(1, 2, 4, 4), n_heads=4, tf_layers=1, d_cond=512)
    return m


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    cfg = UNetConfig(d_cond=512)
    state = synth_unet_state(cfg, 0)
    base = make().load_state_dict(state).set_precision("bf16x3")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 2, 128, 128, generator=g).cuda()
    t = torch.randint(0, 1000, (B,), generator=g).cuda()
    c = torch.randn(B, 1, 512, generator=g).cuda()

    def timeit(fn, iters=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    ref = base(x, t, c).clone()
    ms = timeit(lambda: base(x, t, c))
    print(f"1 stream  x B={B}: {ms:.3f} ms/eval  ({1e3 / ms:.1f} evals/s)")
    for S in (2, 4):
        if B % S:
            continue
        n = B // S
        models = [make() for _ in range(S)]
        for m in models:
            m.bind_packed(base._blob_dev)
            m.set_precision("bf16x3")
        streams = [torch.cuda.Stream() for _ in range(S)]
        xs, ts, cs = x.chunk(S), t.chunk(S), c.chunk(S)
        outs = [torch.empty(n, 2, 128, 128, device="cuda") for _ in range(S)]
        ev_in = torch.cuda.Event()
        evs = [torch.cuda.Event() for _ in range(S)]

        def run():
            ev_in.record()
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    streams[i].wait_event(ev_in)
                    models[i](xs[i], ts[i], cs[i], out=outs[i])
                    evs[i].record()
            for i in range(S):
                torch.cuda.current_stream().wait_event(evs[i])

        run(); torch.cuda.synchronize()
        err = max((outs[i] - ref[i * n:(i + 1) * n]).abs().max().item() for i in range(S))
        ms = timeit(run)
        print(f"{S} streams x B={n}: {ms:.3f} ms/eval  ({1e3 / ms:.1f} evals/s)  max diff vs single {err:.2e}")
        # the same as ONE replayed hipGraph (fork / join inside the capture)
        gr = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        with torch.cuda.stream(cap):
            run()
            cap.synchronize()
            with torch.cuda.graph(gr, stream=cap):
                run()
        ms = timeit(gr.replay)
        print(f"{S} streams x B={n}, one hipGraph: {ms:.3f} ms/eval  ({1e3 / ms:.1f} evals/s)")
        # and the sub-batches back to back on ONE stream (what the split alone costs)
        def serial():
            for i in range(S):
                models[i](xs[i], ts[i], cs[i], out=outs[i])
        ms = timeit(serial)
        print(f"1 stream, {S} x B={n} back to back: {ms:.3f} ms/eval")


if __name__ == "__main__":
    main()
