#!/usr/bin/env python
"""bench.py - denoising steps/sec on MI355X for BASELINE.json config 2
(sdf_chd8bar conditional generation, batch 16 per GPU, DDPM sampler).

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: re-executes itself under torch.distributed.run, one rank per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one iteration of SDFSampler.paint's loop on a batch of 16 piano rolls (`SDFSampler.repaint_step`, the method paint()
itself runs): the conditional UNet evaluation (uncond_scale = 1 -> one eval per sample) and the fused DDPM/RePaint update with its two
noise draws - exactly what `inference_sdf` executes per reverse step.  As in paint(), what the denoiser derives from (t, cond) alone is
prepared once before the loop (`sampler.prepare`).  Inputs (x_T, chord condition from the HIP chord encoder, weights) are resident in
HBM before the timed region.  Rank 0 prints ONE JSON line; `value` is the whole-job aggregate (sum over GPUs, weak scaling: every GPU
denoises its own batch of 16).

Timing protocol: W warm-up steps, then `--windows` (default 5) back-to-back windows of EXACTLY K steps, each between barrier +
synchronize pairs (host clock, max over ranks).  `value` / `ms_per_step` are the MEDIAN window; every window and the spread are in
the line (`windows_ms_per_step`), next to the average shader clock the GPU sustained over each window (`sclk_mhz`, from two probe
kernels that read the shader-cycle and the constant-rate counters) - the part is power-managed, and box-to-box differences of the same
binary show up there.

Extra objects (see DESIGN.md "Measurement"):
  roofline     - the dominant kernel family (the 3x3 convolutions, conv_bf16x3.hip: error-compensated bf16 split on the
                 bf16 matrix pipe, peak 2500/3 TFLOP/s fp32-equivalent): algorithmic FLOPs of its launches / their summed
                 duration, measured with hipEvents around every launch on the launch stream in a separate profiled pass.
  fp32_mode    - the same workload with exact fp32 MFMA (peak 157.3 TFLOP/s), printed beside the headline.
  f16x3_mode   - the same workload in the fp16-split build of the library (libpfhip_f16.so: three fp16 MFMAs per product, fp32-class
                 error), timed in windows alternating with bf16x3 windows, with both splits' distance to the f32 mode on one evaluation.
  cpu_baseline - the CPU oracle (same ATen ops as the reference) timed on this host's cores: one warm-up step, a thread sweep,
                 then the median of 3 steps at batch 16 with the best thread count; rank 0 and N == 1 only.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from polyffusion_amd import _lib, dist as pfdist, synth  # noqa: E402
from polyffusion_amd.arch import UNetConfig  # noqa: E402
from polyffusion_amd.inference_sdf import build_encoders, build_ldm, build_unet  # noqa: E402
from polyffusion_amd.model_sdf import Polyffusion_SDF  # noqa: E402
from polyffusion_amd.params import preset  # noqa: E402
from polyffusion_amd.sampler import SDFSampler  # noqa: E402
from polyffusion_amd.weights import synth_chord_encoder_state, synth_unet_state  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X dense bf16 matrix peak (not the 2:1-sparse marketing figure)
# algorithmic (fp32-equivalent) FLOP/s ceiling of each arithmetic mode: the bf16x3 split issues three bf16 MFMAs per product
PEAK_ALGO = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16x3": PEAK_BF16_MFMA_TFLOPS / 3.0}
F_MIN_PER_SAMPLE_EVAL = 89.338e9  # algorithmic FLOPs per UNet sample-eval, n_cond == 1 dead math removed
F_UPFOLD_SAVED = 6.040e9          # 5/9 of the three UpSample convs, not executed when they run parity-folded (bf16x3 mode)
BATCH = 16
KIND_NAMES = ["conv3x3_mfma", "gemm_mfma", "attention", "gn_stats", "ln_stats", "small"]


def build_model(params, rank, world, x3=None):
    """rank 0 generates + packs the weights; everyone else receives the packed blobs (RCCL broadcast).
    x3="f16": the denoiser lives in the fp16-split build of the library (f16x3_mode leg)."""
    unet = build_unet(params, x3=x3)
    chord_enc, _ = build_encoders(params)
    dev = torch.device("cuda", torch.cuda.current_device())
    if rank == 0:
        ublob = unet.pack_state_dict(synth_unet_state(UNetConfig.from_params(params), 0)).to(dev)
        cblob = chord_enc.pack_state_dict(synth_chord_encoder_state(0)).to(dev)
    else:
        ublob = torch.empty(unet.weight_bytes() // 4, dtype=torch.float32, device=dev)
        cblob = torch.empty(_lib.load().pf_encoder_weight_bytes(chord_enc._h) // 4, dtype=torch.float32, device=dev)
    t0 = time.perf_counter()
    pfdist.broadcast_blob(ublob, 0)
    pfdist.broadcast_blob(cblob, 0)
    torch.cuda.synchronize()
    bcast_s = time.perf_counter() - t0
    unet.bind_packed(ublob)
    chord_enc.bind_packed(cblob)
    model = Polyffusion_SDF(build_ldm(params, unet), params.cond_type, params.cond_mode, chord_enc=chord_enc)
    return model, bcast_s


def cpu_baseline(params, reps=3):
    """Oracle (the reference's ATen ops on the CPU) on a bounded sample of the same workload: single DDPM steps at batch 16.
    One full B=16 step warms up the thread pool and oneDNN's per-shape primitives, every thread count of the sweep is probed
    with one step, and the best count is timed `reps` more times; the value is 1 / median step time."""
    from oracle import sampler_ref, unet_ref
    cfg = UNetConfig.from_params(params)
    w = unet_ref.to_torch(synth_unet_state(cfg, 0))
    ncpu = os.cpu_count() or 1
    x0 = torch.from_numpy(synth.gaussian((BATCH, 2, 128, 128), 1234))
    c = torch.from_numpy(synth.gaussian((BATCH, 1, cfg.d_cond), 77))
    rng = np.random.Generator(np.random.PCG64(5))
    noise = lambda shape: torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
    model = lambda x_, t_, c_: unet_ref.unet_forward(w, cfg, x_, t_, c_)
    s = sampler_ref.SDFSamplerRef(model, 1000, params.linear_start, params.linear_end, noise_fn=noise)
    z = torch.zeros_like(x0)

    def one_step(step=999):
        t0 = time.perf_counter()
        x_kn = s.q_sample(z, step, noise(x0.shape))
        x_un, _, _ = s.p_sample(x0, c, step)
        _ = x_kn * z + x_un * (1 - z)
        return time.perf_counter() - t0

    default_threads = torch.get_num_threads()
    # oneDNN's convolutions stop scaling long before a 256-thread host is full (measured on the GPU boxes in rounds 3-6, every run: 16 threads
    # 0.28, 32: 0.29-0.33, 64: 0.20-0.21, 128: 0.10-0.11, all 256: 0.008 steps/s), so the sweep probes 16 and 32 only - the two larger counts
    # cost 30 s of a default run to confirm that they lose - and the best count is what gets reported (the whole leg: ~25 s of CPU work)
    sweep = sorted({n for n in (16, 32) if n <= ncpu}) or [ncpu]
    probes = {}
    with torch.no_grad():
        torch.set_num_threads(min(32, ncpu))
        one_step()                                   # warm-up: a full step on the same inputs
        for n in sweep:
            torch.set_num_threads(n)
            one_step()                               # primitives are cached per thread count
            probes[n] = one_step()
        best = min(probes, key=probes.get)
        torch.set_num_threads(best)
        times = sorted([probes[best]] + [one_step() for _ in range(reps - 1)])
        torch.set_num_threads(default_threads)
    med = times[len(times) // 2]
    return {"value": 1.0 / med, "unit": "steps/s", "cores": best, "kind": "port",
            "sample": f"median of {len(times)} single DDPM steps at batch {BATCH} (sdf_chd8bar) through oracle/unet_ref.py + sampler_ref.py "
                      f"after a warm-up step on the same inputs; {best} torch threads (best of the sweep) of {ncpu} host CPUs",
            "step_seconds": [round(t, 3) for t in times],
            "spread_pct": round((times[-1] - times[0]) / med * 100, 1),
            "thread_sweep_steps_per_s": {str(n): round(1.0 / t, 4) for n, t in probes.items()}}


def mfma_sustained(launches: int = 10, iters: int = 4000) -> dict:
    """What the bf16 matrix pipe sustains on THIS box at 100 % duty on operands with random signs / mantissas (pf_mfma_probe: 8 waves
    per CU, dependency-free MFMA streams).  The part is power-managed - such a stream clocks ~1.6 GHz where the nominal peak assumes
    2.4 - so this, not the nominal 2.5 PFLOP/s, is what a perfect main loop could reach here.  Median of the later launches (the first
    ones run before the power management has settled)."""
    lib = _lib.load()
    sink = torch.zeros(1, device="cuda")
    fl = C.c_double(0.0)
    evs = []
    for _ in range(launches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.pf_mfma_probe(sink.data_ptr(), iters, C.byref(fl), _lib.current_stream()), "pf_mfma_probe")
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in evs]
    tf = sorted(fl.value / (m * 1e-3) / 1e12 for m in ms[launches // 2:])
    med = tf[len(tf) // 2]
    return {"tflops": round(med, 1), "unit": "TFLOP/s raw bf16 (bf16x3 products: / 3)", "frac_of_nominal": round(med / PEAK_BF16_MFMA_TFLOPS, 4),
            "launch_ms": [round(m, 3) for m in ms],
            "how": "pf_mfma_probe: one 8-wave workgroup per CU, every wave a dependency-free v_mfma_f32_32x32x16_bf16 stream (100 % pipe duty) "
                   "on operands with random signs and mantissas; median rate of the later launches.  The nominal peak is never reached on "
                   "real data: the part clocks to its power budget.  (Register operands only: a loop that also streams its operands through the LDS "
                   "sustains less - tools/micro/tap_pingpong.hip: 1.65 PFLOP/s)"}


class ClockProbe:
    """Average shader clock over a stretch of stream work: pf_clock_probe before and after it.  The reference counter's rate is
    calibrated once against the host clock (two probes 50 ms apart) instead of being assumed."""

    def __init__(self, dev):
        self.lib = _lib.load()
        self.buf = torch.zeros(2, 8, 2, dtype=torch.int64, device=dev)   # [begin | end][XCD][shader cycles, reference ticks]
        self.ref_hz = None
        try:
            self._probe(0); torch.cuda.synchronize(); t0 = time.perf_counter()
            time.sleep(0.05)
            self._probe(1); torch.cuda.synchronize(); t1 = time.perf_counter()
            b = self.buf.tolist()
            rates = sorted((b[1][x][1] - b[0][x][1]) / (t1 - t0) for x in range(8) if b[0][x][1] and b[1][x][1])
            hz = rates[len(rates) // 2] if rates else 0.0
            self.ref_hz = hz if 1e6 < hz < 1e10 else None
        except Exception:
            self.ref_hz = None

    def _probe(self, slot):
        _lib.check(self.lib.pf_clock_probe(self.buf[slot].data_ptr(), _lib.current_stream()), "pf_clock_probe")

    def begin(self):
        self.buf.zero_()
        self._probe(0)

    def end_mhz(self):
        """call after the stream has been synchronised; per-XCD average MHz {"median", "min", "max"} or None when the counters are unusable"""
        self._probe(1); torch.cuda.synchronize()
        b = self.buf.tolist()
        if not self.ref_hz:
            return None
        mhz = sorted((b[1][x][0] - b[0][x][0]) / (b[1][x][1] - b[0][x][1]) * self.ref_hz / 1e6
                     for x in range(8) if b[0][x][1] and b[1][x][1] > b[0][x][1])
        if not mhz:
            return None
        return {"median": round(mhz[len(mhz) // 2], 1), "min": round(mhz[0], 1), "max": round(mhz[-1], 1), "xcds": len(mhz)}


def smi_snapshot():
    """sclk / power as rocm-smi reports them (outside the timed region); {} when the tool is absent or prints nothing usable"""
    import shutil, subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        out = subprocess.run([exe, "-d", str(torch.cuda.current_device()), "--showclocks", "--showpower", "--json"], capture_output=True,
                             text=True, timeout=20).stdout
        card = next(iter(json.loads(out).values()))
        keep = {k: v for k, v in card.items() if any(s in k.lower() for s in ("sclk", "power", "mclk"))}
        return dict(list(keep.items())[:6])
    except Exception:
        return {}


def measure_traffic(precision):
    """HBM bytes per launch of the 3x3 conv family, measured in THIS run: two child runs of this script (3 steps each) under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` - separate passes, counters never combined with another trace domain,
    as MI355X_MICROARCH.md's HBM section prescribes - and its corrections: both counters are in KB, FETCH_SIZE counts a 128-byte request as
    64 B on gfx950 (doubled here).  Returns (bytes per launch, description) or (None, why)."""
    import shutil, sqlite3, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    child = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--windows", "1", "--profile-steps", "0", "--no-cpu-baseline",
             "--small-batch-steps", "0", "--fp32-steps", "0", "--f16x3-steps", "0", "--no-pmc", "--precision", precision]
    pat = "conv_bf3_kernel<" if precision == "bf16x3" else "conv_mfma_kernel<"
    tot, launches = {}, 0
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            try:
                subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "--output-format", "rocpd", "-d", out, "-o", "bench", "--"] + child,
                               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, timeout=300, check=True)
                dbs = [os.path.join(r, f) for r, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
                db = sqlite3.connect(dbs[0])
                rows = db.execute("select name, sum(counter_value), count(distinct dispatch_id) from pmc_events where counter_name = ? group by name",
                                  (ctr,)).fetchall()
            except Exception as e:   # a box without counter access, a changed schema: fall back to the committed collection
                return None, f"{ctr} pass failed: {type(e).__name__}"
            fam = [(v, n) for name, v, n in rows if pat in name and (precision != "bf16x3" or name.split(pat)[1][0] in "23")]
            if not fam:
                return None, f"{ctr} pass: no {pat} dispatches in the counter database"
            tot[ctr] = sum(v for v, _ in fam)
            launches = sum(n for _, n in fam)
    per_launch = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / launches
    return per_launch, (f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over 3 steps of this "
                        f"workload each ({launches} launches of the 3x3 family per pass); KB units x 1024, FETCH_SIZE doubled (gfx950 tallies a 128-byte "
                        "request as 64 B, MI355X_MICROARCH.md HBM section)")


def timed_loop(step_fn, x, t_step, steps, probe=None):
    """K steps between barrier + synchronize pairs (the driver contract: host clock, max over ranks) with a HIP-event pair on the
    launch stream around the same K steps.  Returns (x, t_step, host seconds (max over ranks), event seconds,
    average shader clock in MHz or None)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); pfdist.barrier()
    t0 = time.perf_counter()
    e0.record()
    if probe:
        probe.begin()
    for _ in range(steps):
        x = step_fn(x, t_step); t_step = max(t_step - 1, 1)
    e1.record()
    torch.cuda.synchronize(); pfdist.barrier()
    host = pfdist.max_over_ranks(time.perf_counter() - t0)
    mhz = probe.end_mhz() if probe else None
    return x, t_step, host, e0.elapsed_time(e1) * 1e-3, mhz


def profiled_pass(unet, step_fn, x, t_step, n_steps, precision, dump=False):
    """hipEvents around EVERY launch (pf_unet_set_profiling): per-kernel-family ms per step and the 3x3 family's achieved rate."""
    unet.set_profiling(True)
    agg = {}
    for it in range(n_steps):
        x = step_fn(x, t_step)
        torch.cuda.synchronize()
        direct = unet.read_profile_direct()
        for i, (kind, ms, fl) in enumerate(unet.read_profile()):
            if dump and it == 0:
                print(f"launch {i:3d} {KIND_NAMES[kind]:14s} {ms * 1e3:8.1f} us {fl / 1e9:8.2f} GF {fl / max(ms, 1e-9) / 1e9:7.1f} TF/s"
                      + (f"   (Winograd: direct form {direct[i] / 1e9:.2f} GF)" if direct[i] != fl else ""), file=sys.stderr)
            a = agg.setdefault(kind, [0, 0.0, 0.0, 0.0, 0])
            a[0] += 1; a[1] += ms; a[2] += fl; a[3] += direct[i]; a[4] += int(direct[i] != fl)
    unet.set_profiling(False)
    return x, agg


def _median(vals):
    v = sorted(vals)
    return v[len(v) // 2]


def windowed(run_window, windows: int, probe=None):
    """`windows` back-to-back calls of run_window() (each a fixed number of steps, bracketed by synchronize + host clock inside this
    helper, with the clock probe around it) -> (median seconds, all seconds, clock of the median window).  The protocol of the headline,
    for the secondary lines."""
    recs = []
    for _ in range(max(1, windows)):
        torch.cuda.synchronize()
        if probe:
            probe.begin()
        t0 = time.perf_counter()
        run_window()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        recs.append((dt, probe.end_mhz() if probe else None))
    order = sorted(range(len(recs)), key=lambda i: recs[i][0])
    med = recs[order[len(order) // 2]]
    return med[0], [r[0] for r in recs], med[1]


def small_batch_lines(model, params, steps, precision, windows=3, probe=None):
    """Secondary lines: the launch-bound regime.  Batch 1 is what the reference CLI's --autoreg runs (one 8-bar segment per
    sampling run), batch 8 the per-GPU shape of BASELINE config 5 (8 songs per GPU).  Each is timed through
    SDFSampler.paint() - the RePaint loop body with orig/mask (first half known: the autoregressive overlap), exactly what
    Experiments.predict drives - eagerly and as replays of one captured hipGraph step (device-resident step state); median of
    `windows` paint() calls of `steps`+1 reverse steps each, with the fraction of the bf16x3 ceiling the denoiser reaches."""
    dev = torch.device("cuda", torch.cuda.current_device())
    f_eval = F_MIN_PER_SAMPLE_EVAL - (F_UPFOLD_SAVED if precision == "bf16x3" else 0.0)
    res = {}
    for B in (1, 8):
        chords = torch.from_numpy(synth.chords(B, seed=777)).to(dev)
        cond = model._encode_chord(chords)
        shape = (B, params.out_channels, params.img_h, params.img_w)
        z = torch.zeros(shape, device=dev)
        mask = torch.zeros(shape, device=dev)
        mask[:, :, : params.img_h // 2] = 1        # the autoregressive half-overlap: first half known
        line = {}
        for mode in ("eager", "graph"):
            s = SDFSampler(model.ldm, seed=3, graph=(mode == "graph"))
            x = s.randn(shape, dev)
            s.paint(x, cond, 3, orig=z, mask=mask)                       # warm-up (tile instantiation, workspace, graph capture)
            med, all_s, mhz = windowed(lambda: s.paint(x, cond, steps, orig=z, mask=mask), windows, probe)
            line[mode + "_steps_per_s"] = round((steps + 1) / med, 2)
            line[mode + "_ms_per_step"] = round(med / (steps + 1) * 1e3, 4)
            line[mode + "_windows_ms_per_step"] = [round(t / (steps + 1) * 1e3, 4) for t in all_s]
            line[mode + "_sclk_mhz"] = mhz
            if mode == "graph":
                e0, e1, n = s.last_replay
                line["graph_replay_ms_per_step"] = round(e0.elapsed_time(e1) / n, 4)
        best = min(line["eager_ms_per_step"], line["graph_ms_per_step"])
        line["roofline_frac"] = round(f_eval * B / (best * 1e-3) / (PEAK_ALGO[precision] * 1e12), 4)
        line["roofline_note"] = "whole denoiser path: executed FLOPs per step / the faster of the two step times / the mode's matrix-pipe ceiling"
        res[f"batch{B}"] = line
    res["precision"] = precision
    res["steps_per_window"] = steps + 1
    res["windows"] = windows
    res["workload_batch8"] = ("BASELINE configs[4] per GPU: 8 songs denoised together, DDPM loop body with the first half of every image known "
                              "(the autoregressive overlap of inference_sdf.py:227-283)")
    return res


def config3_line(model, params, steps=10, windows=3, probe=None):
    """BASELINE.json configs[2]: DDIM 50 steps, uncond_scale 5 (classifier-free guidance: every step evaluates 2 x 32 = 64 samples), batch 32.
    Timed through DDIMSampler.p_sample (eager): windows of `steps` reverse steps from tau index 49, median window."""
    from polyffusion_amd.sampler import DDIMSampler
    dev = torch.device("cuda", torch.cuda.current_device())
    B = 32
    cond = model._encode_chord(torch.from_numpy(synth.chords(B, seed=888)).to(dev))
    uc = -torch.ones(B, 1, params.d_cond, device=dev)
    d = DDIMSampler(model.ldm, 50, "uniform", 0.0, seed=4)
    shape = (B, params.out_channels, params.img_h, params.img_w)
    x = d.randn(shape, dev)
    prep = d.prepare(cond, uncond_scale=5.0, uncond_cond=uc)   # what DDIMSampler.paint hoists out of its loop
    out = [x]

    def run(n):
        xx = x
        for i, step in enumerate(np.flip(d.time_steps)[:n]):
            xx, _, _ = d.p_sample(xx, cond, None, int(step), 49 - i, uncond_scale=5.0, uncond_cond=uc, prep=prep)
        out[0] = xx

    run(2)
    med, all_s, mhz = windowed(lambda: run(steps), windows, probe)
    assert torch.isfinite(out[0]).all()
    f_eval = F_MIN_PER_SAMPLE_EVAL - F_UPFOLD_SAVED
    # work actually executed: the guidance evaluation computes everything in front of the first transformer block once for both halves
    # (pf_unet_forward_cfg).  The prefix's FLOPs are read off a profiled plain evaluation: every launch before the first linear layer.
    unet = model.ldm.eps_model
    shared = 0.0
    if d.share_cfg_prefix:
        unet.set_profiling(True)
        unet(x[:2], torch.full((2,), 999, dtype=torch.long, device=dev), cond[:2])
        torch.cuda.synchronize()
        for kind, _ms, fl in unet.read_profile():
            if kind == 1:
                break
            shared += fl / 2.0
        unet.set_profiling(False)
    executed = f_eval * 64 - shared * 32
    return {"workload": "sdf_chd8bar DDIM 50-step, uncond_scale 5, batch 32 (64 UNet sample-evals per step), 1 GPU", "steps": steps, "windows": windows,
            "steps_per_s": round(steps / med, 3), "ms_per_step": round(med / steps * 1e3, 3), "sample_evals_per_s": round(64 * steps / med, 1),
            "windows_ms_per_step": [round(t / steps * 1e3, 3) for t in all_s], "sclk_mhz": mhz,
            "shared_prefix": {"on": bool(d.share_cfg_prefix), "gflop_per_sample": round(shared / 1e9, 3),
                              "note": "layers in front of the first transformer block are evaluated on 32 samples instead of 64 (identical for the "
                                      "conditional and unconditional halves); roofline_frac counts the executed work only"},
            "executed_gflop_per_step": round(executed / 1e9, 1),
            "roofline_frac": round(executed * steps / med / (PEAK_ALGO["bf16x3"] * 1e12), 4)}


def config4_line(steps=20, windows=3, probe=None):
    """The per-GPU shape of BASELINE.json configs[3]: sdf_txt (texture condition through the HIP texture encoder, d_cond 1024), batch 16
    (128 samples sharded over 8 GPUs), DDPM loop body - the headline's step with the other conditioning width."""
    from polyffusion_amd.inference_sdf import synthetic_model
    dev = torch.device("cuda", torch.cuda.current_device())
    p = preset("sdf_txt")
    m = synthetic_model(p)
    m.ldm.eps_model.set_precision("bf16x3")
    cond = m._encode_txt(torch.from_numpy(synth.prmat(BATCH, 32)).to(dev))
    s = SDFSampler(m.ldm, seed=11)
    shape = (BATCH, p.out_channels, p.img_h, p.img_w)
    z = torch.zeros(shape, device=dev)
    st = {"x": s.q_sample(z, p.n_steps - 1, s.randn(shape, dev)), "t": p.n_steps - 1}
    prep = s.prepare(cond)

    def run(n):
        for _ in range(n):
            st["x"] = s.repaint_step(st["x"], cond, st["t"], z, z, prep=prep)
            st["t"] = max(st["t"] - 1, 1)

    run(3)
    med, all_s, mhz = windowed(lambda: run(steps), windows, probe)
    assert torch.isfinite(st["x"]).all()
    return {"workload": "sdf_txt (d_cond 1024, HIP texture encoder) batch 16 per GPU, DDPM loop body: BASELINE configs[3] per GPU", "steps": steps,
            "windows": windows, "steps_per_s": round(steps / med, 3), "ms_per_step": round(med / steps * 1e3, 4),
            "windows_ms_per_step": [round(t / steps * 1e3, 4) for t in all_s], "sclk_mhz": mhz, "cond_shape": list(cond.shape)}


def long_parity_note(model, measure: bool):
    """The full-length f32-vs-bf16x3 comparison of config 3 (50 DDIM steps, guidance 5, B = 32: ~4 s; tools/long_parity.py), measured in this run
    when `measure`.  Nothing cached goes into the line: the 1000-step comparisons live in the GPU suite (tests/test_gpu_long_parity.py)."""
    keep = lambda r: {"final_max_abs": r["max_abs"], "final_rms": r["rms"], "image_rms": r["ref_rms"], "image_max_abs": r["ref_max_abs"],
                      "notes": r["notes"], "curve_max_abs_last": r["curve_max_abs"][-1][2], "what": r["what"]}
    out = {}
    if measure:
        try:
            from tools import long_parity
            out["config3"] = dict(keep(long_parity.config3(model)), source="measured in this run")
        except Exception as e:   # never lose the bench line to a diagnostic
            out["config3_error"] = f"{type(e).__name__}: {e}"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=5, help="back-to-back timed windows of --steps steps each; the median window is reported")
    ap.add_argument("--profile-steps", type=int, default=2, help="profiled steps for the roofline object (0 disables)")
    ap.add_argument("--f16x3-steps", type=int, default=20, help="steps per window of the f16x3 leg (fp16-split build, alternating with bf16x3 windows; 0 disables)")
    ap.add_argument("--fp32-steps", type=int, default=10, help="steps of the exact-fp32-MFMA mode measured in the same run (0 disables)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not spawn the two rocprofv3 counter passes that measure roofline.traffic in the run")
    ap.add_argument("--small-batch-steps", type=int, default=40, help="reverse steps of the batch-1 / batch-8 eager-vs-hipGraph lines (0 disables)")
    ap.add_argument("--secondary-windows", type=int, default=3, help="timed windows of every secondary line (fp32 mode, small batch, configs 3 / 4); median reported")
    ap.add_argument("--no-long-parity", action="store_true", help="do not re-measure the 50-step DDIM f32-vs-bf16x3 comparison for precision_note")
    ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16x3"],
                    help="arithmetic of the dense contractions: exact fp32 MFMA, or the error-compensated bf16x3 split")
    ap.add_argument("--dump-launches", action="store_true", help="print one line per kernel launch of a profiled step (stderr)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=0|1",
                    help="pin a plan option of the UNet handle (pf_unet_set_option: mlp_fused, attn_wide, conv_t16, conv_pp, conv_wino) for A/B runs; "
                         "the default line runs with every option automatic")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher - one rank per GPU under torch.distributed.run (rendezvous on
        # 127.0.0.1, a free port), same arguments; rank 0 of that job prints the ONE JSON line
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        sys.stdout.flush()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    rank, world, local = pfdist.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    params = preset("sdf_chd8bar")
    model, bcast_s = build_model(params, rank, world)
    unet = model.ldm.eps_model
    unet.set_precision(args.precision)
    for o in args.option:
        name, val = o.split("=")
        unet.set_option(name, bool(int(val)))
    ranks_seen, devices_seen = pfdist.ranks_seen()

    # per-rank batch of 16: global sample indices [rank*16, rank*16+16) -> noise streams independent of N
    lo = rank * BATCH
    chords = torch.from_numpy(synth.chords(BATCH * world, seed=4242)[lo:lo + BATCH]).to(dev)
    cond = model._encode_chord(chords)  # [16,1,512] through the HIP chord encoder
    sampler = SDFSampler(model.ldm, seed=1234, sample_offset=lo)
    shape = (BATCH, params.out_channels, params.img_h, params.img_w)
    zeros = torch.zeros(shape, device=dev)
    x = sampler.q_sample(zeros, params.n_steps - 1, sampler.randn(shape, dev))  # what Experiments.predict feeds paint()
    prep = sampler.prepare(cond)     # once per paint(): time-bias table + collapsed cross-attention biases (not part of a step)

    def step_fn(x_t, step):          # the loop body of SDFSampler.paint with a known region (orig = mask = 0, as Experiments.predict passes)
        return sampler.repaint_step(x_t, cond, step, zeros, zeros, prep=prep)

    probe = ClockProbe(dev)
    smi_before = smi_snapshot() if rank == 0 else {}
    t_step = params.n_steps - 1
    for _ in range(args.warmup):
        x = step_fn(x, t_step); t_step = max(t_step - 1, 1)
    wins = []
    for _ in range(max(1, args.windows)):
        x, t_step, el, ev, mhz = timed_loop(step_fn, x, t_step, args.steps, probe)
        wins.append((el, ev, mhz))
    assert torch.isfinite(x).all(), "non-finite sample"
    smi_after = smi_snapshot() if rank == 0 else {}
    order = sorted(range(len(wins)), key=lambda i: wins[i][0])
    elapsed, ev_elapsed, med_mhz = wins[order[len(order) // 2]]

    ms_per_step = elapsed / args.steps * 1e3
    # this rank's own clock over the median window's steps (events on its launch stream): a straggler shows as max >> min
    rank_ms = pfdist.gather_floats(ev_elapsed / args.steps * 1e3)
    # work actually executed (never count skipped work): the bf16x3 plan folds nearest-x2 + conv3x3 into four 2x2 convs
    f_eval = F_MIN_PER_SAMPLE_EVAL - (F_UPFOLD_SAVED if args.precision == "bf16x3" else 0.0)
    value = world * args.steps / elapsed
    out = {
        "metric": "denoising steps/sec (8-bar prmat2c, batch 16)", "value": round(value, 4), "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": "sdf_chd8bar cond generation, batch 16 per GPU, 1000-step DDPM sampler loop body "
                               "(BASELINE.json configs[1]); weights: deterministic synthetic, 41.08M params",
                   "global_batch": BATCH * world, "unet_evals_per_step": BATCH * world, "parallelism": f"batch-shard x{world}",
                   "weight_broadcast_s": round(bcast_s, 4), "launches_per_step": unet.n_launches(BATCH, prepared=True) + 1},
        "ms_per_step_hip_events": round(ev_elapsed / args.steps * 1e3, 4),
        "per_rank_ms_per_step": {"min": round(min(rank_ms), 4), "max": round(max(rank_ms), 4), "all": [round(v, 4) for v in rank_ms],
                                 "how": "each rank's HIP-event time over the median window's steps on its own stream (no barrier inside)"},
        "windows": len(wins), "windows_ms_per_step": [round(w[0] / args.steps * 1e3, 4) for w in wins],
        "windows_spread_pct": round((max(w[0] for w in wins) - min(w[0] for w in wins)) / elapsed * 100, 2),
        "window_rule": "value / ms_per_step = the MEDIAN of the windows (each exactly --steps steps between barrier + synchronize pairs)",
        "sclk_mhz": {"median_window": med_mhz, "per_window": [w[2] for w in wins],
                     "how": "average shader clock of the XCDs over each timed window = d(s_memtime) / d(s_memrealtime) per XCD between two probe "
                            "kernels on the launch stream (median / min / max over the XCDs), reference counter calibrated against the host clock "
                            "(%s Hz)" % (None if probe.ref_hz is None else round(probe.ref_hz)),
                     "rocm_smi_before": smi_before, "rocm_smi_after": smi_after},
        "ranks_seen": ranks_seen, "devices_seen": devices_seen,
        "path_tflops": round(f_eval * BATCH * world * args.steps / elapsed / 1e12, 3),
        "path_frac_of_f32_mfma_peak": round(f_eval * BATCH * args.steps / elapsed / (PEAK_F32_MFMA_TFLOPS * 1e12), 4),
        "path_flops_per_sample_eval": f_eval,
        "plan_options": {n: unet.get_option(n) for n in ("mlp_fused", "attn_wide", "conv_t16", "conv_pp", "conv_wino")},
        "precision_note": ("dense contractions on the bf16 matrix pipe as an error-compensated split (3 MFMAs per product, fp32 accumulate); "
                           "UNet max-abs-diff vs the reference 5.2e-5 (contract 1e-3)") if args.precision == "bf16x3"
        else "dense contractions on the fp32 matrix pipe (exact fp32 FMA chains)",
    }

    if rank == 0 and args.profile_steps > 0:
        x, agg = profiled_pass(unet, step_fn, x, t_step, args.profile_steps, args.precision, args.dump_launches)
        k = agg.get(0)
        # an event pair around a launch measures the kernel plus the pair's own cost (the pair keeps the next kernel from starting
        # early): calibrate it on empty pairs and take it off every launch (rocprofv3's per-kernel averages in profiles/ agree
        # with the corrected figure, not with the raw one)
        pairs = []
        for _ in range(50):
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record(); a1.record()
            pairs.append((a0, a1))
        torch.cuda.synchronize()
        ev_pair_ms = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2]
        if k:
            corr_ms = max(k[1] - k[0] * ev_pair_ms, 0.5 * k[1])
            ach_corr = k[2] / (corr_ms * 1e-3) / 1e12
            ach = k[2] / (k[1] * 1e-3) / 1e12      # `achieved` / `frac` stay on the raw (pessimistic) event durations
            peak = PEAK_ALGO[args.precision]
            traffic, traffic_src = (None, "--no-pmc") if (args.no_pmc or world != 1) else measure_traffic(args.precision)
            tpath = os.path.join(REPO, "profiles", f"pmc_traffic_{args.precision}.json")
            if traffic is None and os.path.exists(tpath):  # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE)
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
                traffic_src = (f"profiles/pmc_traffic_{args.precision}.json: bytes per launch from the committed rocprofv3 --pmc passes of this workload "
                               f"(not re-measured in this run: {traffic_src})")
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
                               "frac": round(ach / peak, 4), "traffic": traffic,
                               "traffic_source": traffic_src,
                               "peak_note": ("fp32-equivalent ceiling = bf16 dense MFMA peak 2500 TFLOP/s / 3 MFMAs per product"
                                             if args.precision == "bf16x3" else "fp32 MFMA dense peak"),
                               "matrix_pipe_frac": round(ach * (3.0 if args.precision == "bf16x3" else 1.0)
                                                         / (PEAK_BF16_MFMA_TFLOPS if args.precision == "bf16x3" else PEAK_F32_MFMA_TFLOPS), 4),
                               "kernel": ("conv_bf3_kernel<3x3> (bf16x3 split MFMA" if args.precision == "bf16x3" else "conv_mfma_kernel<3x3> (fp32 MFMA") + ", fused GN+SiLU prologue)",
                               "launches_per_step": k[0] // args.profile_steps,
                               "avg_launch_ms": round(k[1] / k[0], 4),
                               "flops_per_step": k[2] / args.profile_steps,
                               # the same family priced as if every layer ran its direct form: comparable between plans and rounds
                               # (achieved / frac count the operations actually executed - Winograd launches run 16 / 36 of the direct count)
                               "direct_equivalent_tflops": round(k[3] / (k[1] * 1e-3) / 1e12, 3),
                               "direct_equivalent_flops_per_step": k[3] / args.profile_steps,
                               "winograd_launches_per_step": k[4] // args.profile_steps,
                               "event_pair_overhead_ms": round(ev_pair_ms, 5),
                               "achieved_event_corrected": round(ach_corr, 3), "frac_event_corrected": round(ach_corr / peak, 4),
                               "timing_note": "hipEvents around every launch on the launch stream. An event pair adds its own cost to what it "
                                              "brackets (median of empty pairs: event_pair_overhead_ms); achieved/frac use the raw durations and are "
                                              "therefore a lower bound, *_event_corrected subtract one pair per launch and are an upper bound - "
                                              "rocprofv3's per-kernel averages (profiles/) lie between the two"}
            if args.precision == "bf16x3":
                sus = mfma_sustained()
                rl = out["roofline"]
                rl["sustained"] = sus
                rl["frac_of_sustained"] = round(ach / (sus["tflops"] / 3.0), 4)
                rl["frac_of_sustained_event_corrected"] = round(ach_corr / (sus["tflops"] / 3.0), 4)
        out["kernel_ms_per_step"] = {KIND_NAMES[kind]: round(v[1] / args.profile_steps, 4) for kind, v in sorted(agg.items())}
        # the same with one empty event pair taken off every launch: what rocprofv3's per-kernel durations add up to (profiles/)
        out["kernel_ms_per_step_corrected"] = {KIND_NAMES[kind]: round(max(v[1] - v[0] * ev_pair_ms, 0.0) / args.profile_steps, 4)
                                               for kind, v in sorted(agg.items())}
        out["kernel_ms_per_step_corrected"]["sum"] = round(sum(out["kernel_ms_per_step_corrected"].values()), 4)
        # everything that is not a 3x3 convolution (transformer linears, attention, norms, stem / head, sampler update): time and rate.
        # The figure is the headline's wall time per step apportioned by the per-launch event durations (raw: every launch carries the same
        # event-pair overhead, so small launches weigh slightly more than they cost - the cautious side); the two bounds beside it are the
        # raw event sum (too high: it contains one event pair per launch) and the sum with one pair taken off every launch (too low).
        raw_nonconv = sum(v[1] for kind, v in agg.items() if kind != 0) / args.profile_steps
        raw_total = sum(v[1] for kind, v in agg.items()) / args.profile_steps
        corr_nonconv = sum(max(v[1] - v[0] * ev_pair_ms, 0.0) for kind, v in agg.items() if kind != 0) / args.profile_steps
        nonconv_ms = out["ms_per_step"] * raw_nonconv / max(raw_total, 1e-12)
        nonconv_fl = sum(v[2] for kind, v in agg.items() if kind != 0) / args.profile_steps
        out["nonconv_ms_per_step"] = round(nonconv_ms, 4)
        out["nonconv_ms_per_step_bounds"] = {"event_corrected": round(corr_nonconv, 4), "raw_events": round(raw_nonconv, 4),
                                             "how": "nonconv_ms_per_step = ms_per_step x (raw event time of the launches that are not 3x3 convs / raw event time of all launches)"}
        out["nonconv_frac_of_833"] = round(nonconv_fl / max(nonconv_ms * 1e-3, 1e-12) / 1e12 / PEAK_ALGO["bf16x3"], 4) if args.precision == "bf16x3" else None

    if args.fp32_steps > 0 and args.precision == "bf16x3":
        # the exact-fp32-MFMA mode in the same run, same workload, the headline's protocol (every rank runs it so the barriers line up)
        unet.set_precision("f32")
        for _ in range(2):
            x = step_fn(x, t_step)
        w32 = []
        for _ in range(max(1, args.secondary_windows)):
            x, t_step, el32, _, mhz32 = timed_loop(step_fn, x, t_step, args.fp32_steps, probe)
            w32.append((el32, mhz32))
        o32 = sorted(range(len(w32)), key=lambda i: w32[i][0])
        el32, mhz32 = w32[o32[len(o32) // 2]]
        fp32 = {"steps_per_s": round(world * args.fp32_steps / el32, 4), "ms_per_step": round(el32 / args.fp32_steps * 1e3, 4),
                "steps": args.fp32_steps, "windows": len(w32), "windows_ms_per_step": [round(w[0] / args.fp32_steps * 1e3, 4) for w in w32],
                "sclk_mhz": mhz32,
                "path_frac_of_157": round(F_MIN_PER_SAMPLE_EVAL * BATCH * args.fp32_steps / el32 / (PEAK_F32_MFMA_TFLOPS * 1e12), 4)}
        if rank == 0 and args.profile_steps > 0:
            x, agg32 = profiled_pass(unet, step_fn, x, t_step, 1, "f32")
            if agg32.get(0):
                fp32["conv3x3_frac_of_157"] = round(agg32[0][2] / (agg32[0][1] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
        unet.set_precision(args.precision)
        out["fp32_mode"] = fp32
    if args.f16x3_steps > 0 and args.precision == "bf16x3" and _lib.X3_VARIANT == "":
        # the fp16 split (libpfhip_f16.so) on the same workload: a second model instance in the other build of the library, same weights,
        # same sampler state; windows alternate with bf16x3 windows so both see the same clocks (every rank runs it: barriers line up)
        model16, _ = build_model(params, rank, world, x3="f16")
        u16 = model16.ldm.eps_model
        u16.set_precision("f16x3")
        for o in args.option:
            name, val = o.split("=")
            u16.set_option(name, bool(int(val)))
        s16 = SDFSampler(model16.ldm, seed=1234, sample_offset=lo)
        prep16 = s16.prepare(cond)

        def step16(x_t, step):
            return s16.repaint_step(x_t, cond, step, zeros, zeros, prep=prep16)

        x16 = x.clone()
        for _ in range(3):
            x16 = step16(x16, t_step)
        wa, wb = [], []
        for _ in range(max(1, args.secondary_windows)):
            x, t_step, el_b, _, mhz_b = timed_loop(step_fn, x, t_step, args.f16x3_steps, probe)
            x16, _, el_h, _, mhz_h = timed_loop(step16, x16, t_step, args.f16x3_steps, probe)
            wb.append(el_b); wa.append((el_h, mhz_h))
        assert torch.isfinite(x16).all(), "non-finite sample (f16x3)"
        el_h, mhz_h = sorted(wa, key=lambda w: w[0])[len(wa) // 2]
        el_b = sorted(wb)[len(wb) // 2]
        f16 = {"steps_per_s": round(world * args.f16x3_steps / el_h, 4), "ms_per_step": round(el_h / args.f16x3_steps * 1e3, 4),
               "steps": args.f16x3_steps, "windows": len(wa), "windows_ms_per_step": [round(w[0] / args.f16x3_steps * 1e3, 4) for w in wa],
               "bf16x3_alternating_ms_per_step": round(el_b / args.f16x3_steps * 1e3, 4), "speed_vs_bf16x3": round(el_b / el_h, 4),
               "sclk_mhz": mhz_h, "launches_per_step": u16.n_launches(BATCH, prepared=True) + 1,
               "path_frac_of_833": round(f_eval * BATCH * args.f16x3_steps / el_h / (PEAK_ALGO["bf16x3"] * 1e12), 4)}
        if rank == 0:
            # one evaluation, three arithmetics, same inputs: distance of each split to the exact-fp32-MFMA mode, relative to max|eps|
            xe = sampler.randn(shape, dev)
            te = torch.full((BATCH,), 500, dtype=torch.long, device=dev)
            unet.set_precision("f32")
            unet.track_absmax(True)      # range telemetry on the f32 evaluation: the largest |value| any layer stores (pf_unet_track_absmax)
            e32 = unet(xe, te, cond).clone()
            am = unet.read_absmax()
            unet.track_absmax(False)
            f16["max_abs_activation"] = float(f"{am:.4g}")
            f16["fp16_headroom"] = float(f"{65504.0 / am:.4g}") if am > 0 else None
            f16["max_abs_activation_note"] = ("largest |value| stored by any layer of one evaluation of this workload (B = 16, t = 500), measured in the f32 mode; "
                                              "f16x3's split pieces overflow beyond 65504; the CLI's --precision auto keeps f16x3 only with >= 8x headroom on its probe")
            unet.set_precision("bf16x3")
            eb, eh = unet(xe, te, cond), u16(xe, te, cond)
            sc = e32.abs().max().item()
            f16["eps_rel_diff_vs_f32_mode"] = {"bf16x3": float(f"{(eb - e32).abs().max().item() / sc:.3e}"), "f16x3": float(f"{(eh - e32).abs().max().item() / sc:.3e}")}
            f16["vs_float64"] = ("tests/test_gpu_f16x3.py, B = 2 against the oracle in float64, relative to max|eps|: f32 mode 1.8e-6 / bf16x3 1.6e-5 / "
                                 "f16x3 1.4e-6 on the synthetic weights, 1.7e-4 / 1.6e-3 / 1.3e-4 on the worst stressed net")
        out["f16x3_mode"] = f16
        del s16, prep16, u16, model16
    if rank == 0 and args.small_batch_steps > 0:
        out["small_batch"] = small_batch_lines(model, params, args.small_batch_steps, args.precision, args.secondary_windows, probe)
        out["config3"] = config3_line(model, params, 10, args.secondary_windows, probe)
        if world == 1:
            out["config4_shape"] = config4_line(20, args.secondary_windows, probe)
    if rank == 0 and args.precision == "bf16x3":
        out["long_parity"] = long_parity_note(model, measure=(world == 1 and not args.no_long_parity and args.small_batch_steps > 0))
        lp = out["long_parity"]
        if "config3" in lp:
            k = lp["config3"]
            out["precision_note"] += (f"; full-length config 3 loop f32 vs bf16x3 on one noise tape, measured in this run: final max-abs-diff {k['final_max_abs']:.1e} of images "
                                      f"at rms {k['image_rms']:.0f}, {k['notes']['onset_bits_differ'] + k['notes']['sustain_bits_differ']} of {2 * k['notes']['cells']} note cells differ")
        out["precision_note"] += ("; the 1000-step loops (configs 2 / 5) and all three against the REAL reference: tests/test_gpu_long_parity.py, "
                                  "tests/test_gpu_long_reference.py (source: the GPU suite, not this run)")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(params)
    if rank == 0:
        print(json.dumps(out), flush=True)
    pfdist.barrier()


if __name__ == "__main__":
    main()
