"""f16x3: the split-precision kernels compiled with fp16 pieces (libpfhip_f16.so, -DPF_X3_F16; csrc/pf_internal.h).

Same three MFMAs per product as bf16x3 (v_mfma_f32_32x32x16_f16), 11 + 11 mantissa bits instead of 8 + 8: the arithmetic error drops to
that of the exact-fp32 mode (the reference computes in fp32: ref:stable_diffusion/model/unet.py runs under no autocast), for ~3 % of the
speed and fp16's range.  Checked here:
  * a whole UNet evaluation against the oracle in FLOAT64, ordinary and badly-scaled weights, beside f32 and bf16x3 on the same inputs;
  * the two builds side by side in one process (a model lives in one library; interleaving them changes no bit);
  * a sampler loop (eager and captured) on an f16x3 model;
  * an activation beyond fp16's range is caught by the precision probe;
  * every operator-level test of the split kernels, re-run against the fp16 build in a child process (PF_X3=f16).
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import unet_ref  # noqa: E402
from polyffusion_amd import _lib, synth  # noqa: E402
from polyffusion_amd.arch import UNetConfig  # noqa: E402
from polyffusion_amd.inference_sdf import build_unet, pick_precision, synthetic_model  # noqa: E402
from polyffusion_amd.params import preset  # noqa: E402
from polyffusion_amd.weights import synth_unet_state  # noqa: E402
from test_gpu_long_parity import stress_inputs, stressed_state  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

needs_default_bf16 = pytest.mark.skipif(_lib.X3_VARIANT == "f16", reason="PF_X3=f16: the process default already is the fp16 build")


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    _lib.require_gpu()
    _lib.load("f16")


@needs_default_bf16
@pytest.mark.parametrize("kind", ["synthetic", "harsh0", "harsh1"])
def test_f16x3_unet_against_the_float64_oracle_beside_f32_and_bf16x3(kind):
    """Full-size sdf_chd8bar UNet, B = 2.  Relative to max|eps| of the float64 result: f16x3 is at least 4x closer than bf16x3 (observed
    8-15x) and within 2x of the exact-fp32-MFMA mode (observed: equal or better - its products carry 22 bits, its sums are the same fp32
    accumulators)."""
    p = preset("sdf_chd8bar")
    cfg = UNetConfig(d_cond=p.d_cond)
    st = synth_unet_state(cfg, 0) if kind == "synthetic" else stressed_state(cfg, int(kind[-1]))
    x, t, c = stress_inputs(0 if kind == "synthetic" else int(kind[-1]), p.d_cond)
    if kind == "synthetic":
        x = torch.from_numpy(synth.gaussian((2, 2, 128, 128), 77))
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        truth = unet_ref.unet_forward(unet_ref.to_torch(st, dtype=torch.float64), cfg, x.double(), t, c.double())
    scale = truth.abs().max().item()
    rel = lambda v: (v.double() - truth).abs().max().item() / scale   # noqa: E731
    errs = {}
    u = build_unet(p)
    u.load_state_dict(st)
    for mode in ("f32", "bf16x3"):
        u.set_precision(mode)
        errs[mode] = rel(u(x.cuda(), t.cuda(), c.cuda()).cpu())
    u16 = build_unet(p, x3="f16")
    u16.load_state_dict(st)
    assert u16.split_mode == "f16x3" and u16.precision == "f32"
    u16.set_precision("f16x3")
    got = u16(x.cuda(), t.cuda(), c.cuda()).cpu()
    assert torch.isfinite(got).all()
    errs["f16x3"] = rel(got)
    print(f"{kind}: max|eps| {scale:.3g}; vs float64, relative: f32 mode {errs['f32']:.2e}, bf16x3 {errs['bf16x3']:.2e}, f16x3 {errs['f16x3']:.2e}")
    assert errs["f16x3"] <= errs["bf16x3"] / 4
    assert errs["f16x3"] <= 2 * errs["f32"] + 1e-6
    # the fp16 build's f32 mode is the same fp32 kernels on the same fp32 packing
    u16.set_precision("f32")
    u.set_precision("f32")
    assert torch.equal(u16(x.cuda(), t.cuda(), c.cuda()), u(x.cuda(), t.cuda(), c.cuda()))
    with pytest.raises(ValueError, match="f16x3"):
        u16.set_precision("bf16x3")
    with pytest.raises(ValueError, match="x3='f16'"):
        u.set_precision("f16x3")


@needs_default_bf16
def test_both_builds_in_one_process_do_not_disturb_each_other():
    """A bf16x3 model and an f16x3 model evaluated alternately, on one stream, each with its own weight blob and workspace: every result
    equals the model's own first evaluation bit for bit (nothing is shared between the libraries but the HIP runtime)."""
    p = preset("sdf_chd8bar")
    ma, mb = synthetic_model(p), synthetic_model(p, x3="f16")
    ua, ub = ma.ldm.eps_model, mb.ldm.eps_model
    ua.set_precision("bf16x3")
    ub.set_precision("f16x3")
    B = 4
    x = torch.from_numpy(synth.gaussian((B, 2, 128, 128), 5)).cuda()
    c = ma._encode_chord(torch.from_numpy(synth.chords(B, 6)).cuda())
    t = torch.tensor([999, 500, 20, 0], device="cuda")
    a0, b0 = ua(x, t, c).clone(), ub(x, t, c).clone()
    d = (a0 - b0).abs().max().item()
    assert 0 < d < 2e-4 * a0.abs().max().item()       # two roundings of the same function
    for _ in range(3):
        assert torch.equal(ub(x, t, c), b0)
        assert torch.equal(ua(x, t, c), a0)
    assert ua.n_launches(B) == ub.n_launches(B)        # same plan


@needs_default_bf16
def test_sampler_loop_on_an_f16x3_model_eager_and_captured():
    """10 DDIM steps with guidance (shared-prefix evaluation) on an f16x3 model: the captured step replays the eager loop bit for bit, and
    the result sits closer to the f32-mode loop than the bf16x3 loop does."""
    from polyffusion_amd.sampler import DDIMSampler
    p = preset("sdf_chd8bar")
    B = 2
    chord = torch.from_numpy(synth.chords(B, 31)).cuda()
    x = torch.from_numpy(synth.gaussian((B, 2, 128, 128), 32)).cuda()
    outs = {}
    for name, x3, mode in (("f32", None, "f32"), ("bf16x3", None, "bf16x3"), ("f16x3", "f16", "f16x3")):
        m = synthetic_model(p, x3=x3)
        m.ldm.eps_model.set_precision(mode)
        cond = m._encode_chord(chord)
        uncond = -torch.ones_like(cond)
        for graph in ((False, True) if name == "f16x3" else (False,)):
            d = DDIMSampler(m.ldm, 10, "uniform", 0.0, seed=123, graph=graph)
            outs[(name, graph)] = d.paint(x, cond, 9, uncond_scale=3.0, uncond_cond=uncond).clone()
            if graph:
                assert d.graph_captures == 1
    ref = outs[("f32", False)]
    e16 = (outs[("f16x3", False)] - ref).abs().max().item()
    eb = (outs[("bf16x3", False)] - ref).abs().max().item()
    print(f"10 DDIM steps, scale 3: |f16x3 - f32| {e16:.2e}, |bf16x3 - f32| {eb:.2e}")
    assert torch.isfinite(ref).all() and torch.equal(outs[("f16x3", True)], outs[("f16x3", False)])
    assert e16 < eb / 2 and e16 < 1e-4 * max(1.0, ref.abs().max().item())


@needs_default_bf16
def test_an_activation_beyond_fp16_range_is_caught_by_the_precision_probe_and_a_weight_beyond_it_at_load():
    """GroupNorm gain x 2000 and proj_in x 100 in the first transformer block: that block's residual stream (~2e5) leaves fp16's range
    (65504).  f32 and bf16x3 stay finite; f16x3 produces a non-finite result, which pick_precision (--precision auto-f16x3) reads as
    'does not agree' -> f32.  A WEIGHT that does not fit the fp16 packing (|w| 2^8 > 65504) is refused when it is loaded."""
    p = preset("sdf_chd8bar")
    cfg = UNetConfig(d_cond=p.d_cond)
    st = synth_unet_state(cfg, 0)
    key = next(k for k in st if k.endswith("proj_in.weight"))
    gain = key.replace("proj_in.weight", "norm.weight")
    st[key] = (st[key] * np.float32(100)).astype(np.float32)
    st[gain] = (st[gain] * np.float32(2000)).astype(np.float32)
    c = torch.from_numpy(synth.gaussian((3, 1, p.d_cond), 9)).cuda()
    u = build_unet(p)
    u.load_state_dict(st)
    mode, ratio = pick_precision(u, c)
    assert np.isfinite(ratio)
    u16 = build_unet(p, x3="f16")
    u16.load_state_dict(st)
    mode16, ratio16 = pick_precision(u16, c)
    assert mode16 == "f32" and u16.precision == "f32" and not np.isfinite(ratio16)
    # and on the ordinary weights it keeps f16x3, with a probe distance far below bf16x3's
    st = synth_unet_state(cfg, 0)
    u.load_state_dict(st)
    u16.load_state_dict(st)
    (mb, rb), (m16, r16) = pick_precision(u, c), pick_precision(u16, c)
    print(f"probe distance to f32: bf16x3 {rb:.2e}, f16x3 {r16:.2e}")
    assert mb == "bf16x3" and m16 == "f16x3" and r16 < rb / 3
    st[key] = (st[key] * np.float32(1e5)).astype(np.float32)
    u.load_state_dict(st)                                       # fine for bf16 pieces (fp32's range)
    with pytest.raises(RuntimeError, match="fp16 split packing"):
        u16.load_state_dict(st)


@needs_default_bf16
def test_operator_level_suite_of_the_split_kernels_against_the_fp16_build():
    """PF_X3=f16 makes the fp16 build the process default, so the unchanged operator tests (3x3 / 1x1 / strided / upsampling convs with
    every prologue and epilogue, planes GEMMs, both attention forms and the key split, the fused MLP and pre-attention launches, the
    shared-prefix guidance plan) run against libpfhip_f16.so with the tolerances written for bf16x3."""
    env = dict(os.environ, PF_X3="f16")
    files = ["tests/test_gpu_bf16x3.py", "tests/test_gpu_mlp_fused.py", "tests/test_gpu_cfg_share.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + files, cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=1500)
    tail = "\n".join(r.stdout.strip().splitlines()[-15:])
    assert r.returncode == 0, tail + "\n" + r.stderr[-2000:]
    assert " passed" in tail and "failed" not in tail


@needs_default_bf16
def test_cli_precision_auto_tries_f16x3_before_falling_back_to_f32(tmp_path, monkeypatch, capsys):
    """--precision auto: bf16x3 if it agrees with f32 on the probe, else the same checkpoint in the fp16 build, else f32.  The probe
    threshold is lowered to sit between the two splits' distances (bf16x3 ~1e-5, f16x3 ~2e-6 on this small net), which is what a badly
    conditioned checkpoint does to them at the real threshold (harsh1 above: 1.5e-3 against 1.3e-4)."""
    import yaml
    from polyffusion_amd import inference_sdf
    from ckpt_fixture import full_state, write_legacy_pt
    from test_gpu_checkpoint_cli import PARAMS, states
    run = tmp_path / "run"
    (run / "chkpts").mkdir(parents=True)
    (run / "params.yaml").write_text(yaml.safe_dump(dict(PARAMS, batch_size=16, learning_rate=5e-5)))
    write_legacy_pt(str(run / "chkpts" / "weights_best.pt"), full_state(*states()))
    base = ["--chkpt_path", str(run / "chkpts" / "weights_best.pt"), "--synthetic", "--length", "2", "--ddim", "--ddim_steps", "4",
            "--uncond_scale", "2.0", "--seed", "11", "--num_generate", "1"]

    def go(tag, *extra):
        out = tmp_path / tag
        assert inference_sdf.main(base + ["--output_dir", str(out)] + list(extra)) == 0
        return np.load(out / sorted(f for f in os.listdir(out) if f.endswith(".npy"))[0]), capsys.readouterr().out

    a_auto, log = go("auto")
    assert "precision: bf16x3 (bf16x3 vs f32" in log and "(f16x3 vs f32" not in log
    a_bf, _ = go("bf", "--precision", "bf16x3")
    a_16, _ = go("h", "--precision", "f16x3")
    a_32, _ = go("f", "--precision", "f32")
    assert np.array_equal(a_auto, a_bf)
    assert 0 < np.abs(a_16 - a_32).max() < np.abs(a_bf - a_32).max()
    monkeypatch.setattr(inference_sdf, "PRECISION_PROBE_TOL", 5e-6)
    a_auto2, log = go("auto2")
    assert "precision: f32 (bf16x3 vs f32" in log and "precision: f16x3 (f16x3 vs f32" in log
    assert np.array_equal(a_auto2, a_16)
    monkeypatch.setattr(inference_sdf, "PRECISION_PROBE_TOL", 1e-9)
    a_auto3, log = go("auto3")
    assert "precision: f32 (f16x3 vs f32" in log
    assert np.array_equal(a_auto3, a_32)


@needs_default_bf16
def test_cli_refuses_non_finite_results(tmp_path):
    """An explicit --precision f16x3 on a checkpoint whose residual stream leaves fp16's range: the run stops with a message instead of
    writing silence (the note threshold would read NaN as 'no note'); bf16x3 on the same checkpoint runs."""
    import yaml
    from polyffusion_amd import inference_sdf
    from ckpt_fixture import full_state, write_legacy_pt
    from test_gpu_checkpoint_cli import PARAMS, states
    us, cs = states()
    key = next(k for k in us if k.endswith("proj_in.weight"))
    us[key] = (us[key] * np.float32(100)).astype(np.float32)
    us[key.replace("proj_in.weight", "norm.weight")] = (us[key.replace("proj_in.weight", "norm.weight")] * np.float32(2000)).astype(np.float32)
    run = tmp_path / "run"
    (run / "chkpts").mkdir(parents=True)
    (run / "params.yaml").write_text(yaml.safe_dump(dict(PARAMS, batch_size=16, learning_rate=5e-5)))
    write_legacy_pt(str(run / "chkpts" / "weights_best.pt"), full_state(us, cs))
    base = ["--chkpt_path", str(run / "chkpts" / "weights_best.pt"), "--synthetic", "--length", "2", "--ddim", "--ddim_steps", "4",
            "--uncond_scale", "1.0", "--seed", "5", "--num_generate", "1"]
    assert inference_sdf.main(base + ["--output_dir", str(tmp_path / "b"), "--precision", "bf16x3"]) == 0
    with pytest.raises(SystemExit, match="non-finite.*f16x3 overflows"):
        inference_sdf.main(base + ["--output_dir", str(tmp_path / "h"), "--precision", "f16x3"])
    assert not os.path.exists(tmp_path / "h") or not os.listdir(tmp_path / "h")


def test_range_telemetry_measures_what_the_layers_store():
    """pf_unet_track_absmax: (1) one pf_conv2d launch reports exactly max|out| of what it stored; (2) a full forward reports a modest number on
    the synthetic weights - the same in the f32 and the split modes, up to their rounding - and the headroom 65504 / it; (3) on the net whose
    residual stream is driven beyond fp16's range (the construction of the overflow test above) the f32 measurement says so BEFORE f16x3 is
    tried: pick_precision refuses f16x3 without evaluating into an overflow."""
    import ctypes as C
    from polyffusion_amd import _lib
    from test_gpu_ops import dev, nhwc, rnd, run_conv, gn_scale_shift, pack_w
    from test_gpu_bf16x3 import pack3
    import torch.nn.functional as F
    lib = _lib.load()
    B, H, W, c, cout = 2, 32, 32, 64, 64
    x = rnd((B, c, H, W), 1) * 3.0
    w = rnd((cout, c, 3, 3), 2, (1.0 / (c * 9)) ** 0.5)
    gamma, beta = 1 + 0.1 * rnd((c,), 4), 0.1 * rnd((c,), 5)
    x0 = dev(nhwc(x))
    sc, sh = gn_scale_shift(lib, x0, None, dev(gamma), dev(beta), 1e-5)
    for prec in (0, 1):
        out = torch.empty(B, H, W, cout, device="cuda")
        slot = torch.zeros(1, dtype=torch.int32, device="cuda")
        run_conv(lib, x0=x0, c0=c, batch=B, hin=H, win=W, ks=3, stride=1, ups=0, w=pack3(lib, w) if prec else pack_w(lib, w), n=cout, prologue=1,
                 sc=sc, sh=sh, out=out, ld_out=cout, precision=prec, absmax_slot=slot)
        assert slot.view(torch.float32).item() == out.abs().max().item()
    cfg = UNetConfig(d_cond=512)
    m = build_unet(preset("sdf_chd8bar"))
    m.load_state_dict(synth_unet_state(cfg, 0))
    xx = torch.from_numpy(synth.gaussian((2, 2, 128, 128), 7)).cuda()
    cc = torch.from_numpy(synth.gaussian((2, 1, 512), 8)).cuda()
    tt = torch.tensor([999, 3]).cuda()
    vals = {}
    for mode in ("f32", "bf16x3"):
        m.set_precision(mode)
        base = m(xx, tt, cc).clone()
        m.track_absmax(True)
        tracked = m(xx, tt, cc)
        vals[mode] = m.read_absmax()
        m.track_absmax(False)
        assert (tracked - base).abs().max().item() < 1e-5          # the tracked plan (chains instead of fused launches) computes the same thing
    print("max |stored activation|, synthetic weights:", vals, "fp16 headroom x%.0f" % (65504 / vals["f32"]))
    assert 1.0 < vals["f32"] < 8000 and abs(vals["f32"] - vals["bf16x3"]) < 1e-3 * vals["f32"]
    # (3) the overflow construction: the f32 probe SEES the 2e5 before the fp16 build is asked to evaluate it
    p = preset("sdf_chd8bar")
    st = synth_unet_state(cfg, 0)
    key = next(k for k in st if k.endswith("proj_in.weight"))
    gain = key.replace("proj_in.weight", "norm.weight")
    st[key] = (st[key] * np.float32(100)).astype(np.float32)
    st[gain] = (st[gain] * np.float32(2000)).astype(np.float32)
    u16 = build_unet(p, x3="f16")
    u16.load_state_dict(st)
    c3 = torch.from_numpy(synth.gaussian((3, 1, p.d_cond), 9)).cuda()
    mode16, ratio16 = pick_precision(u16, c3)
    am = pick_precision.last_absmax
    print(f"overflow net: largest |stored activation| on the f32 probe {am:.3g}")
    assert mode16 == "f32" and am > 65504
    u16.load_state_dict(synth_unet_state(cfg, 0))
    mode16, _ = pick_precision(u16, c3)
    assert mode16 == "f16x3" and 65504 / pick_precision.last_absmax >= 8
