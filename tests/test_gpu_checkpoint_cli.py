"""Row f1 on the GPU + the CLI (-m gpu): both checkpoint formats load through ``Polyffusion_SDF.load_trained`` and give the
same eps as directly-loaded weights; the pretrained-encoder loaders read checkpoint files; ``inference_sdf.main`` runs from a
run directory (params.yaml discovery, ``chkpts/weights_best.pt``), batches ``--num_generate`` songs, and its sharded mode
(songs split over ranks) reproduces the unsharded result on one GPU."""
import argparse
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

from ckpt_fixture import full_state, write_legacy_pt, write_lightning_ckpt  # noqa: E402
from polyffusion_amd import _lib, inference_sdf, synth  # noqa: E402
from polyffusion_amd.arch import UNetConfig  # noqa: E402
from polyffusion_amd.model_sdf import ChordEncoder, Polyffusion_SDF, load_pretrained_chd_enc, load_pretrained_txt_enc  # noqa: E402
from polyffusion_amd.params import Params  # noqa: E402
from polyffusion_amd.weights import synth_chord_encoder_state, synth_texture_encoder_state, synth_unet_state  # noqa: E402

# a small denoiser on the real 128x128 image with a real-size chord encoder interface (36 -> hidden 64 -> z 32 = d_cond)
PARAMS = dict(model_name="small_chd", in_channels=2, out_channels=2, channels=32, attention_levels=[1], n_res_blocks=1,
              channel_multipliers=[1, 2], n_heads=2, tf_layers=1, d_cond=32, linear_start=0.00085, linear_end=0.012, n_steps=1000,
              latent_scaling_factor=0.18215, img_h=128, img_w=128, cond_type="chord", cond_mode="mix", use_enc=True,
              chd_n_step=32, chd_input_dim=36, chd_z_input_dim=32, chd_hidden_dim=64, chd_z_dim=32)
CFG = UNetConfig.from_params(PARAMS)


def states():
    return synth_unet_state(CFG, 3), synth_chord_encoder_state(3, 36, 64, 32)


def assemble():
    p = Params(PARAMS)
    unet = inference_sdf.build_unet(p)
    ce, _ = inference_sdf.build_encoders(p)
    return p, unet, ce


def eps_of(model, seed=1):
    x = torch.from_numpy(synth.gaussian((2, 2, 128, 128), seed)).cuda()
    chd = torch.from_numpy(synth.chords(2, seed + 1)).cuda()
    cond = model._encode_chord(chd)
    return model.ldm(x, torch.tensor([10, 900]).cuda(), cond), cond


@pytest.fixture(scope="module")
def direct():
    _lib.require_gpu()
    p, unet, ce = assemble()
    us, cs = states()
    unet.load_state_dict(us)
    ce.load_state_dict(cs)
    m = Polyffusion_SDF(inference_sdf.build_ldm(p, unet), "chord", "mix", chord_enc=ce)
    return eps_of(m)


@pytest.mark.parametrize("fmt", ["pt", "ckpt"])
def test_load_trained_both_formats(tmp_path, direct, fmt):
    us, cs = states()
    st = full_state(us, cs)
    path = str(tmp_path / ("weights_best.pt" if fmt == "pt" else "epoch=9-step=99.ckpt"))
    (write_legacy_pt if fmt == "pt" else (lambda f, s: write_lightning_ckpt(f, s, dict(PARAMS, batch_size=16))))(path, st)
    p, unet, ce = assemble()
    m = Polyffusion_SDF.load_trained(inference_sdf.build_ldm(p, unet), path, "chord", "mix", chord_enc=ce)
    eps, cond = eps_of(m)
    assert torch.equal(cond, direct[1]) and torch.equal(eps, direct[0])    # same packed weights -> bit-identical
    # a checkpoint from a different architecture is refused with torch's wording
    bad = dict(st)
    bad["ldm.eps_model.input_blocks.0.0.weight"] = torch.zeros(64, 2, 3, 3)
    write_legacy_pt(str(tmp_path / "bad.pt"), bad)
    with pytest.raises(RuntimeError, match="size mismatch"):
        Polyffusion_SDF.load_trained(inference_sdf.build_ldm(*assemble()[:2]), str(tmp_path / "bad.pt"), "chord", "mix", chord_enc=assemble()[2])
    del bad["ldm.eps_model.out.2.bias"]
    bad["ldm.eps_model.input_blocks.0.0.weight"] = st["ldm.eps_model.input_blocks.0.0.weight"]
    write_legacy_pt(str(tmp_path / "bad2.pt"), bad)
    with pytest.raises(RuntimeError, match="missing"):
        Polyffusion_SDF.load_trained(inference_sdf.build_ldm(*assemble()[:2]), str(tmp_path / "bad2.pt"), "chord", "mix", chord_enc=assemble()[2])


def test_pretrained_encoder_files(tmp_path):
    cs = synth_chord_encoder_state(0)
    write_legacy_pt(str(tmp_path / "chd.pt"), {**{f"chord_enc.{k}": torch.as_tensor(v) for k, v in cs.items()},
                                               "chord_dec.grucell.weight_ih": torch.zeros(3, 3)})
    enc = load_pretrained_chd_enc(str(tmp_path / "chd.pt"), 36, 512, 512)
    want = ChordEncoder(36, 512, 512).load_state_dict(cs)
    chd = torch.from_numpy(synth.chords(3, 9)).cuda()
    assert torch.equal(enc(chd).mean, want(chd).mean)
    ts = synth_texture_encoder_state(0)
    torch.save({**{f"rhy_encoder.{k}": torch.as_tensor(v) for k, v in ts.items()}, "decoder.w": torch.zeros(2)}, str(tmp_path / "polydis.pt"))
    tenc = load_pretrained_txt_enc(str(tmp_path / "polydis.pt"), 256, 1024, 256, 10)
    pr = torch.from_numpy(synth.prmat(2, 4)).cuda().view(8, 32, 128)
    assert torch.isfinite(tenc(pr).mean).all() and tenc(pr).mean.shape == (8, 256)


def run_dir(tmp_path, fmt="pt"):
    run = tmp_path / "run"
    (run / "chkpts").mkdir(parents=True)
    (run / "params.yaml").write_text(yaml.safe_dump(dict(PARAMS, batch_size=16, learning_rate=5e-5)))
    us, cs = states()
    if fmt == "pt":
        write_legacy_pt(str(run / "chkpts" / "weights_best.pt"), full_state(us, cs))
    else:
        write_lightning_ckpt(str(run / "chkpts" / "last.ckpt"), full_state(us, cs), PARAMS)
    return run


@pytest.mark.parametrize("fmt", ["pt", "ckpt"])
def test_cli_from_run_directory(tmp_path, fmt):
    run = run_dir(tmp_path, fmt)
    out = tmp_path / "out"
    ck = run / "chkpts" / ("weights_best.pt" if fmt == "pt" else "last.ckpt")
    if fmt == "ckpt":
        os.remove(run / "params.yaml")      # a Lightning checkpoint carries its own params
    argv = ["--chkpt_path", str(ck), "--synthetic", "--length", "2", "--ddim", "--ddim_steps", "4", "--uncond_scale", "2.0",
            "--seed", "11", "--num_generate", "2", "--output_dir", str(out)]
    assert inference_sdf.main(argv) == 0
    npys = sorted(f for f in os.listdir(out) if f.endswith(".npy"))
    mids = sorted(f for f in os.listdir(out) if f.endswith(".mid"))
    assert len(npys) == 2 and len(mids) == 2
    a, b = (np.load(out / f) for f in npys)
    assert a.shape == (2, 2, 128, 128) and np.isfinite(a).all() and not np.array_equal(a, b)   # two songs, different noise
    # same seed -> same songs; no --seed -> a fresh seed is drawn and printed
    out2 = tmp_path / "out2"
    assert inference_sdf.main(argv[:-1] + [str(out2)]) == 0
    a2 = np.load(out2 / sorted(f for f in os.listdir(out2) if f.endswith(".npy"))[0])
    assert np.array_equal(a, a2)


def test_ddim_start_index_follows_the_reference(tmp_path):
    """with n_steps % ddim_steps != 0 'uniform' yields ddim_steps+1 entries; the reference still starts at ddim_steps-1."""
    from polyffusion_amd.sampler import DDIMSampler
    p, unet, ce = assemble()
    us, cs = states()
    unet.load_state_dict(us); ce.load_state_dict(cs)
    model = Polyffusion_SDF(inference_sdf.build_ldm(p, unet), "chord", "mix", chord_enc=ce)
    assert len(DDIMSampler(model.ldm, 30).time_steps) == 31
    args = argparse.Namespace(num_generate=1, autoreg=False, ddim=True, ddim_steps=30, ddim_discretize="uniform", ddim_eta=0.0,
                              repaint_n=1, uncond_scale=1.0)
    sampler, t_idx = inference_sdf.make_sampler(model, args, 1)
    assert t_idx == 29 and inference_sdf.Experiments("m", p, sampler, t_idx=t_idx).t_idx == 29


@pytest.mark.parametrize("autoreg", [False, True])
def test_sharded_generation_equals_unsharded(autoreg):
    """The multi-GPU mode on ONE GPU: the code each rank runs (generate_songs with its rank/world) executed for rank 0 and
    rank 1 of a 2-rank job; the concatenation must equal the 1-rank run of the same 4 songs."""
    p, unet, ce = assemble()
    us, cs = states()
    unet.load_state_dict(us); ce.load_state_dict(cs)
    unet.set_precision("bf16x3")
    model = Polyffusion_SDF(inference_sdf.build_ldm(p, unet), "chord", "mix", chord_enc=ce)
    chd = torch.from_numpy(synth.chords(2, 21)).cuda()
    cond, cond_mid = inference_sdf.encode_conditions(model, p, chd, None, autoreg)
    args = argparse.Namespace(num_generate=4, autoreg=autoreg, ddim=True, ddim_steps=2, ddim_discretize="uniform", ddim_eta=1.0,
                              repaint_n=1, uncond_scale=2.0)
    run = lambda r, w: inference_sdf.generate_songs(model, p, args, cond, cond_mid, None, None, 99, r, w)[0]
    full = run(0, 1)
    assert full.shape == ((4, 4, 2, 64, 128) if autoreg else (4, 2, 2, 128, 128))
    halves = torch.cat([run(0, 2), run(1, 2)])
    assert torch.equal(halves, torch.cat([run(0, 2), run(1, 2)]))       # bit-reproducible per shard
    assert (halves - full).abs().max() < 2e-4 and full.std() > 0        # and equal to the unsharded run up to tile-choice rounding
    assert not torch.equal(full[0], full[1])                            # songs differ (own noise streams)
    quarters = torch.cat([run(r, 4) for r in range(4)])
    assert (quarters - full).abs().max() < 2e-4
    # more ranks than songs: the surplus ranks hold zero rows
    args1 = argparse.Namespace(**{**vars(args), "num_generate": 1})
    assert inference_sdf.generate_songs(model, p, args1, cond, cond_mid, None, None, 99, 1, 2)[0].shape[0] == 0


def test_cli_hip_graph_gives_the_same_files(tmp_path):
    """--hip_graph (one captured reverse step replayed) changes nothing in the output: DDPM with inpainting-style blend, 6 steps via
    a short schedule in params.yaml, autoregressive over 2 segments (three sampling runs, each captured on its own)."""
    run = run_dir(tmp_path, "pt")
    p = yaml.safe_load((run / "params.yaml").read_text())
    p["n_steps"] = 6
    (run / "params.yaml").write_text(yaml.safe_dump(p))
    outs = []
    for flag in ([], ["--hip_graph"]):
        out = tmp_path / ("g" if flag else "e")
        argv = ["--chkpt_path", str(run / "chkpts" / "weights_best.pt"), "--synthetic", "--length", "2", "--autoreg", "--uncond_scale", "2.0",
                "--seed", "5", "--output_dir", str(out)] + flag
        assert inference_sdf.main(argv) == 0
        outs.append(np.load(out / sorted(f for f in os.listdir(out) if f.endswith(".npy"))[0]))
    assert outs[0].shape == (4, 2, 64, 128) and np.array_equal(outs[0], outs[1])


def test_cli_concat_blurry_params(tmp_path):
    """params.concat_blurry (ref:inference_sdf.py:797-803): the CLI blurs the given prmat2c and carries it as cond_concat through the
    batched multi-song driver; the result is what Experiments.predict(cond_concat=...) gives for the same seed."""
    from polyffusion_amd.sampler import SDFSampler
    params = dict(PARAMS, model_name="small_concat", in_channels=4, concat_blurry=True, concat_ratio=0.25)
    (tmp_path / "params.yaml").write_text(yaml.safe_dump(params))
    n = 2
    img = synth.prmat2c_image(31, n, 128)
    np.savez(tmp_path / "cond.npz", chord=synth.chords(n, 32), prmat2c=img)
    out = tmp_path / "out"
    argv = ["--custom_params_path", str(tmp_path / "params.yaml"), "--synthetic_weights", "--cond_npz", str(tmp_path / "cond.npz"),
            "--ddim", "--ddim_steps", "4", "--uncond_scale", "2.0", "--seed", "5", "--num_generate", "2", "--output_dir", str(out),
            "--precision", "f32"]     # the hand-built model below runs the library default (exact fp32 MFMA); the CLI's own default is `auto`
    assert inference_sdf.main(argv) == 0
    npys = sorted(f for f in os.listdir(out) if f.endswith(".npy"))
    assert len(npys) == 2
    a = np.load(out / npys[0])
    assert a.shape == (n, 2, 128, 128) and np.isfinite(a).all()
    # the same thing by hand
    args = inference_sdf.make_parser().parse_args(argv)
    p = inference_sdf.load_params(str(tmp_path / "params.yaml"))
    model = inference_sdf.load_model(p, args)
    cond = model._encode_chord(torch.from_numpy(synth.chords(n, 32)).cuda())
    cc = inference_sdf.get_blurry_image(torch.from_numpy(img).cuda(), 0.25)
    gen, _ = inference_sdf.generate_songs(model, p, args, cond, None, None, None, 5, cond_concat=cc)
    assert np.array_equal(gen[0].cpu().numpy(), a)
    # without the image to blur the CLI says so; with the shipped sdf_concat preset (in_channels 3) the 2-channel blurry image does not fit
    # the denoiser - in the reference (cat + conv shape error) as here
    np.savez(tmp_path / "cond2.npz", chord=synth.chords(n, 32))
    with pytest.raises(SystemExit, match="image to blur"):
        inference_sdf.main(argv[:4] + [str(tmp_path / "cond2.npz")] + argv[5:])


def test_cli_from_midi_and_inpaint_from_midi(tmp_path):
    """--from_midi / --inpaint_from_midi (ref:inference_sdf.py:569-575, 606-612): the reference's own example song is quantised, its chords
    extracted (label file written next to the outputs), its 8-bar segments condition the denoiser and supply the image to inpaint."""
    run = run_dir(tmp_path, "pt")
    song = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chord_example.mid")
    out = tmp_path / "out"
    argv = ["--chkpt_path", str(run / "chkpts" / "weights_best.pt"), "--from_midi", song, "--length", "3", "--ddim", "--ddim_steps", "4",
            "--uncond_scale", "2.0", "--seed", "3", "--output_dir", str(out)]
    assert inference_sdf.main(argv) == 0
    labels = open(out / "chords_extracted.out").read()
    assert labels == open(os.path.join(os.path.dirname(song), "chord_example.out")).read()       # the reference's expected labels for this file
    a = np.load(out / sorted(f for f in os.listdir(out) if f.endswith(".npy"))[0])
    assert a.shape == (3, 2, 128, 128) and np.isfinite(a).all()
    out2 = tmp_path / "out2"
    argv2 = ["--chkpt_path", str(run / "chkpts" / "weights_best.pt"), "--uncond_scale", "0", "--inpaint_type", "below", "--inpaint_from_midi", song,
             "--ddim", "--ddim_steps", "4", "--seed", "3", "--output_dir", str(out2)]
    assert inference_sdf.main(argv2) == 0
    b = np.load(out2 / sorted(f for f in os.listdir(out2) if f.endswith(".npy"))[0])
    assert b.shape == (12, 2, 128, 128) and os.path.exists(out2 / "chords_extracted_inpaint.out")       # the whole song: 12 segments
    # the kept region ("below") follows the song's own notes up to the last step's re-noising (q_sample at tau_0, as in the reference)
    from polyffusion_amd import datasample, midi_to_data
    p2c = datasample.DataSample(midi_to_data.get_data_for_single_midi(song)).get_whole_song_data()[0].numpy()
    mask = inference_sdf.get_mask(torch.from_numpy(p2c), "below").numpy()
    kept = np.abs((b - p2c) * mask)
    # (sigma of that re-noising: sqrt(1 - alpha_bar[1]) = 0.041)
    assert kept.max() < 0.25 and kept.sum() / mask.sum() < 0.05 and mask.mean() > 0.1
    assert (np.abs(b - p2c) * (1 - mask)).sum() / (1 - mask).sum() > 2 * kept.sum() / mask.sum()      # the generated part is free


def test_cli_sdf_pnotree_variant(tmp_path):
    """cond_type pnotree through the CLI (ref:inference_sdf.py:676-680, 756-760): the song's piano-tree grid conditions a small denoiser
    through the full-size PianoTreeEncoder (d_cond 2048), autoregressively; synthetic weights travel as one more packed blob."""
    params = dict(PARAMS, model_name="small_pnotree", d_cond=2048, cond_type="pnotree", use_enc=True)
    (tmp_path / "params.yaml").write_text(yaml.safe_dump(params))
    song = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chord_example.mid")
    out = tmp_path / "out"
    argv = ["--custom_params_path", str(tmp_path / "params.yaml"), "--synthetic_weights", "--from_midi", song, "--length", "2", "--autoreg",
            "--ddim", "--ddim_steps", "4", "--uncond_scale", "2.0", "--seed", "9", "--output_dir", str(out)]
    assert inference_sdf.main(argv) == 0
    a = np.load(out / sorted(f for f in os.listdir(out) if f.endswith(".npy"))[0])
    assert a.shape == (4, 2, 64, 128) and np.isfinite(a).all()
    with pytest.raises(SystemExit, match="piano-tree grid"):
        np.savez(tmp_path / "c.npz", chord=synth.chords(1, 1))
        inference_sdf.main(argv[:3] + ["--cond_npz", str(tmp_path / "c.npz")] + argv[5:])
