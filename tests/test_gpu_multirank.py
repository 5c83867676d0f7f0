"""The N-rank code paths executed with TWO processes on ONE GPU (-m gpu): `bench.py --gpus 2` and the sharded CLI under
`torch.distributed.run`.  RCCL refuses two ranks on one device, so the processes talk over gloo (PF_DIST_BACKEND) and both use device 0
(PF_LOCAL_DEVICE) - everything else (rank / offset bookkeeping, blob broadcast from rank 0, barriers, max-over-ranks timing, the
end-of-run gather, rank-0-only output) is the code that runs on an 8-GPU node."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def torchrun(args, port, timeout=600, nproc=2):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PF_DIST_BACKEND="gloo", PF_LOCAL_DEVICE="0", PYTHONPATH=REPO)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=REPO)


def test_bench_two_ranks():
    out = torchrun(["bench.py", "--gpus", "2", "--steps", "4", "--warmup", "2", "--profile-steps", "0", "--fp32-steps", "2",
                    "--f16x3-steps", "2", "--small-batch-steps", "0"], 29641)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 prints the one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["devices_seen"] == [0, 0] and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 32 and d["value"] > 0 and "cpu_baseline" not in d and d["fp32_mode"]["steps_per_s"] > 0
    # the fp16-split leg: a second model in libpfhip_f16.so on every rank (its weights broadcast like the first's), both splits' distance to f32
    h = d["f16x3_mode"]
    assert h["steps_per_s"] > 0 and h["eps_rel_diff_vs_f32_mode"]["f16x3"] < h["eps_rel_diff_vs_f32_mode"]["bf16x3"] < 1e-3
    # value = steps of all ranks / max-over-ranks time
    assert abs(d["value"] - 2 * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) / d["value"] < 1e-3
    # every rank's own step time is in the line (a straggler on a real node shows as max >> min); the host-clock figure bounds them
    pr = d["per_rank_ms_per_step"]
    assert len(pr["all"]) == 2 and pr["min"] == min(pr["all"]) and pr["max"] == max(pr["all"]) and 0 < pr["max"] <= d["ms_per_step"] * 1.05


def test_bench_eight_ranks_on_one_device():
    """The shape the driver's scaling run has - `--gpus 8`, one rank per GPU - with all eight ranks on device 0 over gloo: rank / offset
    bookkeeping for eight shards, the weight broadcast to seven receivers, barriers, max-over-ranks timing and the per-rank step times."""
    out = torchrun(["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1", "--windows", "2", "--profile-steps", "0", "--fp32-steps", "0",
                    "--f16x3-steps", "0", "--small-batch-steps", "0"], 29649, nproc=8, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["devices_seen"] == [0] * 8 and d["config"]["global_batch"] == 128
    assert len(d["per_rank_ms_per_step"]["all"]) == 8 and d["per_rank_ms_per_step"]["max"] <= d["ms_per_step"] * 1.05
    assert abs(d["value"] - 8 * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) / d["value"] < 1e-3 and "cpu_baseline" not in d


def test_bench_bare_command_spawns_its_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (how a driver that mirrors its 1-GPU command would call it): bench.py re-executes
    itself under torch.distributed.run, one rank per GPU, and still prints exactly one JSON line with both ranks in it."""
    env = dict(os.environ, PF_DIST_BACKEND="gloo", PF_LOCAL_DEVICE="0", PYTHONPATH=REPO)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--windows", "2", "--profile-steps", "0",
                          "--fp32-steps", "0", "--small-batch-steps", "0"], capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["devices_seen"] == [0, 0] and d["config"]["global_batch"] == 32
    assert d["windows"] == 2 and len(d["windows_ms_per_step"]) == 2 and d["config"]["weight_broadcast_s"] >= 0


def _small_run_dir(tmp_path):
    from ckpt_fixture import full_state, write_legacy_pt
    from polyffusion_amd.arch import UNetConfig
    from polyffusion_amd.weights import synth_chord_encoder_state, synth_unet_state
    params = dict(model_name="small_chd", in_channels=2, out_channels=2, channels=32, attention_levels=[1], n_res_blocks=1,
                  channel_multipliers=[1, 2], n_heads=2, tf_layers=1, d_cond=32, linear_start=0.00085, linear_end=0.012, n_steps=1000,
                  latent_scaling_factor=0.18215, img_h=128, img_w=128, cond_type="chord", cond_mode="mix", use_enc=True,
                  chd_n_step=32, chd_input_dim=36, chd_z_input_dim=32, chd_hidden_dim=64, chd_z_dim=32)
    run = tmp_path / "run"
    (run / "chkpts").mkdir(parents=True)
    (run / "params.yaml").write_text(yaml.safe_dump(params))
    write_legacy_pt(str(run / "chkpts" / "weights_best.pt"),
                    full_state(synth_unet_state(UNetConfig.from_params(params), 3), synth_chord_encoder_state(3, 36, 64, 32)))
    return run


def test_cli_autoreg_four_ranks_equal_one_rank(tmp_path):
    """BASELINE config 5's sharding with FOUR processes: `--autoreg` on 8 songs of 2 segments = 2 songs per rank, each rank running the
    2B-1 = 3 sequential half-overlapping runs on its own batch of 2 (predict_songs), rank 0 gathering and writing all 8 - against the
    one-process run of the same command (one batch of 8): same noise per song (Philox keyed by the global song index), equal up to
    the tile-choice rounding of different per-GPU batch sizes."""
    run = _small_run_dir(tmp_path)
    argv = ["--chkpt_path", str(run / "chkpts" / "weights_best.pt"), "--synthetic", "--length", "2", "--ddim", "--ddim_steps", "4",
            "--uncond_scale", "1.0", "--seed", "21", "--num_generate", "8", "--autoreg"]
    out4 = torchrun(["-m", "polyffusion_amd.inference_sdf"] + argv + ["--output_dir", str(tmp_path / "four")], 29647, nproc=4)
    assert out4.returncode == 0, out4.stdout[-2000:] + out4.stderr[-2000:]
    env = dict(os.environ, PYTHONPATH=REPO)
    out1 = subprocess.run([sys.executable, "-m", "polyffusion_amd.inference_sdf"] + argv + ["--output_dir", str(tmp_path / "one")],
                          capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert out1.returncode == 0, out1.stdout[-2000:] + out1.stderr[-2000:]
    f4 = sorted(f for f in os.listdir(tmp_path / "four") if f.endswith(".npy"))
    f1 = sorted(f for f in os.listdir(tmp_path / "one") if f.endswith(".npy"))
    assert len(f1) == len(f4) == 8
    key = lambda f: int(f[:-4].rsplit("_", 1)[1])            # ..._<song index>.npy
    for a, b in zip(sorted(f1, key=key), sorted(f4, key=key)):
        x, y = np.load(tmp_path / "one" / a), np.load(tmp_path / "four" / b)
        assert x.shape == y.shape == (4, 2, 64, 128) and np.isfinite(x).all() and np.abs(x - y).max() < 2e-4   # 2 segments -> 4 half images
    assert out4.stdout.count("model_label") == 1 and "on 4 GPU(s)" in out4.stdout


def test_cli_two_ranks_equal_one_rank(tmp_path):
    from ckpt_fixture import full_state, write_legacy_pt
    from polyffusion_amd.arch import UNetConfig
    from polyffusion_amd.weights import synth_chord_encoder_state, synth_unet_state
    params = dict(model_name="small_chd", in_channels=2, out_channels=2, channels=32, attention_levels=[1], n_res_blocks=1,
                  channel_multipliers=[1, 2], n_heads=2, tf_layers=1, d_cond=32, linear_start=0.00085, linear_end=0.012, n_steps=1000,
                  latent_scaling_factor=0.18215, img_h=128, img_w=128, cond_type="chord", cond_mode="mix", use_enc=True,
                  chd_n_step=32, chd_input_dim=36, chd_z_input_dim=32, chd_hidden_dim=64, chd_z_dim=32)
    run = tmp_path / "run"
    (run / "chkpts").mkdir(parents=True)
    (run / "params.yaml").write_text(yaml.safe_dump(params))
    write_legacy_pt(str(run / "chkpts" / "weights_best.pt"),
                    full_state(synth_unet_state(UNetConfig.from_params(params), 3), synth_chord_encoder_state(3, 36, 64, 32)))
    argv = ["--chkpt_path", str(run / "chkpts" / "weights_best.pt"), "--synthetic", "--length", "2", "--ddim", "--ddim_steps", "4",
            "--uncond_scale", "2.0", "--seed", "11", "--num_generate", "3"]
    out2 = torchrun(["-m", "polyffusion_amd.inference_sdf"] + argv + ["--output_dir", str(tmp_path / "two")], 29643)
    assert out2.returncode == 0, out2.stdout[-2000:] + out2.stderr[-2000:]
    env = dict(os.environ, PYTHONPATH=REPO)
    out1 = subprocess.run([sys.executable, "-m", "polyffusion_amd.inference_sdf"] + argv + ["--output_dir", str(tmp_path / "one")],
                          capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert out1.returncode == 0, out1.stdout[-2000:] + out1.stderr[-2000:]
    f2 = sorted(f for f in os.listdir(tmp_path / "two") if f.endswith(".npy"))
    f1 = sorted(f for f in os.listdir(tmp_path / "one") if f.endswith(".npy"))
    assert len(f1) == len(f2) == 3                            # rank 0 alone writes the three songs
    for a, b in zip(f1, f2):                                  # 3 songs over 2 ranks = shards of 2 + 1 vs one batch of 3: tile-choice rounding only
        x, y = np.load(tmp_path / "one" / a), np.load(tmp_path / "two" / b)
        assert x.shape == y.shape == (2, 2, 128, 128) and np.abs(x - y).max() < 2e-4
    assert out2.stdout.count("model_label") == 1              # only rank 0 talks


@pytest.mark.parametrize("nproc", [2, 4])
def test_shard_invariance_bit_for_bit_at_equal_per_gpu_batch(tmp_path, nproc):
    """config 4's shape - sdf_txt, 16 samples per rank - on 2 and 4 ranks (one device, gloo) against ONE process that
    evaluates the same global batch in chunks of 16: same per-launch batch -> same tile choices -> the images are BIT-IDENTICAL
    (tools/shard_invariance.py; weights by load_model's broadcast, noise keyed by the global sample index)."""
    many, one = str(tmp_path / "many.npy"), str(tmp_path / "one.npy")
    out = torchrun(["tools/shard_invariance.py", "--per_rank", "16", "--steps", "3", "--out", many], 29651 + nproc, nproc=nproc, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    env = dict(os.environ, PYTHONPATH=REPO)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    ref = subprocess.run([sys.executable, "tools/shard_invariance.py", "--chunks", str(nproc), "--per_rank", "16", "--steps", "3", "--out", one],
                         capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert ref.returncode == 0, ref.stdout[-2000:] + ref.stderr[-2000:]
    a, b = np.load(many), np.load(one)
    assert a.shape == b.shape == (16 * nproc, 2, 128, 128) and np.isfinite(a).all()
    assert np.array_equal(a.view(np.int32), b.view(np.int32)), f"max-abs-diff {np.abs(a - b).max()}"
    assert not np.array_equal(a[:16], a[16:32])            # different samples really are different
