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


def torchrun(args, port, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PF_DIST_BACKEND="gloo", PF_LOCAL_DEVICE="0", PYTHONPATH=REPO)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=REPO)


def test_bench_two_ranks():
    out = torchrun(["bench.py", "--gpus", "2", "--steps", "4", "--warmup", "2", "--profile-steps", "0", "--fp32-steps", "2",
                    "--small-batch-steps", "0"], 29641)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 prints the one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["devices_seen"] == [0, 0] and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 32 and d["value"] > 0 and "cpu_baseline" not in d and d["fp32_mode"]["steps_per_s"] > 0
    # value = steps of all ranks / max-over-ranks time
    assert abs(d["value"] - 2 * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) / d["value"] < 1e-3


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
