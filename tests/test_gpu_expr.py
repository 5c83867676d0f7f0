"""Row f4 (part): the batched evaluation drivers of the reference's expr.py, driven with synthetic validation batches (-m gpu)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from polyffusion_amd import _lib, expr, inference_sdf, midi, synth  # noqa: E402
from polyffusion_amd.arch import UNetConfig  # noqa: E402
from polyffusion_amd.model_sdf import Polyffusion_SDF  # noqa: E402
from polyffusion_amd.params import Params  # noqa: E402
from polyffusion_amd.sampler import SDFSampler  # noqa: E402
from polyffusion_amd.weights import synth_chord_encoder_state, synth_unet_state  # noqa: E402

PARAMS = dict(model_name="small_chd", in_channels=2, out_channels=2, channels=32, attention_levels=[1], n_res_blocks=1,
              channel_multipliers=[1, 2], n_heads=2, tf_layers=1, d_cond=32, linear_start=0.00085, linear_end=0.012, n_steps=1000,
              latent_scaling_factor=0.18215, img_h=128, img_w=128, cond_type="chord", cond_mode="mix", use_enc=True,
              chd_n_step=32, chd_input_dim=36, chd_z_input_dim=32, chd_hidden_dim=64, chd_z_dim=32)


@pytest.fixture(scope="module")
def setup():
    _lib.require_gpu()
    p = Params(PARAMS)
    unet = inference_sdf.build_unet(p)
    ce, _ = inference_sdf.build_encoders(p)
    unet.load_state_dict(synth_unet_state(UNetConfig.from_params(p), 3))
    ce.load_state_dict(synth_chord_encoder_state(3, 36, 64, 32))
    model = Polyffusion_SDF(inference_sdf.build_ldm(p, unet), "chord", "mix", chord_enc=ce)
    ex = inference_sdf.Experiments("small_chd", p, SDFSampler(model.ldm, seed=9), t_idx=2)   # 3 reverse steps per run
    return model, ex


def batches(n, size=3):
    for i in range(n):
        img = torch.from_numpy((synth.prmat2c_image(40 + i, size, 128) > 0.5).astype(np.float32)).cuda()
        yield img, None, torch.from_numpy(synth.chords(size, 50 + i)).cuda(), torch.from_numpy(synth.prmat(size, 60 + i)).cuda()


def test_drivers(setup, tmp_path):
    model, ex = setup
    out = str(tmp_path)
    g = expr.prompt_generation(ex, batches(3), 2, out, check_integrity=False)   # num = 2 of the 3 batches; a random-weight model emits no notes,
    # and with no notes the reference's integrity ratio is 0/0 (ZeroDivisionError there and here)
    assert g.shape == (6, 2, 128, 128) and torch.isfinite(g).all() and os.path.getsize(f"{out}/uncond.mid") > 20
    g = expr.acc_arrangement(ex, batches(2), 2, out)
    assert g.shape == (6, 2, 128, 128) and os.path.exists(f"{out}/acc_arr.mid")
    src = next(batches(1))[0]
    g = expr.inpaint_bars(ex, batches(1), 1, out)
    assert g.shape == (3, 2, 64, 128) and os.path.exists(f"{out}/inp_bars.mid")
    g = expr.chd_conditioning(ex, model, batches(2), 2, out, uncond_scale=2.0)
    assert g.shape == (6, 2, 128, 128) and np.load(f"{out}/chd[2.0].npy").shape == (2, 3, 32, 36) and os.path.exists(f"{out}/chd_cond[2.0].mid")
    # the known region of an inpainting run converges to the original where the mask keeps it (bars 0-1 and 6-7)
    full = ex.inpaint(src, "bars", src, None, uncond_scale=0.0, bar_list=[2, 3, 4, 5], no_output=True)
    assert (full[:, :, :32] - src[:, :, :32]).abs().max() < 0.2 and 0.0 <= midi.check_prmat2c_integrity(src) <= 1.0
    with pytest.raises(ZeroDivisionError):
        midi.check_prmat2c_integrity(torch.zeros(1, 2, 16, 128))
    assert set(expr.DRIVERS) == {"uncond", "inp_below", "inp_bars", "chd_cond", "txt_cond"}


def test_song_batches(tmp_path):
    paths = []
    for i in range(2):   # the seeded song dictionaries the datasample fixture uses (reference data-dictionary format)
        path = str(tmp_path / f"song{i}.npz")
        np.savez(path, **synth.song_data(100 + i, 12 + 8 * i))
        paths.append(path)
    got = list(expr.song_batches(paths, batch_size=4))
    assert all(b[0].shape[1:] == (2, 128, 128) and b[2].shape[1:] == (32, 36) and b[3].shape[1:] == (128, 128) for b in got)
    assert sum(b[0].shape[0] for b in got) >= 2 and all(b[0].is_cuda for b in got) and all(b[0].shape[0] <= 4 for b in got)
