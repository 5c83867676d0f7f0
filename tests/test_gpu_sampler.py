"""Sampler-loop parity (-m gpu): trajectories with the reference's injected noise tape, the
autoregressive schedule against the oracle, and shard-invariance of on-device noise."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sampler_ref, unet_ref  # noqa: E402
from polyffusion_amd import _lib  # noqa: E402
from polyffusion_amd.arch import UNetConfig  # noqa: E402
from polyffusion_amd.inference_sdf import Experiments, get_autoreg_data  # noqa: E402
from polyffusion_amd.sampler import DDIMSampler, SDFSampler  # noqa: E402
from polyffusion_amd.unet import LatentDiffusion, UNetModel  # noqa: E402
from polyffusion_amd.weights import synth_unet_state  # noqa: E402

SMALL = UNetConfig(in_channels=2, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,),
                   channel_multipliers=(1, 2), n_heads=2, tf_layers=1, d_cond=32)
LIN = (0.00085, 0.012)


@pytest.fixture(scope="module", params=["f32", "bf16x3"])
def ldm(request):
    """Both arithmetic modes: exact fp32 MFMA and the error-compensated bf16 split (the default of bench.py)."""
    _lib.require_gpu()
    m = UNetModel(in_channels=2, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,),
                  channel_multipliers=(1, 2), n_heads=2, tf_layers=1, d_cond=32, img_h=16, img_w=16)
    m.load_state_dict(synth_unet_state(SMALL, 0))
    m.set_precision(request.param)
    return LatentDiffusion(m, None, 0.18215, 1000, *LIN)


class Tape:
    def __init__(self, arr):
        self.arr, self.i = arr, 0

    def __call__(self, shape):
        a = torch.from_numpy(self.arr[self.i])
        self.i += 1
        assert tuple(a.shape) == tuple(shape)
        return a


def test_trajectories_vs_reference_golden(ldm, golden):
    g = golden("trajectories.npz")
    cond, start, orig, mask = (torch.from_numpy(g[k]).cuda() for k in ("cond", "start_noise", "orig", "mask"))
    uc = -torch.ones(2, 1, 32).cuda()
    z = torch.zeros_like(start)
    # (a) DDPM generate path (orig = mask = 0), 10 steps; two draws per step, none at step 0
    tape = Tape(g["ddpm_gen_tape"])
    s = SDFSampler(ldm, noise_fn=tape)
    out = s.paint(s.q_sample(z, 9, start), cond, 9, orig=z, mask=z, orig_noise=start, uncond_scale=1.0, uncond_cond=uc)
    assert tape.i == 18
    assert np.abs(out.cpu().numpy() - g["ddpm_gen_out"]).max() < 1e-3
    # (b) DDPM inpaint, CFG 3.0, repaint_n = 2
    tape = Tape(g["ddpm_inp_tape"])
    s = SDFSampler(ldm, noise_fn=tape)
    out = s.paint(s.q_sample(orig, 5, start), cond, 5, orig=orig, mask=mask, orig_noise=start, uncond_scale=3.0,
                  uncond_cond=uc, repaint_n=2)
    assert tape.i == len(g["ddpm_inp_tape"])
    assert np.abs(out.cpu().numpy() - g["ddpm_inp_out"]).max() < 1e-3
    # (c) DDIM eta 0, CFG 5, masked
    d = DDIMSampler(ldm, 10, "uniform", 0.0)
    out = d.paint(d.q_sample(orig, 4, start), cond, 4, orig=orig, mask=mask, orig_noise=start, uncond_scale=5.0, uncond_cond=uc)
    assert np.abs(out.cpu().numpy() - g["ddim_out"]).max() < 1e-3
    # (d) DDIM eta 1 quad (draws noise), unconditional (scale 0)
    tape = Tape(g["ddim_eta1_tape"])
    d = DDIMSampler(ldm, 10, "quad", 1.0, noise_fn=tape)
    out = d.paint(d.q_sample(z, 9, start), cond, 9, orig=z, mask=z, orig_noise=start, uncond_scale=0.0, uncond_cond=uc)
    assert tape.i == len(g["ddim_eta1_tape"])
    assert np.abs(out.cpu().numpy() - g["ddim_eta1_out"]).max() < 1e-3


def test_autoreg_predict_vs_oracle(ldm):
    """Experiments.predict(autoreg=True): 2B-1 sequential half-overlapping runs, in-place orig/mask edits."""
    B, T = 3, 3  # 3 segments, 4 reverse steps each (t_idx = 3)
    rng = np.random.Generator(np.random.PCG64(8))
    cond = torch.from_numpy(rng.standard_normal((B, 1, 32)).astype(np.float32))
    noise = torch.from_numpy(rng.standard_normal((B, 2, 16, 16)).astype(np.float32))
    draws = rng.standard_normal((200, 1, 2, 16, 16)).astype(np.float32)
    w = unet_ref.to_torch(synth_unet_state(SMALL, 0))
    model = lambda x, t, c: unet_ref.unet_forward(w, SMALL, x, t, c)
    ref_s = sampler_ref.SDFSamplerRef(model, 1000, *LIN, noise_fn=Tape(draws))
    cond_mid = sampler_ref.get_autoreg_data(cond, 1) if False else cond.roll(-1, 0)  # any per-segment mid cond
    ref = sampler_ref.predict(ref_s, cond, 32, [B, 2, 16, 16], T, noise, cond_mid=cond_mid, uncond_scale=2.0, autoreg=True)
    s = SDFSampler(ldm, noise_fn=Tape(draws))
    params = dict(out_channels=2, img_h=16, img_w=16, d_cond=32, n_steps=1000)
    ex = Experiments("small", params, s, t_idx=T)
    got = ex.predict(cond.cuda(), cond_mid.cuda(), uncond_scale=2.0, autoreg=True, noise=noise.cuda())
    assert got.shape == (2 * B, 2, 8, 16)
    assert (got.cpu() - ref).abs().max() < 1e-3
    a = torch.arange(24.).view(3, 4, 2)
    assert torch.equal(get_autoreg_data(a, 1), sampler_ref.get_autoreg_data(a, 1))


def test_on_device_noise_is_shard_invariant(ldm):
    """Batch sharded over ranks (sample_offset) reproduces the unsharded batch bit-for-bit: the multi-GPU contract."""
    B = 4
    cond = torch.from_numpy(np.random.Generator(np.random.PCG64(1)).standard_normal((B, 1, 32)).astype(np.float32)).cuda()
    z = torch.zeros(B, 2, 16, 16).cuda()

    def run(lo, hi):
        s = SDFSampler(ldm, seed=77, sample_offset=lo)
        x = s.randn((hi - lo, 2, 16, 16), z.device)
        return s.paint(x, cond[lo:hi].contiguous(), 5, orig=z[lo:hi], mask=z[lo:hi])

    full = run(0, B)
    assert torch.isfinite(full).all() and full.std() > 0
    # same batch size per shard keeps the tile choice (and thus rounding) identical
    halves = torch.cat([run(0, 2), run(2, 4)])
    whole2 = torch.cat([run(0, 2), run(2, 4)])
    assert torch.equal(halves, whole2)
    assert (halves - full).abs().max() < 1e-4


def test_graph_replay_is_bit_identical_to_the_eager_loop(ldm):
    """hipGraph mode: one captured step replayed with device-resident step state (table row, time step, draw counter) must give
    exactly what the eager loop gives - 10 DDPM steps with inpainting + CFG, a generate-path run without orig, and DDIM eta 0."""
    rng = np.random.Generator(np.random.PCG64(31))
    B = 3
    cond = torch.from_numpy(rng.standard_normal((B, 1, 32)).astype(np.float32)).cuda()
    uc = -torch.ones(B, 1, 32).cuda()
    orig = torch.from_numpy((rng.random((B, 2, 16, 16)) < 0.1).astype(np.float32)).cuda()
    mask = torch.zeros(B, 2, 16, 16).cuda()
    mask[:, :, :6] = 1

    def ddpm(graph, with_orig, scale):
        s = SDFSampler(ldm, seed=42, sample_offset=5, graph=graph)
        x = s.randn((B, 2, 16, 16), cond.device)
        out = s.paint(x, cond, 9, orig=orig if with_orig else None, mask=mask if with_orig else None, uncond_scale=scale, uncond_cond=uc)
        return out, s._draws

    for with_orig, scale in ((True, 2.0), (False, 1.0), (True, 0.0)):
        (a, da), (b, db) = ddpm(False, with_orig, scale), ddpm(True, with_orig, scale)
        assert da == db                      # the host-side draw counter advanced identically
        assert torch.equal(a, b), (with_orig, scale, (a - b).abs().max().item())
        assert torch.isfinite(a).all() and a.std() > 0

    def ddim(graph):
        d = DDIMSampler(ldm, 10, "uniform", 0.0, seed=42, graph=graph)
        x = d.randn((B, 2, 16, 16), cond.device)
        return d.paint(d.q_sample(orig, 6, x), cond, 6, orig=orig, mask=mask, orig_noise=x, uncond_scale=3.0, uncond_cond=uc)

    assert torch.equal(ddim(False), ddim(True))

    def ddim_eta(graph):   # eta = 1: every step draws one noise tensor
        d = DDIMSampler(ldm, 10, "quad", 1.0, seed=43, sample_offset=2, graph=graph)
        x = d.randn((B, 2, 16, 16), cond.device)
        return d.paint(x, cond, 7, uncond_scale=1.0, uncond_cond=uc), d._draws

    (a, da), (b, db) = ddim_eta(False), ddim_eta(True)
    assert da == db == 9 and torch.equal(a, b)
    # a second paint() on the same sampler object continues the draw sequence in both modes
    s1, s2 = SDFSampler(ldm, seed=1, graph=False), SDFSampler(ldm, seed=1, graph=True)
    for s in (s1, s2):
        s.out = [s.paint(s.randn((B, 2, 16, 16), cond.device), cond, 3, orig=orig, mask=mask) for _ in range(2)]
    assert torch.equal(s1.out[0], s2.out[0]) and torch.equal(s1.out[1], s2.out[1]) and not torch.equal(s1.out[0], s1.out[1])
    assert s2.graph_captures == 1            # the second paint() replayed the step captured by the first


def test_captured_step_is_reused_and_dropped_when_the_workspace_moves(ldm):
    """The captured step is cached per (shapes, guidance scale, ...) across paint() calls - the autoregressive schedule calls paint()
    2B-1 times - with the inputs copied into static buffers; a cached graph is dropped when the UNet's workspace was re-allocated
    (a larger batch in between), because its nodes hold the old address.  Results stay bit-identical to the eager loop throughout,
    and a returned image is not a view of the static buffer."""
    rng = np.random.Generator(np.random.PCG64(77))
    B = 2
    def inputs(k):
        g = torch.Generator().manual_seed(k)
        cond = torch.randn(B, 1, 32, generator=g).cuda()
        orig = (torch.rand(B, 2, 16, 16, generator=g) < 0.1).float().cuda()
        mask = torch.zeros(B, 2, 16, 16).cuda()
        mask[:, :, : 4 + k] = 1
        return cond, orig, mask
    uc = -torch.ones(B, 1, 32).cuda()
    eager, graph = DDIMSampler(ldm, 10, "uniform", 0.0, seed=9, graph=False), DDIMSampler(ldm, 10, "uniform", 0.0, seed=9, graph=True)
    outs = {False: [], True: []}
    for k in range(3):
        cond, orig, mask = inputs(k)
        for smp in (eager, graph):
            x = smp.randn((B, 2, 16, 16), cond.device)
            outs[smp.graph].append(smp.paint(smp.q_sample(orig, 5 + k, x), cond, 5 + k, orig=orig, mask=mask, orig_noise=x, uncond_scale=2.0,
                                             uncond_cond=uc))
        if k == 1:   # a larger batch re-allocates the UNet workspace
            ldm(torch.zeros(B + 5, 2, 16, 16).cuda(), torch.zeros(B + 5, dtype=torch.long).cuda(), torch.zeros(B + 5, 1, 32).cuda())
    for a, b in zip(outs[False], outs[True]):
        assert torch.equal(a, b)
    assert not torch.equal(outs[True][0], outs[True][1])         # distinct results: no aliasing of the static buffer
    assert graph.graph_captures == 2                              # captured once, re-captured after the workspace moved
    # DDPM sampler, generate path without known region, then with one: two keys, one capture each
    sd = SDFSampler(ldm, seed=3, graph=True)
    se = SDFSampler(ldm, seed=3, graph=False)
    cond, orig, mask = inputs(0)
    for smp in (se, sd):
        smp.res = [smp.paint(smp.randn((B, 2, 16, 16), cond.device), cond, 4),
                   smp.paint(smp.randn((B, 2, 16, 16), cond.device), cond, 4, orig=orig, mask=mask),
                   smp.paint(smp.randn((B, 2, 16, 16), cond.device), cond, 6)]
    assert all(torch.equal(a, b) for a, b in zip(se.res, sd.res)) and sd.graph_captures == 2


def test_comm_single_rank_broadcast():
    """pf_comm_* (RCCL reached directly through dlopen): a 1-rank communicator initialises from its own unique id and
    a broadcast from root 0 leaves the buffer unchanged.  (Multi-rank use: polyffusion_amd.dist.broadcast_blob with PF_COMM_DIRECT=1.)"""
    import ctypes as C
    lib = _lib.load()
    uid = (C.c_char * 128)()
    _lib.check(lib.pf_comm_unique_id(uid), "pf_comm_unique_id")
    comm = C.c_void_p()
    _lib.check(lib.pf_comm_init(uid, 0, 1, C.byref(comm)), "pf_comm_init")
    buf = torch.arange(4096, dtype=torch.float32, device="cuda")
    _lib.check(lib.pf_comm_bcast(comm, buf.data_ptr(), buf.numel() * 4, 0, _lib.current_stream()), "pf_comm_bcast")
    torch.cuda.synchronize()
    assert torch.equal(buf.cpu(), torch.arange(4096, dtype=torch.float32))
    _lib.check(lib.pf_comm_destroy(comm), "pf_comm_destroy")


def test_direct_broadcast_path_single_rank(monkeypatch):
    """dist.broadcast_blob(PF_COMM_DIRECT=1): TCPStore id exchange + pf_comm_init + pf_comm_bcast, executed with world size 1."""
    from polyffusion_amd import dist as pfdist
    monkeypatch.setenv("PF_COMM_DIRECT", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29731")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    blob = torch.arange(1000, dtype=torch.float32, device="cuda")
    out = pfdist.broadcast_blob(blob, 0)
    torch.cuda.synchronize()
    assert out is blob and float(blob.sum()) == 499500.0
