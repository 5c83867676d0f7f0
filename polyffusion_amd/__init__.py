"""polyffusion_amd - MI355X-native denoising hot path of Polyffusion.

Host side (Python) mirrors the reference's operator interface for the path
(UNetModel / LatentDiffusion / SDFSampler / DDIMSampler / Polyffusion_SDF and
the inference_sdf CLI); all arithmetic runs in hand-written HIP kernels for
gfx950 behind the C ABI declared in ``include/pfhip.h`` (``csrc/``).  There is
no CPU or PyTorch-eager fallback: constructing any of the compute objects
without ``libpfhip.so`` / without a GPU raises.
"""
__version__ = "0.1.0"
