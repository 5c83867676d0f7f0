"""Build libpfhip.so (gfx950) in-tree with hipcc.  No JIT cache, no setuptools: the .so must sit
next to the sources so it travels to the GPU box with the repo snapshot."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["conv_mfma.hip", "conv_bf16x3.hip", "gemm_planes_bf3.hip", "mlp_fused_bf3.hip", "attention.hip", "attention_bf3.hip", "norm_stats.hip", "small_kernels.hip", "encoders.hip", "comm.hip", "unet.hip"]
LIB = os.path.join(HERE, "libpfhip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + (["-DPF_TRACE"] if os.environ.get("PF_TRACE") else [])
# per-file additions.  attention_bf3: the softmax of the 256-query kernel is dealt out between MFMAs one single-issue instruction at a
# time; SLP vectorisation turns its scalar adds into v_pk_add_f32 + v_mov pairs, which cost more issue slots beside MFMAs, not fewer.
EXTRA_FLAGS = {"attention_bf3.hip": ["-fno-slp-vectorize"]}


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    # every header is a dependency of every object (conv_common.h carries the shared epilogue and the asm helpers)
    hdrs = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")] + [os.path.join(HERE, "..", "include", "pfhip.h")]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append(cmd)
    if jobs:   # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            for rc in ex.map(lambda c: subprocess.run(c).returncode, jobs):
                if rc != 0:
                    raise subprocess.CalledProcessError(rc, "hipcc")
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
