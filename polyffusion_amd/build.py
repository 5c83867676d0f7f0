"""Build libpfhip.so (gfx950) in-tree with hipcc.  No JIT cache, no setuptools: the .so must sit
next to the sources so it travels to the GPU box with the repo snapshot."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["conv_mfma.hip", "conv_bf16x3.hip", "conv_wino.hip", "gemm_planes_bf3.hip", "mlp_fused_bf3.hip", "attention.hip", "attention_bf3.hip", "norm_stats.hip", "small_kernels.hip", "encoders.hip", "comm.hip", "unet.hip"]
LIB = os.path.join(HERE, "libpfhip.so")
# variants: the same sources compiled with another element type for the split-precision kernels (csrc/pf_internal.h, PF_X3_F16);
# objects get a suffix, the library another name, the stamp its own keys.  PF_X3=f16 in the environment selects it at load (_lib.py).
VARIANTS = {"": [], "f16": ["-DPF_X3_F16"]}


def lib_path(variant: str = "") -> str:
    return os.path.join(HERE, f"libpfhip_{variant}.so" if variant else "libpfhip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + (["-DPF_TRACE"] if os.environ.get("PF_TRACE") else [])
# per-file additions.  attention_bf3: the softmax of the 256-query kernel is dealt out between MFMAs one single-issue instruction at a
# time; SLP vectorisation turns its scalar adds into v_pk_add_f32 + v_mov pairs, which cost more issue slots beside MFMAs, not fewer.
EXTRA_FLAGS = {"attention_bf3.hip": ["-fno-slp-vectorize"]}


STAMP = os.path.join(CSRC, ".build_stamp.json")


def _digest(paths, extra=()) -> str:
    """sha256 over the CONTENT of the files and the command line: an object is current only if it was compiled from exactly
    these bytes with exactly these flags (mtimes say nothing on a snapshot that ships prebuilt objects)."""
    import hashlib
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    for e in extra:
        h.update(str(e).encode())
        h.update(b"\0")
    return h.hexdigest()


def _read_stamp() -> dict:
    import json
    try:
        with open(STAMP) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def _write_stamp(st: dict) -> None:
    import json
    with open(STAMP + ".new", "w") as f:
        json.dump(st, f, indent=1, sort_keys=True)
    os.replace(STAMP + ".new", STAMP)


def build(force: bool = False, verbose: bool = True, variant: str = "") -> str:
    """Compile what is not provably current and link the library of one variant; returns its path.  See build_all."""
    return build_all(force, verbose, (variant,))[0]


# slowest translation units first, so that the pool's tail is short
_SLOW_FIRST = ["conv_bf16x3.hip", "conv_wino.hip", "attention_bf3.hip", "mlp_fused_bf3.hip", "conv_mfma.hip", "gemm_planes_bf3.hip"]


def build_all(force: bool = False, verbose: bool = True, variants=("", "f16")) -> list:
    """Compile what is not provably current, for every variant asked for, in ONE pool of hipcc processes, then link each library.
    `force` (or PF_FORCE_BUILD=1 in the environment) recompiles every translation unit.  Otherwise an object is reused only when the
    stamp file records, for it, the digest of the source + every header + the flags it would be compiled with now; a library
    likewise against the digests of its objects.  Returns the library paths in the order of `variants`."""
    force = force or os.environ.get("PF_FORCE_BUILD", "") not in ("", "0")
    # every header is a dependency of every object (conv_common.h carries the shared epilogue and the asm helpers)
    hdrs = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")] + [os.path.join(HERE, "..", "include", "pfhip.h")]
    stamp = _read_stamp()
    objs, jobs, want = {v: [] for v in variants}, [], {}
    for variant in variants:
        sfx = f".{variant}" if variant else ""
        for src0 in SOURCES:
            s = os.path.join(CSRC, src0)
            src = src0 + sfx                      # stamp key
            o = os.path.join(CSRC, src0.replace(".hip", sfx + ".o"))
            objs[variant].append(o)
            flags = FLAGS + EXTRA_FLAGS.get(src0, []) + VARIANTS[variant]
            want[src] = _digest([s] + hdrs, flags)
            if force or not os.path.exists(o) or stamp.get(src) != want[src]:
                cmd = [HIPCC] + flags + ["-c", s, "-o", o]
                if verbose:
                    print(" ".join(cmd), flush=True)
                jobs.append((src, cmd, variant, _SLOW_FIRST.index(src0) if src0 in _SLOW_FIRST else len(_SLOW_FIRST)))
    jobs.sort(key=lambda j: j[3])
    if jobs:   # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            rcs = list(ex.map(lambda j: subprocess.run(j[1]).returncode, jobs))
        for (src, _, _, _), rc in zip(jobs, rcs):
            if rc == 0:
                stamp[src] = want[src]
            else:
                stamp.pop(src, None)
        _write_stamp(stamp)
        if any(rcs):
            raise subprocess.CalledProcessError(max(rcs), "hipcc")
    elif verbose:
        print(f"polyffusion_amd.build: {len(SOURCES)} objects per variant current by content digest (PF_FORCE_BUILD=1 recompiles)", flush=True)
    libs = []
    for variant in variants:
        lib, lib_key = lib_path(variant), os.path.basename(lib_path(variant))
        lib_want = _digest(objs[variant])
        if force or any(j[2] == variant for j in jobs) or not os.path.exists(lib) or stamp.get(lib_key) != lib_want:
            cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs[variant] + ["-ldl"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            stamp[lib_key] = lib_want
            _write_stamp(stamp)
        libs.append(lib)
    return libs


if __name__ == "__main__":
    # no --variant: the default library only; --variant=f16: the fp16-split build; --all: both in one pool
    vs = ("", "f16") if "--all" in sys.argv else tuple(a.split("=", 1)[1] for a in sys.argv if a.startswith("--variant=")) or ("",)
    build_all(force="--force" in sys.argv, variants=vs)
