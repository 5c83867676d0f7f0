#!/usr/bin/env python
"""Timing experiments on throw-away variants of a kernel source (results are WRONG by construction; only the
timing matters).  Builds build/exp/libpfhip_<name>.so from a patched copy of one .hip file + the in-tree objects.
usage: python tools/exp_variant.py <file.hip> <name> '<old>' '<new>' ['<old2>' '<new2>' ...]"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "polyffusion_amd", "csrc")
f, name, subs = sys.argv[1], sys.argv[2], sys.argv[3:]
src = open(os.path.join(CSRC, f)).read()
head = ""
if os.environ.get("PF_EXP_AFTER"):   # patch only what follows the first occurrence of this marker (e.g. one kernel of several alike)
    i = src.index(os.environ["PF_EXP_AFTER"])
    head, src = src[:i], src[i:]
for old, new in zip(subs[0::2], subs[1::2]):
    assert src.count(old) >= 1, f"pattern not found: {old[:60]}"
    src = src.replace(old, new)
src = head + src
out_dir = os.path.join(REPO, "build", "exp")
os.makedirs(out_dir, exist_ok=True)
patched = os.path.join(CSRC, f"_exp_{name}_{f}")   # must sit next to its headers
open(patched, "w").write(src)
obj = os.path.join(out_dir, f"{name}.o")
try:
    sys.path.insert(0, REPO)
    from polyffusion_amd.build import EXTRA_FLAGS
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function"] + EXTRA_FLAGS.get(f, []) + os.environ.get("PF_EXP_DEFS", "").split() + ["-c", patched, "-o", obj])
finally:
    os.remove(patched)
objs = [os.path.join(CSRC, o) for o in sorted(os.listdir(CSRC)) if o.endswith(".o") and o != f.replace(".hip", ".o")]
so = os.path.join(out_dir, f"libpfhip_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + objs)
print(so)
