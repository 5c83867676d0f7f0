#!/usr/bin/env python
"""Lint of the generated gfx950 assembly for the kernels that issue loads hipcc does not know about.

conv_bf16x3.hip, gemm_planes_bf3.hip and attention_bf3.hip read LDS fragments (and, in the 3x3 loop, the next halo) through
inline-asm `ds_read_b128` / `global_load_dwordx4` with hand-counted `s_waitcnt`s.  To the compiler the destination registers
hold their value as soon as the asm statement has executed, so nothing stops it from copying such a register, or from
re-using it for something else when the value is dead, while the load is still in flight - the late return then lands in
whatever lives there.  (Round 2: the fragment reads of the 3x3 loop's last tap are dead values; after an unrelated change the
register allocator placed the loop-exit accumulator copies into their registers, ahead of the final wait, and results differed
from run to run.  The sources now keep those registers allocated until the wait - this lint is the regression guard.)

The check walks the control-flow graph of every kernel (each basic block once per distinct in-flight state) with the two
in-order counters modelled as FIFOs (lgkmcnt: LDS accesses; vmcnt: vector loads/stores incl. direct-to-LDS loads) and reports
every instruction OUTSIDE inline asm that reads or writes a register a hidden load is still in flight to.  Scalar loads
are not modelled: they share lgkmcnt, so a pending one only makes a hand-counted `lgkmcnt(N)` stricter (LDS accesses outstanding
<= all outstanding <= N), never weaker.

usage: python tools/lint_asm.py [--variant=f16] [file.s ...]   (no files: compiles the sources to build/asm[_variant]/ and checks them)
exit status 1 when a violation is found.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["conv_bf16x3.hip", "gemm_planes_bf3.hip", "attention_bf3.hip", "mlp_fused_bf3.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
sys.path.insert(0, REPO)
from polyffusion_amd.build import EXTRA_FLAGS, VARIANTS  # noqa: E402  (the lint must read the assembly the library is built from)

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(text: str) -> set:
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def first_operand_regs(ins: str) -> set:
    ops = ins.split(None, 1)
    return regs(ops[1].split(",")[0]) if len(ops) > 1 else set()


BRANCH = re.compile(r"^(s_branch|s_cbranch_\w+)\s+(\.?\w+)")


def parse_blocks(lines):
    """-> (blocks, label -> block index).  A block is a list of (line number, instruction text, inside inline asm)."""
    blocks, labels, cur, in_asm = [], {}, [], False
    for ln, raw in enumerate(lines):
        ins = raw.strip()
        if ins.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if ins.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not ins or ins.startswith((";", "//")):
            continue
        if ins.endswith(":") or re.match(r"^\.?\w+:\s*(;.*)?$", ins):   # label (possibly followed by a comment)
            if cur:
                blocks.append(cur)
                cur = []
            labels[ins.split(":")[0]] = len(blocks)
            continue
        if ins.startswith("."):
            continue
        cur.append((ln, ins.split(";")[0].strip(), in_asm))
        if BRANCH.match(ins) or ins.startswith("s_endpgm"):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    return blocks, labels


def step(state, ins, in_asm, ln, bad):
    """advance the counter model over one instruction; state = (lgkm tuple, vm tuple) of (frozenset dest, hidden, text)"""
    lgkm, vm = list(state[0]), list(state[1])
    op = ins.split()[0]
    if op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", ins)
        if m:
            del vm[: max(0, len(vm) - int(m.group(1)))]
        m = re.search(r"lgkmcnt\((\d+)\)", ins)
        if m:
            del lgkm[: max(0, len(lgkm) - int(m.group(1)))]
        return tuple(lgkm), tuple(vm)
    if not in_asm:
        touched = regs(ins.split(None, 1)[1]) if " " in ins else set()
        for fifo in (lgkm, vm):
            for dst, hidden, what in fifo:
                if hidden and touched & dst:
                    bad[(ln, ins)] = (sorted(touched & dst), what)
    if op.startswith(("ds_read", "ds_load")):
        lgkm.append((frozenset(first_operand_regs(ins)), in_asm, ins))
    elif op.startswith(("ds_write", "ds_store", "ds_swizzle", "ds_bpermute", "ds_permute")):
        lgkm.append((frozenset(), False, ins))
    elif op.startswith("global_load_lds") or (op.startswith("buffer_load") and " lds" in ins):
        vm.append((frozenset(), False, ins))
    elif op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        vm.append((frozenset(first_operand_regs(ins)), in_asm, ins))
    elif op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic")):
        vm.append((frozenset(), False, ins))
    return tuple(lgkm[-15:]), tuple(vm[-63:])   # the hardware counters saturate at 15 / 63 outstanding


def lint_function(name: str, lines) -> list:
    """Walk the control-flow graph (every block once per distinct in-flight state) -> list of (line, instruction, registers,
    load in flight)."""
    blocks, labels = parse_blocks(lines)
    bad = {}
    seen = set()
    work = [(0, ((), ()))]
    while work:
        b, state = work.pop()
        if b >= len(blocks):
            continue
        sig = (b, tuple((d, h) for d, h, _ in state[0]), tuple((d, h) for d, h, _ in state[1]))
        if sig in seen:
            continue
        seen.add(sig)
        if len(seen) > 200000:
            raise SystemExit(f"lint_asm: state explosion in {name}")
        for ln, ins, in_asm in blocks[b]:
            state = step(state, ins, in_asm, ln, bad)
        last = blocks[b][-1][1] if blocks[b] else ""
        m = BRANCH.match(last)
        if last.startswith("s_endpgm"):
            continue
        if m:
            if m.group(2) in labels:
                work.append((labels[m.group(2)], state))
            if m.group(1) != "s_branch":
                work.append((b + 1, state))
        else:
            work.append((b + 1, state))
    return [(ln, ins, r, what) for (ln, ins), (r, what) in sorted(bad.items())]


def lint_file(path: str) -> int:
    text = open(path).read()
    n_bad = 0
    for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end", text, flags=re.S):
        name, body = m.group(1), m.group(2).split("\n")
        if "ASMSTART" not in m.group(2):
            continue
        bad = lint_function(name, body)
        hidden = sum(1 for l in body if "ds_read_b128" in l or "global_load_dwordx4" in l)
        status = "ok" if not bad else f"{len(bad)} VIOLATION(S)"
        print(f"{os.path.basename(path)}: {name[:100]}  [{hidden} wide loads] {status}")
        for ln, ins, r, what in bad[:8]:
            print(f"    +{ln}: `{ins}` touches v{r} while `{what}` is in flight")
        n_bad += len(bad)
    return n_bad


def compile_to_asm(variants=("",)) -> list:
    """hipcc -S of every linted source for every variant asked for, all processes side by side."""
    procs, outs = [], []
    for variant in variants:
        out_dir = os.path.join(REPO, "build", "asm" + ("_" + variant if variant else ""))
        os.makedirs(out_dir, exist_ok=True)
        for src in SOURCES:
            out = os.path.join(out_dir, src.replace(".hip", ".s"))
            outs.append(out)
            procs.append(subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                                           "-Wno-unused-command-line-argument"] + EXTRA_FLAGS.get(src, []) + VARIANTS[variant] +
                                          [os.path.join(REPO, "polyffusion_amd", "csrc", src), "-o", out]))
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("hipcc failed")
    return outs


def main(argv) -> int:
    variants = [a.split("=", 1)[1] for a in argv if a.startswith("--variant=")]
    argv = [a for a in argv if not a.startswith("--variant=")]
    files = argv or compile_to_asm(variants or [""])
    total = sum(lint_file(f) for f in files)
    print("lint_asm:", "clean" if total == 0 else f"{total} violation(s)")
    return 1 if total else 0


if __name__ == "__main__":
    raise SystemExit(main(sys.argv[1:]))
