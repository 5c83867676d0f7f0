"""Regression guard for the hand-scheduled kernels: no compiler-generated instruction may touch a register that an inline-asm
load (LDS fragment read, hidden halo load) is still in flight to.  See tools/lint_asm.py for the failure this catches: a
register-assignment change turned dead-on-arrival fragment reads of the 3x3 loop's last tap into run-to-run differences.
Compiles the three sources to gfx950 assembly (about a minute; hipcc cross-compiles without a GPU)."""
import os
import shutil
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


@pytest.mark.skipif(shutil.which(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) is None, reason="needs hipcc")
def test_no_instruction_touches_a_register_with_a_hidden_load_in_flight(capsys):
    """Both builds of the split kernels (bf16 pieces; fp16 pieces, -DPF_X3_F16): the register allocation differs, the hazard is the same."""
    from tools import lint_asm
    rc = lint_asm.main(["--variant=", "--variant=f16"])
    out = capsys.readouterr().out
    assert rc == 0 and "lint_asm: clean" in out, out[-3000:]
    # every kernel family was actually analysed
    for name in ("conv_bf3_kernelILi3", "conv_bf3_kernelILi2", "gemm_planes_kernel", "attn_bf3_kernel"):
        assert out.count(name) >= 2            # once per build


def test_lint_flags_the_round2_hazard():
    """The analysis itself: a fragment read still in flight when a compiler move overwrites its register is reported, a
    move after the wait is not."""
    from tools import lint_asm
    body = """
	;;#ASMSTART
	ds_read_b128 v[10:13], v1 offset:0
	;;#ASMEND
	s_cbranch_scc1 .LBB0_2
	v_mov_b32_e32 v11, v40
.LBB0_2:
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_mov_b32_e32 v12, v41
	s_endpgm
""".split("\n")
    bad = lint_asm.lint_function("k", body)
    assert [b[1] for b in bad] == ["v_mov_b32_e32 v11, v40"]
