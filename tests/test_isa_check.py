"""The generated gfx950 code of the tile-store GEMM kernels carries no in-place packed-f32 instruction with a half selection -- the form that made the
fused RoPE epilogue differ run to run (tools/check_isa.py).  CPU: disassembles the objects the build left."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_isa  # noqa: E402

_ASM = """
0000000000001000 <void gemm_prefill_kernel<7, false, true, true>(GemmArgs)>:
	v_pk_mul_f32 v[34:35], v[34:35], v[32:33] op_sel:[0,1] op_sel_hi:[0,0]// 000000001000: D3B15022 00024122
	v_pk_fma_f32 v[36:37], v[30:31], v[32:33], v[34:35] op_sel_hi:[1,0,1] neg_lo:[0,0,1]// 000000001008: D3B04024 2C8A411E
	v_pk_mul_f32 v[2:3], v[18:19], v[2:3] op_sel_hi:[0,1]      // 000000001010: D3B14002 10020512
	v_pk_add_f32 v[4:5], v[4:5], v[6:7]                        // 000000001018: D3B24004 18020D04
0000000000002000 <void ln_kernel<5, true, 3>(LnArgs)>:
	v_pk_add_f32 v[66:67], v[66:67], v[66:67] op_sel:[0,1] op_sel_hi:[1,0]// 000000002000: D3B25042 08028542
"""


def test_rule_flags_the_round4_form_and_only_that():
    hits = check_isa.flagged(_ASM)
    assert len(hits) == 1 and "v[34:35], v[34:35]" in hits[0][1] and "gemm_prefill_kernel" in hits[0][0]
    everywhere = check_isa.flagged(_ASM, all_kernels=True)
    assert len(everywhere) == 2 and "ln_kernel" in everywhere[1][0]            # outside the tile-store kernels the form is reported, not failed


def test_built_library_is_clean():
    from indextts_amd import build
    build.build()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_isa.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 flagged" in r.stdout
