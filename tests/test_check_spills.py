"""The build-time ISA guard of cambrian_amd/csrc (check_spills.py) on synthetic assembly: it must flag a VALU write of the
LDS-DMA's SGPR base (v_readlane / v_readfirstlane) or a label closer than 5 wait states to a piece, count s_nop N as N + 1
wait states, ignore writes to other registers — and pass the real kernel's assembly when a build is present."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_spills", os.path.join(ROOT, "cambrian_amd", "csrc", "check_spills.py"))
cs = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cs)

HEAD = "_ZN3foo17gemm_nt_p5_kernelILi0ELi0EEEv:\n"
TAIL = ".Lfunc_end0:\n"
PIECE = "\ts_add_u32 m0, s17, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v9, s[6:7]\n"


def run(tmp_path, body):
    f = tmp_path / "k.s"
    f.write_text(HEAD + body + TAIL)
    return cs.main(str(f))


def test_clean_piece_passes(tmp_path, capsys):
    body = "\tv_readlane_b32 s6, v231, 2\n" + "\tv_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n" * 3 + PIECE
    assert run(tmp_path, body) == 0          # 3 MFMA + s_add + s_nop 0 = 5 wait states


def test_reload_too_close_is_flagged(tmp_path, capsys):
    body = "\tv_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n" * 4 + "\tv_readlane_b32 s7, v231, 3\n" + PIECE
    assert run(tmp_path, body) == 1
    assert "v_readlane_b32 s7" in capsys.readouterr().out


def test_s_nop_counts_its_wait_states(tmp_path):
    assert run(tmp_path, "\tv_readfirstlane_b32 s6, v3\n\ts_nop 2\n" + PIECE) == 0    # 3 + 1 + 1 = 5
    assert run(tmp_path, "\tv_readfirstlane_b32 s6, v3\n\ts_nop 1\n" + PIECE) == 1    # 2 + 1 + 1 = 4


def test_other_registers_do_not_matter(tmp_path):
    assert run(tmp_path, "\tv_mov_b32 v1, v2\n" * 6 + "\tv_readlane_b32 s9, v231, 3\n" + PIECE) == 0


def test_label_in_front_of_a_piece_is_flagged(tmp_path):
    assert run(tmp_path, "\tv_mov_b32 v1, v2\n" * 6 + ".LBB5_3:\n" + PIECE) == 1


def test_real_build_passes_if_present():
    s = os.path.join(ROOT, "cambrian_amd", "csrc", "build", "gemm_p5-hip-amdgcn-amd-amdhsa-gfx950.s")
    if os.path.exists(s):
        assert cs.main(s) == 0
