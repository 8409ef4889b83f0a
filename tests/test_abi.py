"""CPU: the C-ABI library builds, loads, and exports exactly the symbols include/cambrian_amd.h declares;
host-side guards (no CPU fallback) hold.  No kernel is launched here."""
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "cambrian_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cmb_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from cambrian_amd import lib
    return lib


def test_every_declared_symbol_is_exported_and_bound(built):
    syms = _header_symbols()
    assert len(syms) >= 25
    lib = built.load()
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in cambrian_amd.h but not exported"
        assert s in built.SIGNATURES, f"{s} has no ctypes signature in cambrian_amd/lib.py"
    extra = set(built.SIGNATURES) - set(syms)
    assert not extra, f"bound but undeclared symbols: {extra}"
    assert lib.cmb_version().decode().endswith("gfx950")


def test_library_contains_gfx950_code_object(built):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", built.LIB_PATH],
                         capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_no_cpu_fallback(built):
    from cambrian_amd import ops
    with pytest.raises(built.CambrianAmdError):
        ops.k_gemm(torch.zeros(4, 64), torch.zeros(8, 64))
    from cambrian_amd.model.vision_sampler import VisionTokenSampler
    m = VisionTokenSampler(1024, 1024, [1024], [1], 1024, 1)
    q = torch.zeros(2, 1, 1024)
    with pytest.raises(built.CambrianAmdError):
        m(q, q, q, torch.ones(2, 1, dtype=torch.bool))


def test_missing_library_fails_loudly(built, monkeypatch):
    monkeypatch.setattr(built, "_lib", None)
    monkeypatch.setattr(built, "LIB_PATH", "/nonexistent/libcambrian_amd.so")
    with pytest.raises(built.CambrianAmdError, match="no PyTorch fallback"):
        built.load()


def test_state_dict_keys_match_reference_contract():
    """SURVEY.md §8b key list (vision_sampler.py:170-175,254-267)."""
    from cambrian_amd.model.vision_sampler import VisionTokenSampler
    m = VisionTokenSampler(4096, 1024, [1024] * 4, [1, 1, 1, 4], 1024, 1)
    keys = set(m.state_dict().keys())
    want = {"layers.0.proj_context.weight", "layers.0.proj_in.weight", "layers.0.proj_out.linear_1.weight",
            "layers.0.proj_out.linear_2.weight", "layers.0.norm.weight", "layers.0.norm.bias",
            "layers.0.cross_attn.q_proj.0.weight", "layers.0.cross_attn.q_proj.0.bias",
            "layers.0.cross_attn.q_proj.1.weight", "layers.0.cross_attn.o_proj.weight", "layers.0.pos_embed_3"}
    for i in range(4):
        for kv in "kv":
            want |= {f"layers.0.cross_attn.{kv}_proj_{i}.0.weight", f"layers.0.cross_attn.{kv}_proj_{i}.0.bias",
                     f"layers.0.cross_attn.{kv}_proj_{i}.1.weight"}
    assert keys == want
    sd = m.state_dict()
    assert sd["layers.0.proj_in.weight"].shape == (1024, 4096 + 1024)
    assert sd["layers.0.proj_out.linear_2.weight"].shape == (4096, 1024)
    assert sd["layers.0.pos_embed_3"].shape == (16, 1024)


def test_abi_revision_is_checked(built, monkeypatch):
    """a library of another ABI revision resolves every symbol and then receives shifted argument lists; the
    binding compares cmb_abi_version() with the header revision it was written against."""
    text = open(os.path.join(ROOT, "include", "cambrian_amd.h")).read()
    rev = int(re.search(r"#define\s+CMB_ABI_VERSION\s+(\d+)", text).group(1))
    assert built.ABI_VERSION == rev == built.load().cmb_abi_version()
    monkeypatch.setattr(built, "_lib", None)
    monkeypatch.setattr(built, "ABI_VERSION", rev + 1)
    with pytest.raises(built.CambrianAmdError, match="ABI revision"):
        built.load()


def test_only_the_c_abi_leaves_the_library(built):
    """`nm -D`: every defined dynamic symbol is a ``cmb_*`` entry point of the header — no mangled C++ launcher, no per-TU
    marker (csrc/exports.map)."""
    out = subprocess.run(["nm", "-D", "--defined-only", built.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert names, "no dynamic symbols?"
    foreign = [n for n in names if not n.startswith("cmb_")]
    assert not foreign, f"symbols outside the C-ABI are exported: {foreign[:5]}"
    assert sorted(set(names)) == _header_symbols()


def test_knob_values_are_validated(built):
    """cmb_knob_set rejects values outside each knob's closed set: a typo in an A/B run must not launch a bogus
    grid or silently select another variant."""
    lib = built.load()
    ok, bad = 0, -1
    saved = [lib.cmb_knob_get(k) for k in range(8)]
    try:
        assert lib.cmb_knob_set(built.KNOB_DWCONV, -3) == bad
        assert lib.cmb_knob_set(built.KNOB_LN_FWD, -1) == bad
        assert lib.cmb_knob_set(built.KNOB_VIT_ATTN, 4) == bad
        assert lib.cmb_knob_set(built.KNOB_SVA_ABS, 7) == bad
        assert lib.cmb_knob_set(built.KNOB_LN_MULTI_CHUNK, 3) == bad and lib.cmb_knob_set(built.KNOB_LN_MULTI_CHUNK, 8) == bad
        assert lib.cmb_knob_set(99, 0) == bad
        assert lib.cmb_knob_set(built.KNOB_COLSUM_WGS, -1) == bad
        assert lib.cmb_knob_set(built.KNOB_LN_BWD_ROWS, 0) == bad and lib.cmb_knob_set(built.KNOB_LN_BWD_ROWS, 257) == bad
        assert [lib.cmb_knob_get(k) for k in range(8)] == saved          # a rejected value changes nothing
        assert lib.cmb_knob_set(built.KNOB_COLSUM_WGS, 0) == ok and lib.cmb_knob_set(built.KNOB_LN_BWD_ROWS, 64) == ok
        assert lib.cmb_knob_set(built.KNOB_LN_MULTI_CHUNK, 7) == ok and lib.cmb_knob_get(built.KNOB_LN_MULTI_CHUNK) == 7
        assert lib.cmb_knob_set(built.KNOB_DWCONV, 32) == ok
    finally:
        for k, v in enumerate(saved):
            assert lib.cmb_knob_set(k, v) == ok
