"""cmb_gemm_pair (round 6): two independent bf16 GEMMs as ONE launch of the persistent 256 x 256 kernel with the workgroups
split between the problems (gemm_p5.hip, P5Args) — bit-identical to the two cmb_gemm calls, whatever path the library takes —
and vit.py::forward_paired, which advances two frozen ViT trunks in lock-step through it (DINOv2 beside SigLIP: HF / timm blocks
behind dino_encoder.py:156-165 and siglip_encoder.py:95-99; towers independent, cambrian_arch.py:271-278)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(dev, M, N, K, seed, colscale=False, residual=True, act=0, bias=True):
    g = torch.Generator().manual_seed(seed)
    bf, f32 = torch.bfloat16, torch.float32
    kw = dict(a=torch.randn(M, K, generator=g).to(bf).to(dev), w=(torch.randn(N, K, generator=g) * K ** -0.5).to(bf).to(dev), act=act,
              out=torch.empty(M, N, device=dev, dtype=bf))
    if bias:
        kw["bias"] = torch.randn(N, generator=g).to(dev, f32)
    if colscale:
        kw["colscale"] = torch.randn(N, generator=g).to(dev, f32)
    if residual:
        kw["residual"] = torch.randn(M, N, generator=g).to(bf).to(dev)
    return kw


# (problem 0, problem 1, must the library take the one-launch path?)  Release shapes at 24 and 8 images, ragged M, a half column
# tile (N % 256 == 128), a K of two tiles, and pairs the round arithmetic must refuse (whole rounds already / tiny problems).
CASES = [
    ((24 * 730, 1536, 1536, True), (24 * 729, 1152, 1152, False), True),
    ((24 * 730, 1536, 4096, True), (24 * 729, 1152, 4352, False), True),
    ((8 * 730, 1536, 1536, True), (8 * 729, 1152, 1152, False), None),
    ((70000, 1280, 128, False), (33333, 640, 192, True), None),
    ((65536, 2048, 256, False), (65536, 2048, 256, False), False),     # 2048 tiles each = 8 whole rounds: nothing to gain
    ((300, 264, 320, False), (257, 8, 64, False), False),               # not on the persistent kernel at all
]


@pytest.mark.parametrize("p0,p1,must_pair", CASES)
def test_pair_equals_two_launches(dev, p0, p1, must_pair):
    from cambrian_amd import lib as L
    from cambrian_amd import ops
    k0 = _mk(dev, *p0[:3], seed=11, colscale=p0[3])
    k1 = _mk(dev, *p1[:3], seed=12, colscale=p1[3], residual=p1[1] % 256 == 0)
    ops.k_gemm(**k0)
    ops.k_gemm(**k1)
    r0, r1 = k0["out"].clone(), k1["out"].clone()
    k0["out"].fill_(float("nan"))
    k1["out"].fill_(float("nan"))
    o0, o1 = ops.k_gemm_pair(k0, k1)
    paired = bool(L.load().cmb_gemm_pair_last())
    if must_pair is not None:
        assert paired == must_pair
    assert o0 is k0["out"] and o1 is k1["out"]
    assert torch.equal(o0, r0) and torch.equal(o1, r1)
    # and against fp32 matmul of the same rounded operands (the pair path is not only self-consistent)
    ref = k0["a"].float() @ k0["w"].float().T + k0["bias"]
    if "colscale" in k0:
        ref = ref * k0["colscale"]
    ref = ref + k0["residual"].float()
    assert ((o0.float() - ref).abs().max() / ref.abs().max()).item() < 1e-2


def test_pair_with_activation_and_mismatch(dev):
    """Same activation on both sides pairs (GELU epilogue template); different activations or an fp32 output fall back to two
    launches with the same results."""
    from cambrian_amd import lib as L
    from cambrian_amd import ops
    G = L.ACT_CODES["gelu_erf"]
    k0 = _mk(dev, 24 * 730, 1536, 1536, 5, residual=False, act=G)
    k1 = _mk(dev, 24 * 729, 1152, 1152, 6, residual=False, act=G)
    ops.k_gemm(**k0); ops.k_gemm(**k1)
    r0, r1 = k0["out"].clone(), k1["out"].clone()
    ops.k_gemm_pair(k0, k1)
    assert L.load().cmb_gemm_pair_last() == 1
    assert torch.equal(k0["out"], r0) and torch.equal(k1["out"], r1)
    k1["act"] = 0
    ops.k_gemm(**k1)
    r1 = k1["out"].clone()
    k0["out"].zero_(); k1["out"].zero_()
    ops.k_gemm_pair(k0, k1)
    assert L.load().cmb_gemm_pair_last() == 0
    assert torch.equal(k0["out"], r0) and torch.equal(k1["out"], r1)


def test_pair_race_screen(dev):
    """The same pair 30 times back to back: every result identical (no dependence on which workgroup lands where)."""
    from cambrian_amd import ops
    k0 = _mk(dev, 24 * 730, 1536, 1536, 21, colscale=True)
    k1 = _mk(dev, 24 * 729, 1152, 1152, 22)
    ops.k_gemm_pair(k0, k1)
    r0, r1 = k0["out"].clone(), k1["out"].clone()
    for _ in range(30):
        ops.k_gemm_pair(k0, k1)
        assert torch.equal(k0["out"], r0) and torch.equal(k1["out"], r1)


def test_forward_paired_equals_sequential_trunks(dev):
    """Two frozen trunks of different depth / width / token count in lock-step = the two forwards, bit for bit; the longer trunk
    finishes alone."""
    from cambrian_amd.model.multimodal_encoder.vit import ViTConfig, ViTTrunk, forward_paired
    ca = ViTConfig(image_size=378, patch_size=14, hidden_size=1536, num_layers=3, num_heads=24, mlp_dim=4096, act="swiglu",
                   ln_eps=1e-6, has_cls=True, final_ln=True, layerscale=True)
    cb = ViTConfig(image_size=384, patch_size=14, hidden_size=1152, num_layers=2, num_heads=16, mlp_dim=4304, act="gelu", ln_eps=1e-6,
                   has_cls=False, final_ln=True)
    gen = torch.Generator().manual_seed(77)
    ta = ViTTrunk(ca, torch.bfloat16).load_canonical(ViTTrunk.random_canonical(ca, gen), dev)
    tb = ViTTrunk(cb, torch.bfloat16).load_canonical(ViTTrunk.random_canonical(cb, gen), dev)
    B = 24
    xa = torch.randn(B, 3, 378, 378, generator=gen).to(dev, torch.bfloat16)
    xb = torch.randn(B, 3, 384, 384, generator=gen).to(dev, torch.bfloat16)
    ra, rb = ta(xa), tb(xb)
    pa, pb = forward_paired(ta, xa, tb, xb)
    assert torch.equal(pa, ra) and torch.equal(pb, rb)
    pb2, pa2 = forward_paired(tb, xb, ta, xa)     # the shorter trunk first
    assert torch.equal(pa2, ra) and torch.equal(pb2, rb)


@pytest.mark.parametrize("M,N,K", [(1, 1024, 1024), (24, 1024, 1024), (32, 4096, 1024), (24, 96, 64), (7, 1024, 4096), (24, 1024, 192)])
@pytest.mark.parametrize("odt", [torch.bfloat16, torch.float32])
def test_small_m_kernel(dev, M, N, K, odt):
    """gemm_smallm.hip: M <= 32 rows (the SVA layers' per-image context vectors, vision_sampler.py:279-292) — 32 columns per
    workgroup, K split over its four waves — against fp32 matmul of the same rounded operands and the 128-tile kernel."""
    from cambrian_amd import lib as L
    from cambrian_amd import ops
    g = torch.Generator().manual_seed(M * 131 + N + K)
    a_full = torch.randn(M, K + 64, generator=g).to(torch.bfloat16).to(dev)
    a = a_full[:, :K]                                   # row stride != K
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
    out = ops.k_gemm(a, w, out_dtype=odt)
    assert L.load().cmb_gemm_last_kernel() == 32
    ref = a.float() @ w.float().T
    tol = 2e-5 if odt == torch.float32 else 8e-3
    assert ((out.float() - ref).abs().max() / ref.abs().max()).item() < tol
    out128 = ops.k_gemm(a, w, out_dtype=odt, tile=128)
    assert L.load().cmb_gemm_last_kernel() == 128
    assert ((out.float() - out128.float()).abs().max() / ref.abs().max()).item() < tol
