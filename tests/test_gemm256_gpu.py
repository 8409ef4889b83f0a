"""GPU parity + race screen of the 256x256 GEMM configurations: the 8-wave / 8-phase kernel (cambrian_amd/csrc/gemm256.hip)
and the persistent 4-wave kernel (cambrian_amd/csrc/gemm_p5.hip; K < 160 falls back to the former inside the library).

Reference = fp32 matmul of the SAME bf16-rounded operands (so the only difference is the accumulation order);
the 128x128 configuration of the same library is the second, independent check (bf16 outputs must agree to one
bf16 ulp of the fp32 result).  Shapes cover every K-tile count parity (1, 2, 3, 4, 5, 16 tiles of 64), ragged M / N
tails that clamp LDS-DMA source rows, split-K slabs, the fused epilogue and the row maps."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _ops():
    from cambrian_amd import ops, lib
    return ops, lib


SHAPES = [(256, 256, 64), (256, 256, 128), (512, 256, 192), (256, 512, 256), (300, 264, 320), (1000, 2048, 1024),
          (4616, 1024, 1024), (257, 8, 64), (33, 520, 128), (2048, 2048, 2048), (8 * 730, 1536, 1536),
          (4352, 6400, 192), (8192, 8448, 320), (65536, 1536, 384)]   # > 256 tiles: several persistent rounds per CU


# tile_hint: 256x256 tile; 2560 / 2561 = 8-wave kernel, schedule 0 (8-phase ping-pong) / 1 (in-wave pipeline, 1 barrier
# per K-tile); 2590 = the persistent 4-wave register-buffered kernel
# (two LDS buffers, the K tile's fragments in registers: the default where it applies)
SCHEDS = [2560, 2561, 2590]


@pytest.mark.parametrize("tile", SCHEDS)
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm256_matches_fp32_reference(dev, M, N, K, tile):
    ops, L = _ops()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = torch.randn(N, K, generator=g).to(torch.bfloat16).to(dev)
    ref = a.float() @ w.float().T                      # fp32 on the GPU (rocBLAS), same rounded operands
    out = ops.k_gemm(a, w, out_dtype=torch.float32, tile=tile)
    assert out.shape == (M, N)
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-5, err                            # fp32 accumulation in a different order only
    out128 = ops.k_gemm(a, w, out_dtype=torch.float32, tile=128)
    assert (out - out128).abs().max().item() / ref.abs().max().item() < 2e-5


@pytest.mark.parametrize("tile", SCHEDS)
def test_gemm256_race_screen(dev, tile):
    """The schedule's LDS-DMA / ds_read ordering is only as good as its barriers: identical launches must be
    bit-identical, across many repetitions and while other work loads the chip."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(3)
    for M, N, K in [(4096, 4096, 4096), (8 * 10944, 2048, 1024), (1024, 1024, 8192), (16 * 4096, 6144, 1536)]:
        a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        w = torch.randn(N, K, generator=g).to(torch.bfloat16).to(dev)
        first = ops.k_gemm(a, w, tile=tile)
        ref = ops.k_gemm(a, w, tile=128)
        assert rel_err(first, ref.float()) < 1e-2
        side = torch.randn(4096, 4096, device=dev)
        s2 = torch.cuda.Stream()
        for it in range(12):
            with torch.cuda.stream(s2):               # uneven background load on a second stream
                side = side @ side * 1e-4
            out = ops.k_gemm(a, w, tile=tile)
            assert torch.equal(out, first), f"non-deterministic result on repetition {it} for {(M, N, K)}"
        torch.cuda.synchronize()


@pytest.mark.parametrize("tile", SCHEDS)
@pytest.mark.parametrize("act", ["none", "gelu_erf", "silu"])
def test_gemm256_epilogue(dev, act, tile):
    ops, L = _ops()
    g = torch.Generator().manual_seed(11)
    M, N, K = 700, 520, 192
    dt = torch.bfloat16
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    bias, cs, res = torch.randn(N, generator=g), torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    a_, w_, res_ = a.to(dt).float(), w.to(dt).float(), res.to(dt).float()
    pre = a_ @ w_.T * 0.5 + bias
    fn = {"none": lambda x: x, "gelu_erf": F.gelu, "silu": F.silu}[act]
    ref = fn(pre) * cs + res_
    pre_out = torch.empty(M, N, dtype=dt, device=dev)
    out = ops.k_gemm(a.to(dev, dt), w.to(dev, dt), bias=bias.to(dev), act=L.ACT_CODES[act], colscale=cs.to(dev),
                     residual=res.to(dev, dt), pre_out=pre_out, alpha=0.5, tile=tile)
    assert rel_err(out, ref) < 1e-2
    assert rel_err(pre_out, pre) < 1e-2


@pytest.mark.parametrize("tile", SCHEDS)
def test_gemm256_splitk_fp32_accumulate(dev, tile):
    """Weight-gradient shape: small output, huge reduction, fp32 out with beta accumulate."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(5)
    N, K, M = 1024, 1024, 8192                         # dW[N,K] = g^T[N,M] @ x^T[K,M]^T
    gt = torch.randn(N, M, generator=g).to(torch.bfloat16).to(dev)
    xt = torch.randn(K, M, generator=g).to(torch.bfloat16).to(dev)
    ref = gt.float() @ xt.float().T
    prev = torch.randn(N, K, generator=g).to(dev)
    out = prev.clone()
    ops.k_gemm(gt, xt, out=out, split_k=8, beta=1.0, tile=tile)
    assert ((out - (ref + prev)).abs().max() / ref.abs().max()).item() < 2e-5


@pytest.mark.parametrize("tile", SCHEDS)
def test_gemm256_rowmaps(dev, tile):
    """In-LLM hook gather/scatter folded into the A / C / residual row maps (cambrian_llama.py:181-207)."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(5)
    B, S, H, side, p0 = 2, 700, 256, 24, 91
    dt = torch.bfloat16
    hidden = torch.randn(B, S, H, generator=g).to(dt)
    w = (torch.randn(H, H, generator=g) * 0.1).to(dt)
    rows = hidden[:, p0:p0 + side * (side + 1)].reshape(B, side, side + 1, H)[:, :, :side].reshape(-1, H).float()
    ref_rows = rows @ w.float().T + rows
    ref = hidden.float().clone()
    ref[:, p0:p0 + side * (side + 1)].view(B, side, side + 1, H)[:, :, :side] = ref_rows.view(B, side, side, H)
    hd, src = hidden.to(dev), hidden.to(dev).clone()
    amap = L.make_map(side * side, side, S * H, (side + 1) * H, H)
    base, sbase = hd.view(-1)[p0 * H:], src.view(-1)[p0 * H:]
    ops.k_gemm(sbase, w.to(dev), M=B * side * side, a_map=amap, residual=base, r_map=amap, out=base, c_map=amap, tile=tile)
    assert rel_err(hd, ref) < 1e-2
    mask = torch.ones(B, S, dtype=torch.bool)
    mask[:, p0:p0 + side * (side + 1)].view(B, side, side + 1)[:, :, :side] = False
    assert torch.equal(hd.cpu()[mask], hidden[mask])   # untouched rows bit-identical


@pytest.mark.parametrize("tile", [2590])
@pytest.mark.parametrize("mode", ["plain", "bias_gelu", "bias_silu", "bias_cs_res", "res", "cs"])
@pytest.mark.parametrize("M", [1024, 1000, 2300])
def test_gemm_p5_register_epilogue(dev, M, mode, tile):
    """The persistent kernel's epilogue leaves from the accumulators (permlane32 swap -> 16-byte stores) with the column
    vectors through the scalar cache and hand-counted residual loads: every combination, full and ragged row tiles
    (ragged rows of a residual launch take the generic path), several tiles per workgroup (N = 768 -> 3 column tiles)."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(M + len(mode))
    N, K = 768, 320
    dt = torch.bfloat16
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    bias, cs, res = torch.randn(N, generator=g), torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    a_, w_, res_ = a.to(dt).float(), w.to(dt).float(), res.to(dt).float()
    ref = a_ @ w_.T
    kw = {}
    if "bias" in mode:
        ref = ref + bias
        kw["bias"] = bias.to(dev)
    if "gelu" in mode:
        ref = F.gelu(ref)
        kw["act"] = L.ACT_CODES["gelu_erf"]
    if "silu" in mode:
        ref = F.silu(ref)
        kw["act"] = L.ACT_CODES["silu"]
    if "cs" in mode:
        ref = ref * cs
        kw["colscale"] = cs.to(dev)
    if "res" in mode:
        ref = ref + res_
        kw["residual"] = res.to(dev, dt)
    out = ops.k_gemm(a.to(dev, dt), w.to(dev, dt), tile=tile, **kw)
    assert rel_err(out, ref) < 1e-2
    out128 = ops.k_gemm(a.to(dev, dt), w.to(dev, dt), tile=128, **kw)
    # same bf16 rounding of the same fp32 values up to the accumulation order: a few bf16 ulps at most
    assert (out.float() - out128.float()).abs().max().item() <= 4e-2 * ref.abs().max().item() * 2 ** -7 * 8


def test_default_dispatch_reports_its_kernel(dev):
    """cmb_gemm_last_kernel: the default picks the 4-wave register-buffered kernel when N is a multiple of 128 (round 4: half
    column tiles), K holds two 64-deep tiles and the cost model prefers 256 x 256 tiles at all; the 8-wave kernel for a ragged half tile or
    K = 64; the 128 tile for small problems; all agree on the result."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(77)
    dt = torch.bfloat16
    outs = {}
    for (M, N, K), want in [((32768, 2048, 512), 2590), ((32768, 1152, 512), 2590), ((32768, 1160, 512), 256), ((32768, 2048, 64), 256),
                             ((8192, 2048, 512), 2590), ((1536, 2048, 512), 128), ((64, 64, 64), 128)]:
        a, w = torch.randn(M, K, generator=g).to(dev, dt), (torch.randn(N, K, generator=g) * 0.2).to(dev, dt)
        out = ops.k_gemm(a, w)
        assert L.load().cmb_gemm_last_kernel() == want, (M, N, K, L.load().cmb_gemm_last_kernel())
        ref = a.float() @ w.float().T
        assert ((out.float() - ref).abs().max() / ref.abs().max()).item() < 1e-2
        outs[(M, N, K)] = out
    a, w = torch.randn(4096, 512, generator=g).to(dev, dt), (torch.randn(1024, 512, generator=g) * 0.2).to(dev, dt)
    assert torch.equal(ops.k_gemm(a, w, tile=2590), ops.k_gemm(a, w, tile=2560))   # same MFMA order, same rounding


@pytest.mark.parametrize("M,N,K,mode", [(11680, 1536, 1536, "plain"), (11664, 4352, 1152, "bias_gelu_res"),
                                         (9232, 4096, 1024, "bias_quick_cs"), (11680, 1536, 4096, "f32out"),
                                         (16896, 4096, 256, "rowmap4096"), (16896, 4096, 256, "rowmap4224")])
def test_tail_split_is_the_same_gemm(dev, M, N, K, mode):
    """gemm.hip "Tail split": a default-dispatch launch whose 256-tile count is a little more than whole rounds runs as a
    256-tile head + a 128-tile tail (two launches, one cmb_gemm call).  Same result as the single-kernel launches, for
    every epilogue, fp32 output, and a row map whose outer period divides the cut."""
    ops, L = _ops()
    lib = L.load()
    g = torch.Generator().manual_seed(M + N + K)
    dt = torch.bfloat16
    a = torch.randn(M, K, generator=g).to(dt).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.1).to(dt).to(dev)
    kw, ref = {}, a.float() @ w.float().T
    if "bias" in mode:
        b = torch.randn(N, generator=g).to(dev)
        kw["bias"], ref = b, ref + b
    if "gelu" in mode:
        kw["act"], ref = L.ACT_GELU_ERF, F.gelu(ref)
    if "quick" in mode:
        kw["act"], ref = L.ACT_QUICK_GELU, ref * torch.sigmoid(1.702 * ref)
    if "cs" in mode:
        cs = torch.randn(N, generator=g).to(dev)
        kw["colscale"], ref = cs, ref * cs
    if "res" in mode:
        r = torch.randn(M, N, generator=g).to(dt).to(dev)
        kw["residual"], ref = r, ref + r.float()
    if mode == "f32out":
        kw["out_dtype"] = torch.float32
    if mode.startswith("rowmap"):
        # A rows gathered from a [nb, T + 7, K] buffer through a row map of outer period T.  The cut (16384 rows) is a
        # multiple of T = 4096 (-> split, the tail's base pointer is the mapped row) but not of 4224 (-> launched whole).
        T = int(mode[6:])
        nb = (M + T - 1) // T
        buf = torch.randn(nb, T + 7, K, generator=g).to(dt).to(dev)
        a = buf[:, :T].reshape(nb * T, K)[:M].contiguous()
        ref = a.float() @ w.float().T
        kw.update(M=M, a_map=L.make_map(T, T, (T + 7) * K, 0, K))   # row r -> (r // T) * (T + 7) * K + (r % T) * K
        out = ops.k_gemm(buf.view(-1), w, **kw)
        whole = ops.k_gemm(a, w, tile=2560)
        assert rel_err(out, ref) < 1e-2 and rel_err(out, whole.float()) < 8e-3
        return
    m1 = lib.cmb_gemm_tail_rows(M, N)
    assert m1 > 0 and m1 % 256 == 0 and m1 < M, (M, N, m1)
    out = ops.k_gemm(a, w, **kw)                          # default dispatch: head + tail
    assert lib.cmb_gemm_last_kernel() in (256, 2590)
    whole = ops.k_gemm(a, w, tile=2560, **kw)             # one 8-wave launch over all rows
    tol = 2e-5 if mode == "f32out" else 1e-2
    assert rel_err(out, ref) < tol and rel_err(whole, ref) < tol
    assert torch.equal(out[:m1], whole[:m1])              # the head rows: same 256-tile arithmetic, bit for bit
    assert rel_err(out[m1:], whole[m1:].float()) < (2e-5 if mode == "f32out" else 8e-3)


@pytest.mark.parametrize("M,N,K,tile,dt,has_bias", [
    (1000, 512, 256, 2590, torch.bfloat16, True), (1000, 512, 256, 2560, torch.bfloat16, True), (1000, 512, 256, 128, torch.bfloat16, True),
    (17520 // 8, 8192, 1536, 0, torch.bfloat16, True), (777, 1024, 128, 2590, torch.bfloat16, False), (300, 96, 64, 128, torch.float32, True),
    (513, 2048, 192, 2590, torch.bfloat16, True)])
def test_gemm_swiglu_pairs_epilogue(dev, M, N, K, tile, dt, has_bias):
    """CMB_ACT_SWIGLU_PAIRS (round 4): interleaved (gate_j, up_j) weight rows, C[m, j] = silu(v[m, 2j]) * v[m, 2j + 1] in the
    GEMM's epilogue, C is N / 2 wide — HF Dinov2SwiGLUFFN's weights_in + silu(x1) * x2 in one launch; every kernel (4-wave,
    8-wave, 128-tile; fp32 on the exact MFMA), ragged row counts, with and without bias; all three kernels agree bit for bit."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dt).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dt).to(dev)
    b = torch.randn(N, generator=g).to(dev) if has_bias else None
    v = a.float() @ w.float().T + (b if has_bias else 0.0)
    ref = torch.nn.functional.silu(v[:, 0::2]) * v[:, 1::2]
    out = ops.k_gemm(a, w, bias=b, act=L.ACT_SWIGLU_PAIRS, tile=tile)
    assert out.shape == (M, N // 2) and out.dtype == dt
    assert rel_err(out, ref) < (2e-5 if dt == torch.float32 else 1e-2)
    if dt == torch.bfloat16 and N % 256 == 0 and K >= 128:
        outs = [ops.k_gemm(a, w, bias=b, act=L.ACT_SWIGLU_PAIRS, tile=t) for t in (2590, 2560, 128)]
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # what it replaces: the plain GEMM followed by the gated product on the packed halves
    if dt == torch.bfloat16:
        wp = torch.cat([w[0::2], w[1::2]], 0).contiguous()
        bp = None if b is None else torch.cat([b[0::2], b[1::2]], 0).contiguous()
        full = ops.k_gemm(a, wp, bias=bp, tile=tile)
        from cambrian_amd.model.multimodal_encoder import vit_ops
        two = vit_ops.k_act_mul(full[:, : N // 2], full[:, N // 2:], L.ACT_SILU)
        assert rel_err(out, two.float()) < 1e-2      # (the unfused form rounds the [M, N] intermediate to bf16)


def test_gemm_swiglu_pairs_rejects_other_epilogues(dev):
    ops, L = _ops()
    a = torch.zeros(256, 128, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(256, 128, dtype=torch.bfloat16, device=dev)
    with pytest.raises(L.CambrianAmdError):
        ops.k_gemm(a, w, act=L.ACT_SWIGLU_PAIRS, colscale=torch.ones(256, device=dev))
    with pytest.raises(L.CambrianAmdError):
        ops.k_gemm(a, w, act=L.ACT_SWIGLU_PAIRS, residual=torch.zeros(256, 128, dtype=torch.bfloat16, device=dev))


@pytest.mark.parametrize("M,N,K,mode", [(1000, 384, 256, "bias_gelu"), (2187, 1152, 512, "bias_cs_res"), (4096, 128, 1536, "plain"),
                                         (700, 1408, 128, "bias"), (33000, 384, 1536, "bias_cs_res")])
def test_p5_half_column_tiles(dev, M, N, K, mode):
    """gemm_nt_p5_kernel on N % 256 == 128 (round 4: SigLIP's 1152-wide projections, ConvNeXt stage 1's 384-wide fc2): the last
    column tile's upper waves skip the epilogue, nothing is written beyond column N (the output buffer is followed by a
    sentinel), every epilogue form; equal to the 8-wave kernel bit for bit where their arithmetic is the same."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    dt = torch.bfloat16
    a = torch.randn(M, K, generator=g).to(dt).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dt).to(dev)
    kw, ref = {}, a.float() @ w.float().T
    if "bias" in mode:
        b = torch.randn(N, generator=g).to(dev)
        kw["bias"], ref = b, ref + b
    if "gelu" in mode:
        kw["act"], ref = L.ACT_GELU_ERF, F.gelu(ref)
    if "cs" in mode:
        cs = torch.randn(N, generator=g).to(dev)
        kw["colscale"], ref = cs, ref * cs
    if "res" in mode:
        r = torch.randn(M, N, generator=g).to(dt).to(dev)
        kw["residual"], ref = r, ref + r.float()
    buf = torch.full((M * N + 4096,), 7.0, dtype=dt, device=dev)
    out = buf[: M * N].view(M, N)
    ops.k_gemm(a, w, out=out, tile=2590, **kw)
    assert L.load().cmb_gemm_last_kernel() == 2590
    assert rel_err(out, ref) < 1e-2
    assert bool((buf[M * N:] == 7.0).all())
    if "cs" not in mode:
        assert torch.equal(out, ops.k_gemm(a, w, tile=2560, **kw))
    # default dispatch takes the 4-wave kernel for such shapes now (when the grid is large enough), possibly as head + tail
    out2 = ops.k_gemm(a, w, **kw)
    assert rel_err(out2, ref) < 1e-2
