"""cmb_weight_prep / the prepared-weight window of ops.LinearFn (round 6): bf16 copy + transposed bf16 copy of the trainable
weights in one launch per step — bit-identical to the per-use cmb_cast + cmb_transpose pair it replaces, never stale."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from cambrian_amd import lib as L
    from cambrian_amd import ops
    return ops, L


@pytest.fixture()
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


@pytest.mark.parametrize("src_dtype", [torch.float32, torch.bfloat16])
def test_weight_prep_table_equals_cast_and_transpose(dev, src_dtype):
    ops, L = _ops()
    g = torch.Generator().manual_seed(3)
    shapes = [(1024, 1024), (72, 1152), (1000, 64), (8, 8), (4096, 1024), (65, 136)]
    big = torch.randn(300, 2048, generator=g).to(dev, src_dtype)
    srcs = [torch.randn(r, c, generator=g).to(dev, src_dtype) for r, c in shapes] + [big[10:210, 512:1024]]   # + a strided view
    lib = L.load()
    jobs = (L.PrepJob * len(srcs))()
    outs, tile0 = [], 0
    for i, w in enumerate(srcs):
        r, c = w.shape
        rp = ops.pad_to(r, 64)
        wc = torch.full((r, c), 7.0, dtype=torch.bfloat16, device=dev)
        wt = torch.full((c, rp), 7.0, dtype=torch.bfloat16, device=dev)
        j = jobs[i]
        j.src, j.dst, j.dst_t, j.ld_src, j.src_dtype = w.data_ptr(), wc.data_ptr(), wt.data_ptr(), w.stride(0), L.dtype_code(src_dtype)
        j.rows, j.cols, j.rows_pad, j.tile0 = r, c, rp, tile0
        tile0 += lib.cmb_weight_prep_tiles(rp, c)
        outs.append((wc, wt, rp))
    table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
    L.check(lib.cmb_weight_prep(table.data_ptr(), len(srcs), tile0, L.stream_ptr(dev)), "cmb_weight_prep")
    for w, (wc, wt, rp) in zip(srcs, outs):
        want = w.to(torch.bfloat16)
        assert torch.equal(wc, want)
        assert torch.equal(wt[:, :w.shape[0]], want.t())
        assert not wt[:, w.shape[0]:].any()          # zero padding columns
        # one job by value
        wc2, wt2 = torch.empty_like(wc), torch.empty_like(wt)
        j = L.PrepJob()
        j.src, j.dst, j.dst_t, j.ld_src, j.src_dtype = w.data_ptr(), wc2.data_ptr(), wt2.data_ptr(), w.stride(0), L.dtype_code(src_dtype)
        j.rows, j.cols, j.rows_pad = w.shape[0], w.shape[1], rp
        L.check(lib.cmb_weight_prep_one(C.byref(j), L.stream_ptr(dev)), "cmb_weight_prep_one")
        assert torch.equal(wc2, wc) and torch.equal(wt2, wt)
    bad = L.PrepJob()
    assert lib.cmb_weight_prep_one(C.byref(bad), L.stream_ptr(dev)) != 0


def test_linear_inside_a_window_is_the_same_function_and_never_stale(dev):
    ops, L = _ops()
    g = torch.Generator().manual_seed(5)
    lin = torch.nn.Linear(512, 320, bias=True).to(dev)
    wide = torch.nn.Parameter(torch.randn(320, 1024, generator=g).to(dev) * 0.05)     # used through column slices, as proj_in is
    x = torch.randn(384, 512, generator=g).to(dev, torch.bfloat16).requires_grad_()
    x2 = torch.randn(384, 512, generator=g).to(dev, torch.bfloat16)

    def run(window: bool):
        for p in (lin.weight, lin.bias, wide, x):
            p.grad = None
        if window:
            ops.weight_step_begin()
        try:
            y = (ops.linear(x, lin.weight, lin.bias, act=L.ACT_GELU_ERF) + ops.linear(x2, wide[:, 512:])[:, :320]
                 + ops.linear(x2, wide[:, :512])[:, :320])
        finally:
            ops.weight_step_end()
        y.float().square().sum().backward()
        return y.detach().clone(), [t.grad.clone() for t in (lin.weight, lin.bias, wide, x)]

    y0, g0 = run(False)
    n_before = len(ops._PREP)
    y1, g1 = run(True)          # weights seen for the first time: prepared on the spot, registered
    assert len(ops._PREP) == n_before + 3
    y2, g2 = run(True)          # second step: one cmb_weight_prep launch refreshes all three
    def same(ga, gb):
        # weight / input gradients bit for bit; the bias gradient is cmb_colsum's fp32 atomics (order not fixed —
        # with round 6's rows per workgroup 384 rows are 24 partial sums, not 2)
        for i, (a, b) in enumerate(zip(ga, gb)):
            if i == 1:
                assert torch.allclose(a, b, rtol=2e-6, atol=0)
            else:
                assert torch.equal(a, b)

    for y, gs in ((y1, g1), (y2, g2)):
        assert torch.equal(y, y0)
        same(gs, g0)
    with torch.no_grad():       # an optimizer step: the next window serves the NEW weights
        lin.weight.mul_(0.5)
        wide.add_(0.01)
    y3, g3 = run(True)
    y4, g4 = run(False)
    assert torch.equal(y3, y4) and not torch.equal(y3, y0)
    same(g3, g4)
    # outside a window nothing is served from the cache
    assert ops.prepared_weight(lin.weight, torch.bfloat16) is None
