"""CPU: host logic of the start-up GEMM calibration (cambrian_amd/ops.py::gemm_census / calibrate_gemm_dispatch) and of the
library's dispatch tables — no kernel is launched.  The census keeps only problems where BOTH 256 x 256 kernels apply and
the tail split does not; the per-shape policy table accepts / replaces / removes entries and rejects unknown kernels; the
tail-split rule takes the grids it was written for (DINOv2 / SigLIP / CLIP row counts at 16 images) and leaves whole
multiples of a round alone."""
import pytest


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from cambrian_amd import lib as L
    return L.load()


def test_tail_split_rule(lib):
    cases = {(11680, 1536): 10752, (11664, 4352): 11520, (11680, 4608): 10752, (9232, 4096): 8192, (9216, 4096): 8192,
             (65536, 6144): 0, (147456, 2048): 0, (11680, 8192): 0, (9232, 1024): 0, (9216, 1024): 0, (11664, 1152): 0}
    for (M, N), want in cases.items():
        assert lib.cmb_gemm_tail_rows(M, N) == want, (M, N, lib.cmb_gemm_tail_rows(M, N))
    for M in range(256, 40000, 1237):
        for N in (256, 1024, 1536, 4096):
            m1 = lib.cmb_gemm_tail_rows(M, N)
            if not m1:
                continue
            tn, T = N // 256, ((M + 255) // 256) * (N // 256)
            head = (m1 // 256) * tn
            assert m1 % 256 == 0 and 0 < m1 < M
            assert (T // 256) * 256 - tn < head <= (T // 256) * 256          # as many whole rounds as whole row tiles allow
            assert -(-head // 256) + 1.6 * (T - head) / 256 + 0.1 < 0.93 * -(-T // 256)   # the cost model's own inequality


def test_policy_table(lib):
    assert lib.cmb_gemm_policy_clear() == 0
    assert lib.cmb_gemm_policy_set(65536, 6144, 1536, 1, 2590) == 0
    assert lib.cmb_gemm_policy_set(65536, 6144, 1536, 1, 2560) == 0      # replaces
    assert lib.cmb_gemm_policy_set(65536, 6144, 1536, 1, 0) == 0         # removes
    assert lib.cmb_gemm_policy_set(65536, 6144, 1536, 1, 0) == 0         # removing a missing entry is fine
    assert lib.cmb_gemm_policy_set(1, 2, 3, 0, 777) != 0                 # unknown kernel id
    for i in range(64):
        assert lib.cmb_gemm_policy_set(1000 + i, 256, 256, 0, 2590) == 0
    assert lib.cmb_gemm_policy_set(5000, 256, 256, 0, 2590) != 0         # table full
    assert lib.cmb_gemm_policy_clear() == 0


def test_census_top_keeps_only_calibratable_problems(lib):
    from cambrian_amd import ops
    c = ops.gemm_census()
    with c:
        pass
    c.shapes.update({
        (65536, 6144, 1536, 1, False): 30,    # ConvNeXt fc1 + GELU: kept, most FLOPs
        (65536, 1536, 6144, 0, False): 30,    # fc2: kept
        (11680, 1536, 4096, 0, False): 40,    # tail split applies: not calibrated
        (11664, 1152, 4352, 0, False): 27,    # N = 4.5 tile columns: the 4-wave kernel applies since round 4 (half tiles); tail split -> skipped
        (11664, 1160, 4352, 0, False): 27,    # a ragged HALF tile: generic epilogue, not calibrated
        (9216, 1024, 1024, 0, False): 83,     # a single round of tiles
        (147456, 2048, 1024, 0, True): 13,    # pre-activation copy: generic epilogue, 8-wave kernel only
        (147456, 2048, 1024, 0, False): 13,   # kept
        (1048576, 384, 64, 0, False): 1,      # K = 64
    })
    top = c.top(8)
    assert top == [(65536, 6144, 1536, 1), (65536, 1536, 6144, 0), (147456, 2048, 1024, 0)]
    assert c.top(1) == [(65536, 6144, 1536, 1)]


def test_tn_weight_gradient_policy_and_host_checks(lib):
    """ops._tn_wgrad_wins: cmb_gemm_tn takes the query-side weight gradients (tools/bench_tn.py), the large products stay on
    transposes + the NT kernels; cmb_gemm_tn validates its descriptor before any launch (no GPU here: the host checks are
    what runs) and k_gemm_tn refuses CPU tensors like every other operator."""
    import ctypes as C
    import torch
    from cambrian_amd import lib as L
    from cambrian_amd import ops
    assert ops._tn_wgrad_wins(9216, 1024, 1024) and ops._tn_wgrad_wins(9216, 2048, 1024) and ops._tn_wgrad_wins(13824, 1024, 1536)
    assert ops._tn_wgrad_wins(9216, 4096, 1024) and ops._tn_wgrad_wins(9216, 1024, 4096) and ops._tn_wgrad_wins(147456, 1024, 1024)
    assert not ops._tn_wgrad_wins(147456, 1024, 3072) and not ops._tn_wgrad_wins(9216, 4096, 4096)
    # split-K of the TN kernel: one round of its two workgroups per CU (512), never more slices than 128-row pieces
    assert ops._tn_splits(1024, 1024, 9216) == 8 and ops._tn_splits(2048, 1024, 13824) == 4 and ops._tn_splits(4096, 4096, 9216) == 1
    assert ops._tn_splits(64, 1024, 9216, 16) == 4 and ops._tn_splits(1024, 1024, 100) == 1
    with pytest.raises(L.CambrianAmdError):
        ops.k_gemm_tn(torch.zeros(64, 128, dtype=torch.bfloat16), torch.zeros(64, 128, dtype=torch.bfloat16))
    d = L.GemmDesc()
    assert lib.cmb_gemm_tn(C.byref(d), None) != 0                        # null operands: rejected before any launch
