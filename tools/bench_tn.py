"""Weight-gradient GEMM per shape: cmb_gemm_tn on g, x as they lie vs transposed copies + the NT kernels (LinearFn's two
paths).  Prints us per weight gradient (transposes included) and TFLOP/s.  python tools/bench_tn.py [rows n_out k_in ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cambrian_amd import ops
from cambrian_amd.ops import pad_to

SHAPES = [(147456, 1024, 3072), (147456, 1024, 1024), (9216, 1024, 1152), (9216, 1024, 1024), (9216, 2048, 1024),
          (9216, 1024, 1536), (9216, 4096, 1024), (9216, 4096, 4096), (9216, 1024, 4096), (16, 1024, 1024), (9232, 1024, 1024)]
if len(sys.argv) > 3:
    a = [int(v) for v in sys.argv[1:]]
    SHAPES = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
dev = torch.device("cuda:0")

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

print("| rows | n_out | k_in | splits nt/tn | tn us | tn TF/s | transposes+nt us | of which gemm us | nt TF/s |\n|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for rows, n, k in SHAPES:
    g = torch.randn(rows, n, device=dev).to(torch.bfloat16)
    x = torch.randn(rows, k, device=dev).to(torch.bfloat16)
    sp = ops._wgrad_splits(n, k, pad_to(rows, 64), 64)       # the NT path's split
    spt = ops._tn_splits(n, k, rows)                            # the TN path's split
    t_tn = timeit(lambda: ops.k_gemm_tn(g, x, split_k=spt))
    mp = pad_to(rows, 64)
    def old():
        gt, xt = ops.k_transpose(g, mp), ops.k_transpose(x, mp)
        return ops.k_gemm(gt, xt, out_dtype=torch.float32, split_k=sp)
    t_old = timeit(old)
    gt, xt = ops.k_transpose(g, mp), ops.k_transpose(x, mp)
    t_g = timeit(lambda: ops.k_gemm(gt, xt, out_dtype=torch.float32, split_k=sp))
    fl = 2.0 * rows * n * k
    print(f"| {rows} | {n} | {k} | {sp}/{spt} | {t_tn:.1f} | {fl / t_tn / 1e6:.0f} | {t_old:.1f} | {t_g:.1f} | {fl / t_g / 1e6:.0f} |", flush=True)
    del g, x, gt, xt
