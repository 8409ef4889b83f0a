import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
M = 32768
def t(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for N, K in [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336), (128256, 4096)]:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    wt = w.t().contiguous()
    g = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    a = t(lambda: F.linear(x, w)); b = t(lambda: torch.mm(x, wt)); c = t(lambda: torch.mm(g, w)); d = t(lambda: torch.mm(g, wt.t()))
    fl = 2.0 * M * N * K
    print(f"N={N} K={K}: fwd NT {a:.3f} ms ({fl/a/1e9:.0f} TF)  fwd NN(w_t) {b:.3f} ms ({fl/b/1e9:.0f} TF) | bwd g@W {c:.3f} ms ({fl/c/1e9:.0f} TF)  bwd g@(w_t).T {d:.3f} ms", flush=True)
    del x, w, wt, g
