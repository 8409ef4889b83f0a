"""How much does one more small launch cost a stream?  N dependent tiny launches between two events, eager and as a graph replay;
and the same tiny launches interleaved with a 40-us GEMM (does the small launch's cost hide behind anything?)."""
import json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cambrian_amd import lib as L, ops
dev = torch.device("cuda", 0)
x = torch.zeros(1024, device=dev)
a = torch.randn(13824, 1024, device=dev).bfloat16(); w = torch.randn(1024, 1024, device=dev).bfloat16()
big = torch.randn(17520, 1536, device=dev).bfloat16()
g = torch.ones(1536, device=dev); b = torch.zeros(1536, device=dev)

def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best

N = 400
out = {}
out["fill_x%d_us_each" % N] = timed(lambda: [x.zero_() for _ in range(N)]) / N
out["gemm_x%d_us_each" % N] = timed(lambda: [ops.k_gemm(a, w) for _ in range(N)]) / N
out["gemm+fill_x%d_us_each_pair" % N] = timed(lambda: [(ops.k_gemm(a, w), x.zero_()) for _ in range(N)]) / N
out["gemm+3fill_x%d_us_each_group" % N] = timed(lambda: [(ops.k_gemm(a, w), x.zero_(), x.zero_(), x.zero_()) for _ in range(N)]) / N
out["ln_x%d_us_each" % N] = timed(lambda: [ops.k_layernorm_fwd(big, g, b, 1e-5, want_stats=False) for _ in range(N)]) / N
out["ln+fill_x%d_us_each_pair" % N] = timed(lambda: [(ops.k_layernorm_fwd(big, g, b, 1e-5, want_stats=False), x.zero_()) for _ in range(N)]) / N
# graph replay of the fills
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): x.zero_()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        for _ in range(N): x.zero_()
torch.cuda.synchronize()
out["graph_fill_x%d_us_each" % N] = timed(lambda: gr.replay()) / N
print(json.dumps(out))
