"""Tune (PyTorch TunableOp) the hipBLASLt / rocBLAS solution of the frozen Llama-3-8B projection GEMMs at the bench
shape (M = 16 images x 2048 tokens) and append to tools/llm_gemm_tuning/llama3_8b_b16_gfx950.csv (load_tuned.py reads it; a round-1 experiment)
with tuning DISABLED (so the driver's run pays nothing).  Usage: python tools/llm_gemm_tuning/tune_llm_gemms.py [M] (GPU box)."""
import os
import sys
import time

import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "llama3_8b_b16_gfx950.csv")
t = torch.cuda.tunable
t.enable(True)
t.tuning_enable(True)
t.set_max_tuning_duration(8)
t.set_max_tuning_iterations(4)
t.set_filename(path, insert_device_ordinal=False)
if os.path.exists(path):
    t.read_file(path)
dev = torch.device("cuda:0")
H, I, QKV, V = 4096, 14336, 6144, 128256
# (rows of A, K of A) x (weight [N_out, N_in]) forward = A @ W^T ("TN"), backward dX = G @ W ("NN")
fwd = [(H, QKV), (H, H), (H, 2 * I), (I, H)]
for k_in, n_out in fwd:
    a = torch.randn(M, k_in, device=dev, dtype=torch.bfloat16)
    w = torch.randn(n_out, k_in, device=dev, dtype=torch.bfloat16)
    t0 = time.time()
    torch.nn.functional.linear(a, w)
    g = torch.randn(M, n_out, device=dev, dtype=torch.bfloat16)
    torch.mm(g, w)
    torch.cuda.synchronize()
    print(f"tuned fwd/dX for K={k_in} N={n_out} in {time.time() - t0:.1f} s", flush=True)
    del a, w, g
t.write_file(path)
print(open(path).read())
