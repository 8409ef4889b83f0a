import torch, time
import torch.nn.functional as F
print(torch.__version__)
try:
    print("preferred fa lib:", torch.backends.cuda.preferred_rocm_fa_library())
except Exception as e:
    print("no preferred_rocm_fa_library:", e)
dev = torch.device("cuda:0")
B, H, Hkv, S, D = 8, 32, 8, 2048, 128
q = torch.randn(B, H, S, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
k = torch.randn(B, Hkv, S, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
v = torch.randn(B, Hkv, S, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
def run(tag):
    for it in range(3):
        o = F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True)
        o.sum().backward()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for it in range(5):
        o = F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True)
    e[1].record()
    g = torch.ones_like(o)
    for it in range(5):
        o.backward(g, retain_graph=True)
    e[2].record()
    torch.cuda.synchronize()
    print(tag, "fwd ms", e[0].elapsed_time(e[1]) / 5, "bwd ms", e[1].elapsed_time(e[2]) / 5)
run("default")
for lib in ("ck", "aotriton"):
    try:
        torch.backends.cuda.preferred_rocm_fa_library(lib)
        print("set", lib, "->", torch.backends.cuda.preferred_rocm_fa_library())
        run(lib)
    except Exception as ex:
        print("cannot use", lib, ":", repr(ex)[:300])
# expanded-KV (no GQA flag) variant
k2 = k.detach().repeat_interleave(4, dim=1).requires_grad_()
v2 = v.detach().repeat_interleave(4, dim=1).requires_grad_()
for it in range(2):
    o = F.scaled_dot_product_attention(q, k2, v2, is_causal=True); o.sum().backward()
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
e[0].record()
for it in range(5):
    o = F.scaled_dot_product_attention(q, k2, v2, is_causal=True)
e[1].record()
g = torch.ones_like(o)
for it in range(5):
    o.backward(g, retain_graph=True)
e[2].record(); torch.cuda.synchronize()
print("expanded kv: fwd ms", e[0].elapsed_time(e[1]) / 5, "bwd ms", e[1].elapsed_time(e[2]) / 5)
