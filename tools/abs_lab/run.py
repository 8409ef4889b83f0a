"""Lab timing (NOT product): each tools/abs_lab/build/v<N>.so on the bench shape (16 images x 576 queries, 4 x 4 window, three
one-key towers).  Prints a table of us per launch."""
import ctypes as C, glob, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cambrian_amd import lib as L

NAMES = {0: "baseline", 1: "token mixes not stored", 2: "no token mixes", 3: "score product: 1 of 32 MFMAs", 4: "U / dXb operand: 2 of 32 loads (U only)",
         5: "no window DMA", 6: "bwd: no dX pass", 7: "bwd: dX not stored", 8: "fwd limited to 256 registers (5 workgroups per CU)"}
dev = torch.device("cuda:0")
B, qside, ra = 16, 24, 4
Bq, T = B * qside * qside, ra * ra
g = torch.Generator().manual_seed(0)
def rn(*s): return torch.randn(*s, generator=g).to(torch.bfloat16).to(dev)
q = rn(Bq, 1024); kvs = [rn(Bq, 2048) for _ in range(3)]; xhat = rn(B * (qside * ra) ** 2, 1024)
U = rn(Bq, 16, 1024) * 0.05; bk = torch.randn(1024, generator=g).to(dev); bv = torch.randn(1024, generator=g).to(dev)
out = torch.empty_like(q); xbar = torch.empty_like(U); m3 = torch.empty(Bq, 16, device=dev); P = torch.empty(Bq, 16, 20, device=dev)
dout = rn(Bq, 1024); dxbar = rn(Bq, 16, 1024)
dq = torch.empty_like(q); dkvs = [torch.empty_like(k) for k in kvs]; dU = torch.empty_like(U); dcb = torch.empty(Bq, 16, device=dev)
dxhat = torch.empty_like(xhat)
d = L.SvaAbsDesc()
d.B, d.qside, d.heads, d.hd, d.ntowers, d.window_major, d.ra = B, qside, 16, 64, 3, 0, ra
d.q, d.ldq = q.data_ptr(), 1024
for i, kv in enumerate(kvs):
    d.r[i] = 1; d.kv[i], d.ldkv[i] = kv.data_ptr(), 2048; d.mask[i] = None; d.dkv[i] = dkvs[i].data_ptr()
d.xhat, d.ldx, d.mask_a, d.U, d.bk, d.bv = xhat.data_ptr(), 1024, None, U.data_ptr(), bk.data_ptr(), bv.data_ptr()
d.out, d.ldo, d.xbar, d.m3, d.P = out.data_ptr(), 1024, xbar.data_ptr(), m3.data_ptr(), P.data_ptr()
d.dout, d.lddo, d.dxbar = dout.data_ptr(), 1024, dxbar.data_ptr()
d.dq, d.lddq, d.dU, d.dcb, d.dxhat, d.lddx = dq.data_ptr(), 1024, dU.data_ptr(), dcb.data_ptr(), dxhat.data_ptr(), 1024
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, n=20):
    for _ in range(3): assert fn(C.byref(d), st) == 0
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn(C.byref(d), st)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("| variant | fwd us | bwd us |\n|---|---:|---:|")
for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "build", "v*.so")), key=lambda s: int(os.path.basename(s)[1:-3])):
    n = int(os.path.basename(f)[1:-3])
    so = C.CDLL(f)
    for s in (so.cmb_sva_abs_fwd, so.cmb_sva_abs_bwd):
        s.restype, s.argtypes = C.c_int, [C.POINTER(L.SvaAbsDesc), C.c_void_p]
    print(f"| {n} {NAMES.get(n, '')} | {timeit(so.cmb_sva_abs_fwd):.1f} | {timeit(so.cmb_sva_abs_bwd):.1f} |", flush=True)
