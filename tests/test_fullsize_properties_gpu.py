"""`-m gpu`: parity at BASELINE.json's FULL sizes through size-independent properties (the CPU oracle only finishes
small cases in seconds; tests/test_sva_gpu.py, test_towers_gpu.py and test_model_gpu.py compare those against it).

Release-8B shapes (scripts/cambrian/pretrain_cambrian_8b.sh:15-27): 576 latent queries per image on a 24 x 24 grid,
towers with 576 / 576 / 576 / 9216 tokens (windows of 1, 1, 1 and 4 x 4 keys), hidden 1024, q_dim 1024 | 4096,
2048-token sequences of width 4096, 128256-word vocabulary.  Properties:
  * batch independence (bit-exact): an image's visual tokens do not depend on what else is in the batch;
  * masked keys are dead (bit-exact): perturbing a masked key's features changes nothing (vision_sampler.py:200-220);
  * uniform scores -> the attention output is the mean of the unmasked values (softmax of equal logits);
  * window gather: a query whose mask leaves ONE key open returns exactly that key's value row
    (the index arithmetic that replaces cambrian_arch.py:271-287's view/permute/contiguous);
  * embedding splice (bit-exact) against plain torch indexing of cambrian_arch.py:457-490 at [8, 2048, 4096];
  * cross-entropy: loss of uniform logits = log(V); gradient rows sum to zero; ignored rows have zero gradient."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

KV_SIZES = [1, 1, 1, 4]
QSIDE, HID = 24, 1024


def _sampler(dev, q_dim, layers=1, seed=0):
    from cambrian_amd.model.vision_sampler import VisionTokenSampler
    torch.manual_seed(seed)
    m = VisionTokenSampler(q_dim, HID, [HID] * 4, KV_SIZES, HID, layers)
    return m.to(dev)


def _inputs(B, q_dim, dev, dt, seed=1, mask_p=0.3):
    g = torch.Generator().manual_seed(seed)
    Bq = B * QSIDE * QSIDE
    q = torch.randn(Bq, q_dim, generator=g).to(dev, dt)
    ctx = torch.randn(B, HID, generator=g).to(dev, dt)
    feats = [torch.randn(B * (QSIDE * s) ** 2, HID, generator=g).to(dev, dt) for s in KV_SIZES]
    masks = []
    for s in KV_SIZES:
        m = torch.rand(Bq, s * s, generator=g) > mask_p
        m[m.sum(1) == 0] = True                       # the collator never leaves a window fully masked (train_fsdp.py:1133-1137)
        masks.append(m)
    return q, ctx, feats, masks


def _run(m, q, ctx, feats, masks, B):
    from cambrian_amd import ops
    holders = [ops.GradAccumulator() for _ in KV_SIZES]
    mu8 = [x.to(torch.uint8).to(q.device) for x in masks]
    with torch.no_grad():
        return m.forward_fused(q, ctx, feats, mu8, holders, B, QSIDE)


@pytest.mark.parametrize("q_dim", [1024, 4096])
def test_sva_batch_independence_fullsize(dev, q_dim):
    B, dt = 4, torch.bfloat16
    m = _sampler(dev, q_dim)
    q, ctx, feats, masks = _inputs(B, q_dim, dev, dt)
    full = _run(m, q, ctx, feats, masks, B)
    assert full.shape == (B * 576, q_dim) and torch.isfinite(full.float()).all()
    n = 576
    for b in (0, B - 1):
        one = _run(m, q[b * n:(b + 1) * n], ctx[b:b + 1],
                   [f[b * (QSIDE * s) ** 2:(b + 1) * (QSIDE * s) ** 2] for f, s in zip(feats, KV_SIZES)],
                   [mk[b * n:(b + 1) * n] for mk in masks], 1)
        assert torch.equal(one, full[b * n:(b + 1) * n]), "an image's tokens changed with the batch composition"


def test_sva_masked_keys_are_dead_fullsize(dev):
    B, dt, q_dim = 2, torch.bfloat16, 1024
    m = _sampler(dev, q_dim, layers=2)
    q, ctx, feats, masks = _inputs(B, q_dim, dev, dt)
    base = _run(m, q, ctx, feats, masks, B)
    # tower 3 (4 x 4 windows): token (y, x) of image b belongs to query (y // 4, x // 4), window slot (y % 4) * 4 + x % 4
    G = QSIDE * 4
    f3 = feats[3].clone().view(B, G, G, HID)
    mk = masks[3].view(B, QSIDE, QSIDE, 4, 4).permute(0, 1, 3, 2, 4).reshape(B, G, G).to(f3.device)  # token-major mask
    noise = torch.randn_like(f3) * 50
    f3 = torch.where(mk[..., None], f3, f3 + noise)
    assert (~mk).any()
    pert = _run(m, q, ctx, [feats[0], feats[1], feats[2], f3.view(-1, HID)], masks, B)
    assert torch.equal(base, pert), "a masked key influenced the output"


def test_sva_attention_uniform_scores_and_single_key_fullsize(dev):
    """Kernel-level (cmb_sva_attn_fwd) at the release grid: q = 0 -> uniform softmax over the open keys -> mean of their
    V rows; one open key -> that key's V row exactly (bf16 round trip of an fp32 copy)."""
    from cambrian_amd import ops
    B, heads, hd, dt = 2, 16, 64, torch.bfloat16
    g = torch.Generator().manual_seed(5)
    Bq = B * 576
    kvs = [torch.randn(B * (QSIDE * s) ** 2, 2 * heads * hd, generator=g).to(dev, dt) for s in KV_SIZES]
    qz = torch.zeros(Bq, heads * hd, device=dev, dtype=dt)
    masks = [(torch.rand(Bq, s * s, generator=g) > 0.4) for s in KV_SIZES]
    masks[0][:] = True
    out, _ = ops.k_sva_attn_fwd(qz, kvs, [x.to(torch.uint8).to(dev) for x in masks], KV_SIZES, B, QSIDE, heads, hd)
    # reference mean with the window gather written as the reference does (cambrian_arch.py:280-283)
    vsum = torch.zeros(Bq, heads * hd, device=dev)
    cnt = torch.zeros(Bq, 1, device=dev)
    for kv, s, mk in zip(kvs, KV_SIZES, masks):
        v = kv[:, heads * hd:].float().view(B, QSIDE, s, QSIDE, s, heads * hd).permute(0, 1, 3, 2, 4, 5).reshape(Bq, s * s, -1)
        mkd = mk.to(dev).float()[..., None]
        vsum += (v * mkd).sum(1)
        cnt += mkd.sum(1)
    ref = vsum / cnt
    assert ((out.float() - ref).abs().max() / ref.abs().max()).item() < 1e-2
    # one open key per query, in tower 3 only
    slot = torch.randint(0, 16, (Bq,), generator=g)
    only = [torch.zeros(Bq, s * s, dtype=torch.uint8) for s in KV_SIZES]
    only[3][torch.arange(Bq), slot] = 1
    qr = torch.randn(Bq, heads * hd, generator=g).to(dev, dt)
    out1, _ = ops.k_sva_attn_fwd(qr, kvs, [x.to(dev) for x in only], KV_SIZES, B, QSIDE, heads, hd)
    v3 = kvs[3][:, heads * hd:].view(B, QSIDE, 4, QSIDE, 4, heads * hd).permute(0, 1, 3, 2, 4, 5).reshape(Bq, 16, -1)
    want = v3[torch.arange(Bq, device=dev), slot.to(dev)]
    assert torch.equal(out1, want), "window gather returned the wrong key"


def test_embed_splice_bit_exact_fullsize(dev):
    from cambrian_amd import ops
    B, S, H, V, side, p0 = 8, 2048, 4096, 128256, 24, 91
    g = torch.Generator().manual_seed(9)
    table = torch.randn(V, H, generator=g).to(torch.bfloat16).to(dev)
    ids = torch.randint(1000, 30000, (B, S), generator=g)
    ids[:, p0] = -200
    ids[:, p0 + 1:p0 + 600] = 0
    ids[3, :] = torch.randint(1000, 30000, (S,), generator=g)       # a pure-text row (no image token)
    feat = torch.randn(B, side * side, H, generator=g).to(torch.bfloat16).to(dev)
    newline = torch.randn(H, generator=g).to(torch.bfloat16).to(dev)
    out, pos = ops.embed_splice(ids.to(dev), table, feat, newline, side, -200)
    # cambrian_arch.py:413-420 + :457-490 with torch ops
    vis = torch.cat([feat.view(B, side, side, H), newline.view(1, 1, 1, H).expand(B, side, 1, H)], dim=2).reshape(B, 600, H)
    emb = table[torch.where(ids == -200, 0, ids).to(dev)]
    for b in range(B):
        if b != 3:
            emb[b, p0:p0 + 600] = vis[b]
    assert torch.equal(out, emb)
    assert pos.tolist() == [p0, p0, p0, -1, p0, p0, p0, p0]


def test_cross_entropy_properties_fullsize(dev):
    from cambrian_amd import ops
    T, V = 4096, 128256
    g = torch.Generator().manual_seed(3)
    labels = torch.randint(0, V, (T,), generator=g)
    labels[::5] = -100
    x = torch.zeros(T, V, device=dev, dtype=torch.bfloat16)
    loss = ops.cross_entropy(x, labels.to(dev), -100)
    assert abs(loss.item() - math.log(V)) < 1e-4                      # uniform logits
    x = (torch.randn(T, V, generator=g) * 2).to(torch.bfloat16).to(dev).requires_grad_()
    y = x * 1.0
    loss = ops.cross_entropy(y, labels.to(dev), -100, inplace=True)
    ref_rows = [0, 1, 7, T - 1]
    ref = torch.nn.functional.cross_entropy(x.detach()[ref_rows].float(), labels[ref_rows].to(dev), ignore_index=-100,
                                            reduction="none")
    loss.backward()
    gsum = x.grad.float().sum(1)
    assert gsum.abs().max().item() < 2e-3 / (labels != -100).sum().item() * 50   # softmax - onehot sums to 0 per row
    assert torch.count_nonzero(x.grad[::5]) == 0                               # ignored rows
    lse = torch.logsumexp(x.detach()[ref_rows].float(), dim=1)
    want = torch.where(labels[ref_rows].to(dev) == -100, torch.zeros_like(lse),
                       lse - x.detach()[ref_rows].float().gather(1, labels[ref_rows].clamp_min(0).to(dev)[:, None])[:, 0])
    assert torch.allclose(ref, want, atol=1e-4)


def test_towers_batch_independence_fullsize(dev):
    """CLIP-L/14@336 and ConvNeXt-XXL@1024 at full depth: image 0 alone == image 0 inside a batch of 2 (bit-exact)."""
    from types import SimpleNamespace
    from cambrian_amd.model.multimodal_encoder.builder import build_vision_tower_aux_list
    cfg = SimpleNamespace(mm_vision_tower_aux_list=["openai/clip-vit-large-patch14-336", "clip-convnext-XXL-multi-stage"],
                          mm_vision_tower_aux_token_len_list=[576, 9216], mm_vision_select_layer=-2,
                          mm_vision_select_feature="patch", unfreeze_mm_vision_tower=False)
    for t, res in zip(build_vision_tower_aux_list(cfg), (336, 1024)):
        g = torch.Generator().manual_seed(res)
        x = torch.randn(2, 3, res, res, generator=g).to(dev, torch.bfloat16)
        both, one = t(x), t(x[:1])
        assert torch.isfinite(both.float()).all()
        assert torch.equal(both[:1], one)


def test_convnext_xxl_batch_24_has_no_index_overflow(dev):
    """bench.py's default batch of 24 images makes ConvNeXt-XXL's stage-1 MLP intermediate 24 x 65536 x 1536 = 2.4e9
    elements (> 2^31) and the aux-projector input 221184 x 5760: the same 24 images as two batches of 12 must give the same
    features (row ranges of a GEMM may change kernels with the grid — tail split — so a few bf16 ulps, not bit equality),
    and the last image — the highest addresses — is checked on its own."""
    from types import SimpleNamespace
    from conftest import rel_err
    from cambrian_amd.model.multimodal_encoder.builder import build_vision_tower_aux_list
    cfg = SimpleNamespace(mm_vision_tower_aux_list=["clip-convnext-XXL-multi-stage"], mm_vision_tower_aux_token_len_list=[9216],
                          mm_vision_select_layer=-2, mm_vision_select_feature="patch", unfreeze_mm_vision_tower=False)
    (t,) = build_vision_tower_aux_list(cfg)
    x = torch.randn(24, 3, 1024, 1024, generator=torch.Generator().manual_seed(24)).to(dev, torch.bfloat16)
    whole = t(x)
    assert whole.shape == (24, 9216, 5760) and torch.isfinite(whole.float()).all()
    halves = torch.cat([t(x[:12]), t(x[12:])], 0)
    assert rel_err(whole, halves.float()) < 2e-2
    assert rel_err(whole[23], halves[23].float()) < 2e-2 and rel_err(whole[23], t(x[23:24])[0].float()) < 2e-2
