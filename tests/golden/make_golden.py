"""Generates the golden fixtures in this directory by running the REFERENCE's own Python
(/root/reference, read-only) on seeded inputs.  Run in the build container only:

    python tests/golden/make_golden.py

The GPU box has no /root/reference; tests read the committed ``*.pt`` files.  Nothing here is imported
by the product.  Loader recipe: SURVEY.md §8c (vision_sampler.py imports as-is; cambrian_arch.py needs
stub packages because cambrian/__init__.py drags in timm/open_clip; the collator functions are exec'd
from their line range because train_fsdp.py imports torch_xla unconditionally).
"""
from __future__ import annotations

import importlib.util
import logging
import os
import sys
import types

import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("CAMBRIAN_GOLDEN_OUT") or HERE   # (tests/test_golden_regen.py writes elsewhere)


def load_ref_vision_sampler():
    spec = importlib.util.spec_from_file_location("ref_vision_sampler", f"{REF}/cambrian/model/vision_sampler.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_ref_arch():
    """cambrian_arch.py with stub packages (no timm/open_clip/ezcolorlog needed)."""
    for name, path in [("cambrian", f"{REF}/cambrian"), ("cambrian.model", f"{REF}/cambrian/model"),
                       ("cambrian.model.multimodal_encoder", f"{REF}/cambrian/model/multimodal_encoder"),
                       ("cambrian.model.multimodal_projector", f"{REF}/cambrian/model/multimodal_projector"),
                       ("cambrian.model.language_model", f"{REF}/cambrian/model/language_model")]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
    if "ezcolorlog" not in sys.modules:
        ez = types.ModuleType("ezcolorlog")
        ez.root_logger = logging.getLogger("ref")
        sys.modules["ezcolorlog"] = ez
    if "cambrian.model.multimodal_encoder.builder" not in sys.modules:
        b = types.ModuleType("cambrian.model.multimodal_encoder.builder")
        b.build_vision_tower_aux_list = lambda cfg, **kw: cfg._fake_towers
        sys.modules["cambrian.model.multimodal_encoder.builder"] = b
    import cambrian.model.cambrian_arch as A  # noqa: E402
    A.IS_XLA_AVAILABLE = True  # static (training) path
    return A


def load_ref_collator():
    src = open(f"{REF}/cambrian/train/train_fsdp.py").read().split("\n")
    ns = {"torch": torch, "IMAGE_TOKEN_INDEX": -200, "IGNORE_INDEX": -100}
    exec("\n".join(src[1038:1165]), ns)  # get_padding_offset, prepare_image_info, prepare_multimodal_data
    return ns


# ------------------------------------------------------------------------------------------------
def golden_sva():
    vs = load_ref_vision_sampler()
    torch.manual_seed(1234)
    q_dim, ctx_dim, hidden = 96, 80, 64
    kv_dims, kv_sizes, layers, bq = [64, 64, 64], [1, 1, 2], 2, 8
    m = vs.VisionTokenSampler(q_dim, ctx_dim, kv_dims, kv_sizes, hidden, layers).float()
    with torch.no_grad():  # perturb LayerNorm affines so they are exercised
        for n, p in m.named_parameters():
            if n.endswith(".0.weight") or n.endswith("norm.weight"):
                p.add_(0.1 * torch.randn_like(p))
            if n.endswith(".0.bias") or n.endswith("norm.bias"):
                p.add_(0.1 * torch.randn_like(p))
    q = torch.randn(bq, 1, q_dim, requires_grad=True)
    ctx = torch.randn(bq, 1, ctx_dim, requires_grad=True)
    kvs = [torch.randn(bq, s * s, d, requires_grad=True) for s, d in zip(kv_sizes, kv_dims)]
    masks = [torch.ones(bq, s * s, dtype=torch.bool) for s in kv_sizes]
    masks[2][1, 0] = False
    masks[2][3, 1:] = False
    masks[0][5, 0] = False
    out = m(q, ctx, *kvs, *masks)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    fx = {
        "cfg": dict(q_dim=q_dim, ctx_dim=ctx_dim, hidden=hidden, kv_dims=kv_dims, kv_sizes=kv_sizes, layers=layers),
        "state": {k: v.detach().clone() for k, v in m.state_dict().items()},
        "q": q.detach(), "ctx": ctx.detach(), "kvs": [k.detach() for k in kvs], "masks": masks,
        "out": out.detach(), "w": w,
        "dq": q.grad, "dctx": ctx.grad, "dkvs": [k.grad for k in kvs],
        "dparams": {n: p.grad.clone() for n, p in m.named_parameters()},
    }
    torch.save(fx, f"{OUT}/sva_small.pt")
    print("sva_small.pt", sum(v.numel() for v in fx["state"].values()), "params")


def golden_sva_k1024():
    """The REAL vision_sampler.py at dimensions the HIP kernels accept (hidden 1024, 16 heads x 64, kv windows [1,1,1,4],
    query grid 4 x 4, one image) so that the fixture replays straight through the HIP module (one hop: reference ->
    HIP), for q_dim 1024 (connector) and 4096 (in-LLM).  Weights come from tests/golden_recipes.fill_state (not stored:
    15.8 M / 22 M parameters); parameter gradients are stored as summaries (sum, norm, 3 seeded projections)."""
    sys.path.insert(0, os.path.dirname(HERE))
    from golden_recipes import fill_state, grad_summary, sva_k1024_inputs
    vs = load_ref_vision_sampler()
    fx = {}
    for q_dim in (1024, 4096):
        seed = 4242 + q_dim
        hidden, kv_sizes, qside, B = 1024, [1, 1, 1, 4], 4, 1
        m = vs.VisionTokenSampler(q_dim, hidden, [hidden] * 4, kv_sizes, hidden, 1).float()
        m.load_state_dict(fill_state(m.state_dict(), seed), strict=True)
        q, ctx, kvs, masks, w = sva_k1024_inputs(q_dim, hidden, kv_sizes, qside, B, seed)
        q.requires_grad_(); ctx.requires_grad_()
        kvs = [k.requires_grad_() for k in kvs]
        out = m(q, ctx, *kvs, *masks)
        (out * w).sum().backward()
        fx[q_dim] = {
            "cfg": dict(q_dim=q_dim, hidden=hidden, kv_sizes=kv_sizes, qside=qside, B=B, seed=seed),
            "out": out.detach(), "dq": q.grad, "dctx": ctx.grad, "dkvs": [k.grad for k in kvs],
            "dparams": {n: grad_summary(p.grad, n, seed) for n, p in m.named_parameters()},
        }
        print("sva_k1024 q_dim", q_dim, "out", tuple(out.shape), float(out.detach().abs().max()))
    torch.save(fx, f"{OUT}/sva_k1024.pt")


def golden_towers():
    """Outputs of the installed HF modules (the third-party arithmetic the reference delegates to) on seeded
    small configs; replayed through oracle/towers.py via cambrian_amd/.../weight_maps.py."""
    from transformers import (CLIPVisionConfig, CLIPVisionModel, ConvNextConfig, ConvNextModel, Dinov2Config,
                              Dinov2Model, SiglipVisionConfig, SiglipVisionModel)
    torch.manual_seed(4321)
    fx = {}

    def perturb(m):
        with torch.no_grad():
            for n, p_ in m.named_parameters():
                if p_.dim() == 1:
                    p_.add_(0.1 * torch.randn_like(p_))
                elif "position" in n or "cls" in n or "class" in n:
                    p_.add_(0.1 * torch.randn_like(p_))

    img = torch.randn(2, 3, 28, 28)
    m = CLIPVisionModel(CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3,
                                         num_attention_heads=4, image_size=28, patch_size=14, hidden_act="quick_gelu",
                                         layer_norm_eps=1e-5)).eval()
    perturb(m)
    with torch.no_grad():
        hs = m(img, output_hidden_states=True).hidden_states
    fx["clip"] = dict(sd=m.state_dict(), img=img, out=hs[-2][:, 1:].clone(), n_hidden=len(hs))
    m = Dinov2Model(Dinov2Config(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, image_size=28,
                                 patch_size=14, use_swiglu_ffn=True, mlp_ratio=4, layer_norm_eps=1e-6,
                                 layerscale_value=0.5)).eval()
    perturb(m)
    with torch.no_grad():
        out = m(img).last_hidden_state[:, 1:].clone()
    fx["dino_swiglu"] = dict(sd=m.state_dict(), img=img, out=out)
    m = Dinov2Model(Dinov2Config(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, image_size=28,
                                 patch_size=14, use_swiglu_ffn=False, mlp_ratio=2, layer_norm_eps=1e-6)).eval()
    perturb(m)
    with torch.no_grad():
        out = m(img).last_hidden_state[:, 1:].clone()
    fx["dino_mlp"] = dict(sd=m.state_dict(), img=img, out=out)
    m = SiglipVisionModel(SiglipVisionConfig(hidden_size=64, intermediate_size=144, num_hidden_layers=2,
                                             num_attention_heads=4, image_size=28, patch_size=14,
                                             hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6)).eval()
    perturb(m)
    with torch.no_grad():
        out = m(img).last_hidden_state.clone()
    fx["siglip"] = dict(sd={k: v for k, v in m.state_dict().items() if "head." not in k}, img=img, out=out)
    img2 = torch.randn(2, 3, 64, 64)
    m = ConvNextModel(ConvNextConfig(num_channels=3, hidden_sizes=[16, 32, 64, 128], depths=[1, 1, 2, 1],
                                     layer_scale_init_value=0.5)).eval()
    perturb(m)
    with torch.no_grad():
        hs = m(img2, output_hidden_states=True).hidden_states
    fx["convnext"] = dict(sd=m.state_dict(), img=img2, stages=[h.clone() for h in hs[1:]], depths=[1, 1, 2, 1],
                          dims=[16, 32, 64, 128])
    torch.save(fx, f"{OUT}/towers_small.pt")
    print("towers_small.pt written")


def golden_collator():
    """The reference's own get_padding_offset / prepare_image_info / prepare_multimodal_data on a set of cases
    (square, wide, tall, extreme aspect, image token at several positions, padded rows)."""
    ns = load_ref_collator()
    g = torch.Generator().manual_seed(99)
    cases = []
    for (image_token_len, aux_lens, max_len, sizes, positions) in [
        (576, [576, 576, 576, 9216], 2048, [(336, 336), (336, 224), (200, 640), (1000, 90)], [91, 35, 0, 1300]),
        (144, [144, 576, 2304], 512, [(100, 100), (640, 480), (90, 1000)], [5, 87, 200]),
        (16, [16, 64], 64, [(3, 2), (2, 3)], [0, 20]),
    ]:
        B = len(sizes)
        ids = torch.randint(1000, 30000, (B, max_len), generator=g)
        labels = ids.clone()
        att = torch.ones(B, max_len, dtype=torch.bool)
        for b, p in enumerate(positions):
            ids[b, p] = -200
            labels[b, :p + 1] = -100
        att[B - 1, max_len // 2:] = False   # a padded tail on the last row
        out = ns["prepare_multimodal_data"](ids.clone(), labels.clone(), att.clone(), sizes, image_token_len, aux_lens, max_len)
        cases.append(dict(image_token_len=image_token_len, aux_lens=aux_lens, max_len=max_len, sizes=sizes, ids=ids,
                          labels=labels, att=att, out_ids=out[0], out_labels=out[1], out_att=out[2], out_pos=out[3],
                          out_aux=out[4]))
    offs = {}
    for cur in [(24, 24), (96, 96), (12, 12), (4, 4)]:
        for orig in [(336, 336), (336, 224), (224, 336), (1000, 90), (90, 1000), (3, 2), (641, 479)]:
            offs[(cur, orig)] = ns["get_padding_offset"](cur, orig)
    info = {}
    for size in [(336, 336), (336, 224), (224, 336), (1000, 90)]:
        for tl in [576, 9216, 16]:
            for nl in [False, True]:
                m, p = ns["prepare_image_info"](size, tl, newline=nl)
                info[(size, tl, nl)] = (m, p)
    torch.save(dict(cases=cases, offsets=offs, info=info), f"{OUT}/collator_cases.pt")
    print("collator_cases.pt written")


def golden_arch_groups():
    """Same, with two query groups [16, 4]: the second group's 2 x 2 queries are resized to the final 4 x 4 grid
    (cambrian_arch.py:395-401) -> arch_groups_small.pt."""
    golden_arch(query_nums=[16, 4], out_name="arch_groups_small.pt")


def golden_arch(query_nums=None, out_name="arch_small.pt"):
    """The real CambrianMetaForCausalLM.prepare_inputs_labels_for_multimodal (static branch) with fake towers."""
    import torch.nn as nn
    A = load_ref_arch()
    ns = load_ref_collator()
    torch.manual_seed(777)
    H, vh, side, B, S, V = 96, 64, 4, 2, 64, 50
    tower_dims, token_lens = [48, 80], [16, 64]

    class FakeTower(nn.Module):
        def __init__(self, hidden, tokens):
            super().__init__()
            self.hidden_size, self.tokens, self.is_loaded = hidden, tokens, True
            self.out = None

        def load_model(self):
            pass

        def forward(self, images):
            return self.out

    towers = [FakeTower(d, t) for d, t in zip(tower_dims, token_lens)]

    class Cfg:
        pass

    cfg = Cfg()
    cfg.hidden_size, cfg.vision_hidden_size = H, vh
    cfg.mm_vision_tower_aux_list = ["a", "b"]
    cfg.mm_vision_tower_aux_token_len_list = token_lens
    cfg.mm_projector_type = "sva"
    query_nums = query_nums or [side * side]
    cfg.num_query_group, cfg.query_num_list, cfg.connector_only, cfg.connector_depth = len(query_nums), query_nums, False, 2
    cfg.image_token_len = side * side
    cfg.num_of_vision_sampler_layers, cfg.start_of_vision_sampler_layers, cfg.stride_of_vision_sampler_layers = 2, 0, 1
    cfg._fake_towers = towers

    class Base(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.config = config
            self.embed_tokens = nn.Embedding(V, H)

        @property
        def dtype(self):
            return torch.float32

    class Model(A.CambrianMetaModel, Base):
        pass

    class LM(nn.Module, A.CambrianMetaForCausalLM):
        def __init__(self):
            super().__init__()
            self.config = cfg
            self.model = Model(cfg)

        def get_model(self):
            return self.model

        @property
        def device(self):
            return torch.device("cpu")

    lm = LM()
    with torch.no_grad():
        lm.model.image_newline.copy_(torch.randn(H) / H ** 0.5)
        lm.model.vision_query.mul_(1 / vh ** 0.5)
        for n, p_ in lm.named_parameters():
            if p_.dim() == 1 and "newline" not in n:
                p_.add_(0.1 * torch.randn_like(p_))
    ids = torch.randint(1, V, (B, 40))
    labels = ids.clone()
    att = torch.ones(B, 40, dtype=torch.bool)
    ids[0, 5] = -200
    ids[1, 17] = -200
    sizes = [(336, 336), (336, 150)]
    new_ids, new_lab, new_att, new_pos, aux_masks = ns["prepare_multimodal_data"](ids, labels, att, sizes, side * side,
                                                                               token_lens, S)
    feats = [torch.randn(B, t, d, requires_grad=True) for t, d in zip(token_lens, tower_dims)]
    for t, f in zip(towers, feats):
        t.out = f
    images = [torch.zeros(B, 3, 8, 8) for _ in towers]
    out = lm.prepare_inputs_labels_for_multimodal(new_ids, new_pos, new_att, None, new_lab, images, aux_masks, sizes)
    emb = out[4]
    w = torch.randn_like(emb)
    (emb * w).sum().backward()
    fx = dict(cfg=dict(H=H, vh=vh, side=side, B=B, S=S, V=V, tower_dims=tower_dims, token_lens=token_lens,
                       connector_depth=2, n_in_llm=2, query_nums=list(query_nums)),
              state={k: v.detach().clone() for k, v in lm.model.state_dict().items()},
              ids=new_ids, pos=new_pos, att=new_att, labels=new_lab, aux_masks=aux_masks, sizes=sizes,
              feats=[f.detach().clone() for f in feats], embeds=emb.detach().clone(),
              kv_final=[t.detach().clone() for t in out[6]], mask_final=[t.clone() for t in out[7]],
              final_size=out[8], ctx_final=out[9].detach().clone(), w=w,
              dfeats=[f.grad.clone() for f in feats],
              dparams={n: p_.grad.clone() for n, p_ in lm.model.named_parameters() if p_.grad is not None})
    torch.save(fx, f"{OUT}/{out_name}")
    print("arch_small.pt written:", sorted(fx["dparams"].keys())[:4], "...")


def golden_llama():
    """(a) the reference's vendored Phi3RMSNorm / rotary embedding / apply_rotary_pos_emb
    (phi3/modeling_phi3.py:83-97,114-141,257-281) on seeded tensors; (b) the in-LLM SVA hook, static branch:
    cambrian_llama.py lines 181-207 exec'd verbatim with the reference VisionTokenSampler; (c) logits of the
    installed HF LlamaForCausalLM (tiny config) for the bare decoder arithmetic."""
    import textwrap
    load_ref_arch()  # registers the stub packages so the phi3 module can be imported by path
    spec = importlib.util.spec_from_file_location(
        "cambrian.model.language_model.phi3.modeling_phi3",
        f"{REF}/cambrian/model/language_model/phi3/modeling_phi3.py")
    for name, path in [("cambrian.model.language_model.phi3", f"{REF}/cambrian/model/language_model/phi3")]:
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    phi3 = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = phi3
    spec.loader.exec_module(phi3)
    torch.manual_seed(2024)
    fx = {}
    x = torch.randn(3, 10, 96)
    norm = phi3.Phi3RMSNorm(96, eps=1e-5)
    with torch.no_grad():
        norm.weight.add_(0.1 * torch.randn(96))
    fx["rms"] = dict(x=x, w=norm.weight.detach().clone(), eps=1e-5, out=norm(x).detach())
    rot = phi3.Phi3RotaryEmbedding(32, max_position_embeddings=4096, base=500000.0)
    pos = torch.randint(0, 3000, (2, 12))
    q, k = torch.randn(2, 4, 12, 32), torch.randn(2, 2, 12, 32)
    cos, sin = rot(q, pos, seq_len=None)
    qe, ke = phi3.apply_rotary_pos_emb(q, k, cos, sin, pos)
    fx["rope"] = dict(pos=pos, q=q, k=k, base=500000.0, cos=cos, sin=sin, q_out=qe, k_out=ke)

    # (b) hook: exec the reference lines verbatim
    vs = load_ref_vision_sampler()
    H, vh, side, B, S, p0 = 96, 64, 4, 2, 40, 7
    sampler = vs.VisionTokenSampler(H, vh, [vh, vh], [1, 2], vh, 1).float()
    src = open(f"{REF}/cambrian/model/language_model/cambrian_llama.py").read().split("\n")
    body = textwrap.dedent("\n".join(src[176:207]))  # lines 177-207: the IS_XLA_AVAILABLE branch body
    # :187 `latent_query.view(bs*latent_query_num, 1, -1)` is applied to a non-contiguous slice: legal on XLA
    # (functional views), a RuntimeError on eager CPU/GPU torch.  Same values with reshape.
    assert "latent_query = latent_query.view(bs*latent_query_num, 1, -1)" in body
    body = body.replace("latent_query = latent_query.view(bs*latent_query_num, 1, -1)",
                        "latent_query = latent_query.reshape(bs*latent_query_num, 1, -1)")

    class Self:
        pass

    me = Self()
    me.config = types.SimpleNamespace(image_token_len=side * side, image_position=p0)
    me.gradient_checkpointing, me.training = False, False
    me.vision_sampler_layers = [sampler]
    hidden = torch.randn(B, S, H)
    kvs = [torch.randn(B * side * side, s * s, vh) for s in (1, 2)]
    masks = [torch.ones(B * side * side, s * s, dtype=torch.bool) for s in (1, 2)]
    masks[1][3, :2] = False
    ctx = torch.randn(B * side * side, 1, vh)
    ns = dict(self=me, torch=torch, hidden_states=hidden.clone(), latent_query_start_idx=p0,
              vision_tower_aux_feature_list=kvs, vision_tower_aux_attention_masks_list=masks,
              global_context_feature=ctx, i=0, cross_layers_start_idx=0, cross_index_step=1, IS_XLA_AVAILABLE=True)
    with torch.no_grad():
        exec(body, ns)
    fx["hook"] = dict(cfg=dict(H=H, vh=vh, side=side, B=B, S=S, p0=p0), state={k: v.detach().clone() for k, v in sampler.state_dict().items()},
                      hidden=hidden, kvs=kvs, masks=masks, ctx=ctx, out=ns["hidden_states"].detach().clone())

    # (c) bare decoder
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(vocab_size=101, hidden_size=64, intermediate_size=160, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, rms_norm_eps=1e-5, rope_theta=500000.0, max_position_embeddings=128,
                      attn_implementation="eager")
    lm = LlamaForCausalLM(cfg).eval()
    ids = torch.randint(0, 101, (2, 24))
    pos = torch.arange(24)[None].expand(2, -1)
    with torch.no_grad():
        logits = lm(input_ids=ids, position_ids=pos).logits
    fx["decoder"] = dict(state={k: v.detach().clone() for k, v in lm.state_dict().items()}, ids=ids, pos=pos, logits=logits,
                         cfg=dict(hidden_size=64, intermediate_size=160, num_hidden_layers=2, num_attention_heads=4,
                                  num_key_value_heads=2, rms_norm_eps=1e-5, rope_theta=500000.0, vocab_size=101))
    torch.save(fx, f"{OUT}/llama_small.pt")
    print("llama_small.pt written")


def golden_arch_dynamic(query_nums=None):
    """The real prepare_inputs_labels_for_multimodal with IS_XLA_AVAILABLE = False: the eval / generate branch
    (cambrian_arch.py:289-330 rearrange_inference + unpad, :422-451 per-sample unpad + newline, :492-609 variable-length
    merge) with fake towers, non-square images and a padded batch; plus the dynamic in-LLM hook, cambrian_llama.py lines
    209-253 exec'd verbatim."""
    import textwrap
    import torch.nn as nn
    A = load_ref_arch()
    A.IS_XLA_AVAILABLE = False
    torch.manual_seed(4242)
    H, vh, side, B, V = 128, 64, 4, 3, 50          # GEMM-friendly widths: the GPU test runs the HIP path on this fixture
    tower_dims, token_lens = [64, 128], [16, 64]

    class FakeTower(nn.Module):
        def __init__(self, hidden, tokens):
            super().__init__()
            self.hidden_size, self.tokens, self.is_loaded = hidden, tokens, True
            self.out = None

        def load_model(self):
            pass

        def forward(self, images):
            return self.out

    towers = [FakeTower(d, t) for d, t in zip(tower_dims, token_lens)]

    class Cfg:
        pass

    cfg = Cfg()
    cfg.hidden_size, cfg.vision_hidden_size = H, vh
    cfg.mm_vision_tower_aux_list = ["a", "b"]
    cfg.mm_vision_tower_aux_token_len_list = token_lens
    cfg.mm_projector_type = "sva"
    query_nums = query_nums or [side * side]
    cfg.num_query_group, cfg.query_num_list, cfg.connector_only, cfg.connector_depth = len(query_nums), query_nums, False, 2
    cfg.image_token_len = side * side
    cfg.num_of_vision_sampler_layers, cfg.start_of_vision_sampler_layers, cfg.stride_of_vision_sampler_layers = 2, 0, 1
    cfg._fake_towers = towers

    class Base(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.config = config
            self.embed_tokens = nn.Embedding(V, H)

        @property
        def dtype(self):
            return torch.float32

    class Model(A.CambrianMetaModel, Base):
        pass

    class LM(nn.Module, A.CambrianMetaForCausalLM):
        def __init__(self):
            super().__init__()
            self.config = cfg
            self.model = Model(cfg)

        def get_model(self):
            return self.model

        @property
        def device(self):
            return torch.device("cpu")

    lm = LM()
    with torch.no_grad():
        lm.model.image_newline.copy_(torch.randn(H) / H ** 0.5)
        lm.model.vision_query.mul_(1 / vh ** 0.5)
        for n, p_ in lm.named_parameters():
            if p_.dim() == 1 and "newline" not in n:
                p_.add_(0.1 * torch.randn_like(p_))
    L = 30
    ids = torch.randint(1, V, (B, L))
    att = torch.ones(B, L, dtype=torch.bool)
    ids[0, 5] = -200
    ids[1, 17] = -200
    ids[2, 2] = -200
    att[1, 24:] = False          # right-padded prompt
    sizes = [(336, 336), (336, 150), (120, 400)]   # square, wide (rows unpadded), tall (columns unpadded)
    feats = [torch.randn(B, t, d) for t, d in zip(token_lens, tower_dims)]
    for t, f in zip(towers, feats):
        t.out = f
    images = [torch.zeros(B, 3, 8, 8) for _ in towers]
    with torch.no_grad():
        out = lm.prepare_inputs_labels_for_multimodal(ids.clone(), None, att.clone(), None, None, images, None, sizes)
    fx = dict(cfg=dict(H=H, vh=vh, side=side, B=B, V=V, tower_dims=tower_dims, token_lens=token_lens, connector_depth=2,
                       n_in_llm=2),
              state={k: v.detach().clone() for k, v in lm.model.state_dict().items()},
              ids=ids, att=att, sizes=sizes, feats=[f.clone() for f in feats],
              out_pos=out[1], out_att=out[2], embeds=out[4].clone(), out_labels=out[5],
              kv_final=[t.clone() for t in out[6]], mask_final=[t.clone() for t in out[7]], final_size=out[8],
              ctx_final=out[9].clone())

    # dynamic hook: lines 209-253 of cambrian_llama.py (the `else:` body of `if IS_XLA_AVAILABLE:`)
    src = open(f"{REF}/cambrian/model/language_model/cambrian_llama.py").read().split("\n")
    assert src[207].strip() == "else:", src[207]
    body = textwrap.dedent("\n".join(src[208:253]))

    class Self:
        pass

    me = Self()
    me.gradient_checkpointing, me.training = False, False
    me.vision_sampler_layers = [lm.model.vision_sampler_layers[0]]
    S2 = max(e.shape[0] for e in [out[4][b] for b in range(B)])
    hidden = torch.randn(B, out[4].shape[1], H)
    p0 = 2
    ns = dict(self=me, torch=torch, hidden_states=hidden.clone(), latent_query_start_idx=p0,
              final_vision_feature_size=out[8], vision_tower_aux_feature_list=out[6],
              vision_tower_aux_attention_masks_list=out[7], global_context_feature=out[9], i=0, cross_layers_start_idx=0,
              cross_index_step=1)
    with torch.no_grad():
        exec(body, ns)
    fx["hook"] = dict(p0=p0, hidden=hidden, out=ns["hidden_states"].detach().clone())
    torch.save(fx, f"{OUT}/arch_dynamic_small.pt")
    A.IS_XLA_AVAILABLE = True
    print("arch_dynamic_small.pt written: embeds", tuple(out[4].shape), "final sizes", out[8])


def load_ref_phi3():
    """The reference's vendored Phi-3 (cambrian/model/language_model/phi3/) by path, with its static (XLA) branch on."""
    load_ref_arch()
    name = "cambrian.model.language_model.phi3"
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__path__ = [f"{REF}/cambrian/model/language_model/phi3"]
        sys.modules[name] = m
    mods = {}
    for part in ("configuration_phi3", "modeling_phi3"):
        full = f"{name}.{part}"
        if full not in sys.modules:
            spec = importlib.util.spec_from_file_location(full, f"{REF}/cambrian/model/language_model/phi3/{part}.py")
            mod = importlib.util.module_from_spec(spec)
            sys.modules[full] = mod
            spec.loader.exec_module(mod)
        mods[part] = sys.modules[full]
    mods["modeling_phi3"].IS_XLA_AVAILABLE = True
    return mods["configuration_phi3"], mods["modeling_phi3"]


def golden_config0():
    """BASELINE configs[0] (one CLIP tower + mlp2x_gelu projector into Phi-3), small:
    (a) the real prepare_inputs_labels_for_multimodal, static branch, mm_projector_type = 'mlp2x_gelu'
        (cambrian_arch.py:79-87 construction, :407-420 concat -> projector -> newline, :457-490 splice), fake tower;
    (b) the reference's vendored Phi3ForCausalLM (phi3/modeling_phi3.py) on CPU: bare decoder logits, the same with a
        sliding window shorter than the sequence (eager mask, :1180-1186), and with the in-LLM SVA hook
        (:1221-1260, static branch) driven by real VisionTokenSampler layers."""
    import torch.nn as nn
    A = load_ref_arch()
    ns = load_ref_collator()
    torch.manual_seed(31337)
    H, side, B, S, V, tdim = 64, 4, 2, 48, 60, 40

    class FakeTower(nn.Module):
        def __init__(self, hidden, tokens):
            super().__init__()
            self.hidden_size, self.tokens, self.is_loaded = hidden, tokens, True
            self.out = None

        def load_model(self):
            pass

        def forward(self, images):
            return self.out

    towers = [FakeTower(tdim, side * side)]

    class Cfg:
        pass

    cfg = Cfg()
    cfg.hidden_size = H
    cfg.mm_vision_tower_aux_list = ["clip"]
    cfg.mm_vision_tower_aux_token_len_list = [side * side]
    cfg.mm_projector_type = "mlp2x_gelu"
    cfg.image_token_len = side * side
    cfg.query_num_list = [side * side]
    cfg.connector_only = True
    cfg._fake_towers = towers

    class Base(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.config = config
            self.embed_tokens = nn.Embedding(V, H)

        @property
        def dtype(self):
            return torch.float32

    class Model(A.CambrianMetaModel, Base):
        pass

    class LM(nn.Module, A.CambrianMetaForCausalLM):
        def __init__(self):
            super().__init__()
            self.config = cfg
            self.model = Model(cfg)

        def get_model(self):
            return self.model

        @property
        def device(self):
            return torch.device("cpu")

    lm = LM()
    with torch.no_grad():
        lm.model.image_newline.copy_(torch.randn(H) / H ** 0.5)
    ids = torch.randint(1, V, (B, 28))
    labels = ids.clone()
    att = torch.ones(B, 28, dtype=torch.bool)
    ids[0, 3] = -200
    ids[1, 9] = -200
    sizes = [(224, 224), (224, 100)]
    new_ids, new_lab, new_att, new_pos, aux_masks = ns["prepare_multimodal_data"](ids, labels, att, sizes, side * side,
                                                                               [side * side], S)
    feat = torch.randn(B, side * side, tdim, requires_grad=True)
    towers[0].out = feat
    out = lm.prepare_inputs_labels_for_multimodal(new_ids, new_pos, new_att, None, new_lab, [torch.zeros(B, 3, 8, 8)],
                                                  aux_masks, sizes)
    emb = out[4]
    assert out[6] is None and out[7] is None and out[9] is None
    w = torch.randn_like(emb)
    (emb * w).sum().backward()
    fx = {"arch": dict(cfg=dict(H=H, side=side, B=B, S=S, V=V, tower_dim=tdim),
                       state={k: v.detach().clone() for k, v in lm.model.state_dict().items()},
                       ids=new_ids, pos=new_pos, att=new_att, labels=new_lab, sizes=sizes, feat=feat.detach().clone(),
                       embeds=emb.detach().clone(), final_size=out[8], w=w, dfeat=feat.grad.clone(),
                       dparams={n: p_.grad.clone() for n, p_ in lm.model.named_parameters() if p_.grad is not None})}

    # ---- (b) vendored Phi-3 -----------------------------------------------------------------------------------
    C, M = load_ref_phi3()
    vs = load_ref_vision_sampler()
    geo = dict(vocab_size=V, hidden_size=H, intermediate_size=96, num_hidden_layers=3, num_attention_heads=2,
               num_key_value_heads=2, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=128)

    def make(sliding_window, connector_only):
        c = C.Phi3Config(original_max_position_embeddings=128, sliding_window=sliding_window, pad_token_id=0,
                         bos_token_id=1, eos_token_id=2, attn_implementation="eager", **geo)
        c.rope_scaling = None            # transformers 5.x turns it into a dict the 4.37-era _init_rope cannot read
        c.connector_only = connector_only
        return c

    torch.manual_seed(99)
    bare = M.Phi3ForCausalLM(make(None, True)).eval()
    state = {k: v.detach().clone() for k, v in bare.state_dict().items()}
    S2 = 24
    pids = torch.randint(1, V, (B, S2))
    pos = torch.arange(S2)[None].expand(B, -1).contiguous()
    with torch.no_grad():
        logits = bare(input_ids=pids, position_ids=pos, use_cache=False, return_dict=True).logits
    win = M.Phi3ForCausalLM(make(8, True)).eval()
    win.load_state_dict(state)
    with torch.no_grad():
        logits_win = win(input_ids=pids, position_ids=pos, use_cache=False, return_dict=True).logits
    assert (logits - logits_win).abs().max() > 1e-4   # the window really bites

    vh, p0 = 32, 2
    hk = make(None, False)
    hk.image_token_len, hk.image_position = side * side, p0
    hk.start_of_vision_sampler_layers, hk.stride_of_vision_sampler_layers = 0, 2
    hooked = M.Phi3ForCausalLM(hk).eval()
    hooked.load_state_dict(state)
    samplers = nn.ModuleList([vs.VisionTokenSampler(H, vh, [vh, vh], [1, 2], vh, 1).float() for _ in range(2)])
    hooked.model.vision_sampler_layers = samplers
    kvs = [torch.randn(B * side * side, s * s, vh) for s in (1, 2)]
    masks = [torch.ones(B * side * side, s * s, dtype=torch.bool) for s in (1, 2)]
    masks[1][5, 1:] = False
    ctx = torch.randn(B * side * side, 1, vh)
    emb2 = torch.randn(B, S2, H)
    # :1241 `latent_query.view(bs*latent_query_num, 1, -1)` is applied to a non-contiguous slice: legal on XLA (functional
    # views), a RuntimeError on eager CPU torch.  Same values with reshape — patched in for the duration of the call.
    orig_view = torch.Tensor.view

    def lenient_view(self, *shape, **kw):
        try:
            return orig_view(self, *shape, **kw)
        except RuntimeError:
            return self.reshape(*shape)
    torch.Tensor.view = lenient_view
    with torch.no_grad():
        logits_hook = hooked(inputs_embeds=emb2, position_ids=pos, use_cache=False, return_dict=True,
                             vision_tower_aux_feature_list=kvs, vision_tower_aux_attention_masks_list=masks,
                             final_vision_feature_size=[(side, side)] * B, global_context_feature=ctx).logits
        logits_nohook = hooked(inputs_embeds=emb2, position_ids=pos, use_cache=False, return_dict=True).logits
    torch.Tensor.view = orig_view
    assert (logits_hook - logits_nohook).abs().max() > 1e-3
    fx["phi3"] = dict(cfg=dict(geo, vh=vh, side=side, p0=p0, start=0, stride=2), state=state, ids=pids, pos=pos,
                      logits=logits, sliding_window=8, logits_window=logits_win,
                      sampler_state={k: v.detach().clone() for k, v in samplers.state_dict().items()},
                      embeds=emb2, kvs=kvs, masks=masks, ctx=ctx, logits_hook=logits_hook)
    torch.save(fx, f"{OUT}/config0_small.pt")
    print("config0_small.pt written; arch dparams:", sorted(fx["arch"]["dparams"].keys()))


def golden_sampler():
    """get_length_grouped_indices / get_modality_length_grouped_indices / split_to_even_chunks of
    cambrian/train/cambrian_trainer.py (lines 69-130 exec'd: the module itself imports torch_xla / gcsfs)."""
    src = open(f"{REF}/cambrian/train/cambrian_trainer.py").read().split("\n")
    ns = {"torch": torch}
    exec("\n".join(src[68:130]), ns)
    cases = []
    g = torch.Generator().manual_seed(5)
    for n, bs, ws, mixed in [(64, 4, 2, False), (100, 4, 8, False), (37, 2, 4, False), (96, 4, 2, True), (203, 8, 4, True),
                             (50, 2, 8, True), (16, 4, 4, False)]:
        lengths = torch.randint(1, 2000, (n,), generator=g).tolist()
        if mixed:
            sign = torch.randint(0, 3, (n,), generator=g).tolist()
            lengths = [l if s else -l for l, s in zip(lengths, sign)]
        torch.manual_seed(1000 + n)          # the per-modality grouping draws from the GLOBAL generator
        gen = torch.Generator().manual_seed(n * 7 + bs)
        fn = ns["get_modality_length_grouped_indices"] if mixed else ns["get_length_grouped_indices"]
        out = fn(lengths, bs, ws, generator=gen)
        cases.append(dict(lengths=lengths, batch_size=bs, world_size=ws, mixed=mixed, global_seed=1000 + n,
                          gen_seed=n * 7 + bs, indices=list(out)))
    chunks = []
    for n, k in [(12, 4), (13, 4), (8, 8), (30, 3)]:
        lengths = torch.randint(1, 50, (n,), generator=g).tolist()
        idx = sorted(range(n), key=lambda i: lengths[i], reverse=True)
        chunks.append(dict(indices=idx, lengths=lengths, k=k, out=ns["split_to_even_chunks"](idx, lengths, k)))
    torch.save(dict(cases=cases, chunks=chunks), f"{OUT}/sampler_cases.pt")
    print("sampler_cases.pt written:", len(cases), "cases")


if __name__ == "__main__":
    which = sys.argv[1:] or ["sva"]
    for w in which:
        globals()[f"golden_{w}"]()
