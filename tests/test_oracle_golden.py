"""CPU: the oracle (oracle/*.py) replayed against golden vectors produced by the REFERENCE's own code
(tests/golden/make_golden.py ran /root/reference on seeded inputs in the build container)."""
import os

import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    path = os.path.join(GOLD, name)
    if not os.path.exists(path):
        pytest.fail(f"golden fixture {name} is missing; run tests/golden/make_golden.py in the build container")
    return torch.load(path, weights_only=False)


def test_sva_oracle_matches_reference_forward_and_backward():
    from oracle import sva
    fx = _load("sva_small.pt")
    p = {k: v.clone().requires_grad_() for k, v in fx["state"].items()}
    q, ctx = fx["q"].clone().requires_grad_(), fx["ctx"].clone().requires_grad_()
    kvs = [k.clone().requires_grad_() for k in fx["kvs"]]
    out = sva.vision_token_sampler(p, q, ctx, kvs, fx["masks"])
    assert torch.allclose(out, fx["out"], atol=2e-6, rtol=1e-5)
    (out * fx["w"]).sum().backward()
    assert torch.allclose(q.grad, fx["dq"], atol=5e-6, rtol=1e-4)
    assert torch.allclose(ctx.grad, fx["dctx"], atol=5e-6, rtol=1e-4)
    for a, b in zip(kvs, fx["dkvs"]):
        assert torch.allclose(a.grad, b, atol=5e-6, rtol=1e-4)
    for name, g in fx["dparams"].items():
        assert torch.allclose(p[name].grad, g, atol=2e-5, rtol=1e-4), name


@pytest.mark.parametrize("q_dim", [1024, 4096])
def test_sva_oracle_matches_reference_at_kernel_dims(q_dim):
    """sva_k1024.pt: the real vision_sampler.py at the dimensions the HIP kernels run (hidden 1024, windows [1,1,1,4]);
    weights and inputs are regenerated from the fixture's seed (tests/golden_recipes.py)."""
    from golden_recipes import fill_state, grad_summary, sva_k1024_inputs
    from oracle import sva
    fx = _load("sva_k1024.pt")[q_dim]
    c = fx["cfg"]
    shapes = sva.init_sampler_params(c["q_dim"], c["hidden"], [c["hidden"]] * 4, c["kv_sizes"], c["hidden"], 1,
                                     torch.Generator().manual_seed(0))
    p = {k: v.requires_grad_() for k, v in fill_state(shapes, c["seed"]).items()}
    q, ctx, kvs, masks, w = sva_k1024_inputs(c["q_dim"], c["hidden"], c["kv_sizes"], c["qside"], c["B"], c["seed"])
    q.requires_grad_(); ctx.requires_grad_()
    kvs = [k.requires_grad_() for k in kvs]
    out = sva.vision_token_sampler(p, q, ctx, kvs, masks)
    assert torch.allclose(out, fx["out"], atol=1e-5, rtol=1e-4)
    (out * w).sum().backward()
    assert torch.allclose(q.grad, fx["dq"], atol=2e-5, rtol=1e-3)
    assert torch.allclose(ctx.grad, fx["dctx"], atol=2e-5, rtol=1e-3)
    for a, b in zip(kvs, fx["dkvs"]):
        assert torch.allclose(a.grad, b, atol=2e-5, rtol=1e-3)
    assert set(fx["dparams"]) == set(p)
    for name, want in fx["dparams"].items():
        got = grad_summary(p[name].grad, name, c["seed"])
        assert torch.allclose(got, want, atol=1e-3 * float(want[1]) + 1e-5, rtol=0), name


def test_sva_oracle_mask_size_check():
    from oracle import sva
    fx = _load("sva_small.pt")
    bad = [m.clone() for m in fx["masks"]]
    bad[2] = bad[2][:, :3]
    with pytest.raises(ValueError, match="Attention mask should be of size"):
        sva.vision_token_sampler(fx["state"], fx["q"], fx["ctx"], fx["kvs"], bad)


# ------------------------------------------------------------------------------------------------ towers
def _vit_cfg(**kw):
    from cambrian_amd.model.multimodal_encoder.vit import ViTConfig
    return ViTConfig(**kw)


def test_tower_oracle_matches_hf_clip():
    from oracle import towers
    from cambrian_amd.model.multimodal_encoder import weight_maps as W
    fx = _load("towers_small.pt")["clip"]
    cfg = _vit_cfg(image_size=28, patch_size=14, hidden_size=64, num_layers=3, num_heads=4, mlp_dim=128, act="quick_gelu",
                   ln_eps=1e-5, has_cls=True, pre_ln=True, final_ln=False, patch_bias=False, run_layers=3 + 1 - 2)
    out = towers.vit_forward(cfg, W.hf_clip_to_canonical(fx["sd"], 3), fx["img"])  # hidden_states[-2], CLS dropped
    assert torch.allclose(out, fx["out"], atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("key,swiglu,mlp_dim", [("dino_swiglu", True, 176), ("dino_mlp", False, 128)])
def test_tower_oracle_matches_hf_dinov2(key, swiglu, mlp_dim):
    from oracle import towers
    from cambrian_amd.model.multimodal_encoder import weight_maps as W
    fx = _load("towers_small.pt")[key]
    cfg = _vit_cfg(image_size=28, patch_size=14, hidden_size=64, num_layers=2, num_heads=4, mlp_dim=mlp_dim,
                   act="swiglu" if swiglu else "gelu", ln_eps=1e-6, has_cls=True, final_ln=True, layerscale=True)
    out = towers.vit_forward(cfg, W.hf_dinov2_to_canonical(fx["sd"], 2, swiglu), fx["img"])
    assert torch.allclose(out, fx["out"], atol=2e-5, rtol=1e-4)


def test_tower_oracle_matches_hf_siglip():
    from oracle import towers
    from cambrian_amd.model.multimodal_encoder import weight_maps as W
    fx = _load("towers_small.pt")["siglip"]
    cfg = _vit_cfg(image_size=28, patch_size=14, hidden_size=64, num_layers=2, num_heads=4, mlp_dim=144, act="gelu_tanh",
                   ln_eps=1e-6, has_cls=False, final_ln=True)
    out = towers.vit_forward(cfg, W.hf_siglip_to_canonical(fx["sd"], 2), fx["img"])
    assert torch.allclose(out, fx["out"], atol=2e-5, rtol=1e-4)


def test_tower_oracle_matches_hf_convnext():
    from oracle import towers
    from cambrian_amd.model.multimodal_encoder import weight_maps as W
    from cambrian_amd.model.multimodal_encoder.convnext import ConvNeXtConfig
    fx = _load("towers_small.pt")["convnext"]
    cfg = ConvNeXtConfig(depths=fx["depths"], dims=fx["dims"], ln_eps=1e-6)
    stages = towers.convnext_stages(cfg, W.hf_convnext_to_canonical(fx["sd"], fx["depths"]), fx["img"])
    for a, b in zip(stages, fx["stages"]):
        assert torch.allclose(a, b, atol=5e-5, rtol=1e-4)


def test_dinov2_pos_interpolation_follows_the_4_37_pin():
    """transformers==4.37.0 Dinov2Embeddings.interpolate_pos_encoding: bicubic with scale_factor=(g+0.1)/sqrt(N)."""
    import torch.nn.functional as F
    from cambrian_amd.model.multimodal_encoder.dino_encoder import interpolate_pos_encoding
    g = torch.Generator().manual_seed(0)
    pos = torch.randn(1 + 37 * 37, 32, generator=g)
    out = interpolate_pos_encoding(pos, 27)
    assert out.shape == (1 + 27 * 27, 32) and torch.equal(out[0], pos[0])
    ref = F.interpolate(pos[1:].reshape(1, 37, 37, 32).permute(0, 3, 1, 2), scale_factor=(27.1 / 37, 27.1 / 37),
                        mode="bicubic", align_corners=False).permute(0, 2, 3, 1).reshape(-1, 32)
    assert torch.allclose(out[1:], ref, atol=1e-6)
    assert torch.equal(interpolate_pos_encoding(pos, 37), pos)


# ------------------------------------------------------------------------------------------- glue / collator
def test_collator_oracle_bit_exact_vs_reference():
    """train_fsdp.py:1039-1165 — INT/bool arithmetic: bit-exact."""
    from oracle import arch
    fx = _load("collator_cases.pt")
    for (cur, orig), want in fx["offsets"].items():
        assert arch.get_padding_offset(cur, orig) == tuple(want)
    for (size, tl, nl), (m, p) in fx["info"].items():
        gm, gp = arch.prepare_image_info(size, tl, newline=nl)
        assert torch.equal(gm, m) and torch.equal(gp, p)
    for c in fx["cases"]:
        ids, lab, att, pos, aux = arch.prepare_multimodal_data(c["ids"], c["labels"], c["att"], c["sizes"],
                                                               c["image_token_len"], c["aux_lens"], c["max_len"])
        assert torch.equal(ids, c["out_ids"]) and torch.equal(lab, c["out_labels"])
        assert torch.equal(att, c["out_att"]) and torch.equal(pos, c["out_pos"])
        assert len(aux) == len(c["out_aux"])
        for a, b in zip(aux, c["out_aux"]):
            assert a.dtype == torch.bool and torch.equal(a, b)


class _NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


@pytest.mark.parametrize("fixture", ["arch_small.pt", "arch_groups_small.pt"])
def test_arch_oracle_matches_reference_prepare_inputs(fixture):
    """cambrian_arch.py:340-490 static branch, forward and gradients, plus the bit-exact window gather (:271-287);
    arch_groups_small.pt: two query groups, the second resized from 2 x 2 to the final grid (:395-401, S5)."""
    from oracle import arch
    fx = _load(fixture)
    c = fx["cfg"]
    cfg = _NS(image_token_len=c["side"] ** 2, query_num_list=c.get("query_nums", [c["side"] ** 2]))
    p = {k: v.clone().requires_grad_() for k, v in fx["state"].items()}
    feats = [f.clone().requires_grad_() for f in fx["feats"]]
    emb, kv_final, mask_final, ctx_final = arch.prepare_inputs_static(p, cfg, fx["ids"], feats, fx["aux_masks"],
                                                                      p["embed_tokens.weight"])
    assert torch.allclose(emb, fx["embeds"], atol=2e-6, rtol=1e-5)
    for a, b in zip(kv_final, fx["kv_final"]):
        assert torch.allclose(a, b, atol=2e-6, rtol=1e-5)
    for a, b in zip(mask_final, fx["mask_final"]):
        assert torch.equal(a, b)
    assert torch.allclose(ctx_final, fx["ctx_final"], atol=2e-6, rtol=1e-5)
    (emb * fx["w"]).sum().backward()
    for a, b in zip(feats, fx["dfeats"]):
        assert torch.allclose(a.grad, b, atol=1e-5, rtol=1e-4)
    for name, g in fx["dparams"].items():
        if name == "embed_tokens.weight":
            continue
        assert torch.allclose(p[name].grad, g, atol=2e-5, rtol=1e-4), name
    # text rows / visual rows are pure copies
    pos0 = int((fx["ids"][0] == -200).nonzero()[0])
    assert torch.equal(emb[0, :pos0], p["embed_tokens.weight"][fx["ids"][0, :pos0]])


# ------------------------------------------------------------------------------------------------ LLM side
def test_rmsnorm_rope_oracle_match_reference_phi3():
    """phi3/modeling_phi3.py:83-97 (RMSNorm), :114-141 + :257-281 (RoPE) — the reference's own classes."""
    from oracle import llama
    fx = _load("llama_small.pt")
    r = fx["rms"]
    assert torch.allclose(llama.rms_norm(r["x"], r["w"], r["eps"]), r["out"], atol=1e-6, rtol=1e-6)
    t = fx["rope"]
    cos, sin = llama.rope_cos_sin(t["pos"], 32, t["base"])
    assert torch.allclose(cos, t["cos"], atol=1e-6) and torch.allclose(sin, t["sin"], atol=1e-6)
    q, k = llama.apply_rope(t["q"], t["k"], cos, sin)
    assert torch.allclose(q, t["q_out"], atol=1e-6) and torch.allclose(k, t["k_out"], atol=1e-6)


def test_hook_oracle_matches_reference_lines():
    """cambrian_llama.py:177-207 exec'd verbatim (tests/golden/make_golden.py) vs oracle.llama.sva_hook."""
    from oracle import llama
    fx = _load("llama_small.pt")["hook"]
    c = fx["cfg"]
    out = llama.sva_hook(fx["hidden"], fx["state"], "", c["p0"], c["side"] ** 2, fx["ctx"], fx["kvs"], fx["masks"])
    assert torch.allclose(out, fx["out"], atol=2e-6, rtol=1e-5)
    keep = torch.ones(c["S"], dtype=torch.bool)
    keep[c["p0"]:c["p0"] + c["side"] * (c["side"] + 1)].view(c["side"], c["side"] + 1)[:, :c["side"]] = False
    assert torch.equal(out[:, keep], fx["hidden"][:, keep])  # text rows and the newline column untouched


def test_decoder_oracle_matches_hf_llama():
    from oracle import llama
    fx = _load("llama_small.pt")["decoder"]
    cfg = _NS(**fx["cfg"])
    p = fx["state"]
    emb = p["model.embed_tokens.weight"][fx["ids"]]
    hidden = llama.decoder_forward(p, cfg, emb, fx["pos"])
    _, logits = llama.lm_loss(hidden, p["lm_head.weight"], fx["ids"])
    assert torch.allclose(logits, fx["logits"], atol=2e-5, rtol=1e-4)


# ------------------------------------------------------------------------------------------------ dynamic (eval) branch
def test_arch_oracle_dynamic_matches_reference():
    """cambrian_arch.py eval branch (IS_XLA_AVAILABLE False) run for real by make_golden.py::golden_arch_dynamic —
    unpad / unmask, variable-length merge with a padded row, in-LLM KV lists — vs oracle.arch.prepare_inputs_dynamic."""
    from oracle import arch
    fx = _load("arch_dynamic_small.pt")
    c = fx["cfg"]
    cfg = _NS(image_token_len=c["side"] ** 2, query_num_list=[c["side"] ** 2])
    p = fx["state"]
    emb, att, kv_final, mask_final, final_size, ctx_final = arch.prepare_inputs_dynamic(
        p, cfg, fx["ids"], fx["att"], fx["feats"], fx["sizes"], p["embed_tokens.weight"])
    assert [tuple(s) for s in final_size] == [tuple(s) for s in fx["final_size"]] == [(4, 4), (2, 4), (4, 2)]
    assert emb.shape == fx["embeds"].shape and torch.allclose(emb, fx["embeds"], atol=2e-6, rtol=1e-5)
    assert torch.equal(att, fx["out_att"])
    for a, b in zip(kv_final, fx["kv_final"]):
        assert a.shape == b.shape and torch.allclose(a, b, atol=2e-6, rtol=1e-5)
    for a, b in zip(mask_final, fx["mask_final"]):
        assert torch.equal(a, b)
    assert torch.allclose(ctx_final, fx["ctx_final"], atol=2e-6, rtol=1e-5)
    assert fx["out_pos"] is None and fx["out_labels"] is None   # position_ids / labels None in -> None out (:601-607)


def test_hook_oracle_dynamic_matches_reference_lines():
    """cambrian_llama.py:209-253 exec'd verbatim vs oracle.llama.sva_hook_dynamic."""
    from oracle import llama
    fx = _load("arch_dynamic_small.pt")
    hk = fx["hook"]
    out = llama.sva_hook_dynamic(hk["hidden"], fx["state"], "vision_sampler_layers.0.", hk["p0"], fx["final_size"],
                                 fx["ctx_final"], fx["kv_final"], fx["mask_final"])
    assert torch.allclose(out, hk["out"], atol=2e-6, rtol=1e-5)
    assert not torch.equal(out, hk["hidden"])


# ------------------------------------------------------------------------------------------------ BASELINE configs[0]
def test_arch_oracle_mlp_projector_branch_matches_reference():
    """cambrian_arch.py:79-87,407-420,457-490: one tower + mlp2x_gelu projector (no SVA), forward and gradients."""
    from oracle import arch
    fx = _load("config0_small.pt")["arch"]
    c = fx["cfg"]
    cfg = _NS(image_token_len=c["side"] ** 2, query_num_list=[c["side"] ** 2], mm_projector_type="mlp2x_gelu")
    p = {k: v.clone().requires_grad_() for k, v in fx["state"].items()}
    feat = fx["feat"].clone().requires_grad_()
    emb, kv, masks, ctx = arch.prepare_inputs_static(p, cfg, fx["ids"], [feat], None, p["embed_tokens.weight"])
    assert kv is None and masks is None and ctx is None
    assert torch.allclose(emb, fx["embeds"], atol=2e-6, rtol=1e-5)
    (emb * fx["w"]).sum().backward()
    assert torch.allclose(feat.grad, fx["dfeat"], atol=1e-5, rtol=1e-4)
    for name, g in fx["dparams"].items():
        if name != "embed_tokens.weight":
            assert torch.allclose(p[name].grad, g, atol=2e-5, rtol=1e-4), name


def test_decoder_oracle_matches_reference_phi3():
    """The reference's vendored Phi3ForCausalLM (phi3/modeling_phi3.py) run on CPU: packed qkv / gate_up weights, the
    sliding-window mask, and the in-LLM SVA hook twin (:1221-1260) with real VisionTokenSampler layers."""
    from oracle import llama
    fx = _load("config0_small.pt")["phi3"]
    c = fx["cfg"]
    keys = ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
            "num_key_value_heads", "rms_norm_eps", "rope_theta")
    p = fx["state"]
    emb = p["model.embed_tokens.weight"][fx["ids"]]
    cfg = _NS(**{k: c[k] for k in keys})
    _, logits = llama.lm_loss(llama.decoder_forward(p, cfg, emb, fx["pos"]), p["lm_head.weight"], fx["ids"])
    assert torch.allclose(logits, fx["logits"], atol=2e-5, rtol=1e-4)
    cfg_w = _NS(sliding_window=fx["sliding_window"], **{k: c[k] for k in keys})
    _, logits_w = llama.lm_loss(llama.decoder_forward(p, cfg_w, emb, fx["pos"]), p["lm_head.weight"], fx["ids"])
    assert torch.allclose(logits_w, fx["logits_window"], atol=2e-5, rtol=1e-4)
    assert not torch.allclose(logits_w, logits, atol=1e-4)
    hooks = {c["start"] + k * c["stride"]: k for k in range(2)}

    def hook(i, x):
        if i not in hooks:
            return x
        return llama.sva_hook(x, fx["sampler_state"], f"{hooks[i]}.", c["p0"], c["side"] ** 2, fx["ctx"], fx["kvs"],
                              fx["masks"])
    _, logits_h = llama.lm_loss(llama.decoder_forward(p, cfg, fx["embeds"], fx["pos"], None, hook), p["lm_head.weight"],
                                fx["ids"])
    assert torch.allclose(logits_h, fx["logits_hook"], atol=3e-5, rtol=1e-4)
