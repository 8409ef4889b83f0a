"""CPU: load_pretrained_model (model/builder.py:29-173) on synthetic local checkpoints — full checkpoint (single file and
sharded), base model + mm_projector.bin, token-embedding growth, tower / processor hand-off, refused branches."""
import json
import os

import pytest
import torch
import torch.nn as nn


class _Tok:
    def __init__(self, n):
        self.n, self.added = n, []

    def add_tokens(self, toks, special_tokens=False):
        self.added += toks
        self.n += len(toks)

    def __len__(self):
        return self.n


class _Tower(nn.Module):
    def __init__(self, hidden, tokens):
        super().__init__()
        self.hidden_size, self.tokens, self.is_loaded, self.image_processor = hidden, tokens, False, f"proc{hidden}"

    def load_model(self, device_map=None):
        self.is_loaded = True


def _make(monkeypatch, seed):
    import cambrian_amd.model.cambrian_arch as A
    from cambrian_amd.model.language_model import cambrian_llama as CL
    monkeypatch.setattr(A, "build_vision_tower_aux_list", lambda cfg, **kw: [_Tower(128, 16), _Tower(384, 64)])
    cfg = CL.CambrianConfig(vocab_size=70, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                            num_key_value_heads=2, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=64)
    CL.apply_release_8b_vision_config(cfg, towers=["a", "b"], token_lens=[16, 64])
    cfg.image_token_len, cfg.query_num_list, cfg.connector_depth = 16, [16], 2
    cfg.num_of_vision_sampler_layers, cfg.start_of_vision_sampler_layers, cfg.stride_of_vision_sampler_layers = 2, 0, 1
    cfg.vision_hidden_size = 64
    torch.manual_seed(seed)
    m = CL.CambrianLlamaForCausalLM(cfg, device="cpu", llm_dtype=torch.float32)
    with torch.no_grad():
        m.model.image_newline.copy_(torch.randn(64))
    return m, cfg


def _same(model, ref_state, n_old=70):
    sd = model.state_dict()
    for k, v in ref_state.items():
        got = sd[k][:n_old] if k in ("model.embed_tokens.weight", "lm_head.weight") else sd[k]
        assert torch.equal(got.float(), v.float()), k


def test_full_checkpoint_single_and_sharded(tmp_path, monkeypatch):
    from safetensors.torch import save_file
    from cambrian_amd.model.builder import load_pretrained_model
    src, cfg = _make(monkeypatch, 1)
    state = {k: v.detach().clone().contiguous() for k, v in src.state_dict().items()}
    d1 = tmp_path / "cambrian-8b-test"
    d1.mkdir()
    cfg.save_pretrained(str(d1))
    save_file(state, str(d1 / "model.safetensors"))
    tok = _Tok(70)
    tokenizer, model, procs, ctx = load_pretrained_model(str(d1), None, "cambrian-8b-test", device="cpu", tokenizer=tok,
                                                         torch_dtype=torch.float32)
    assert tokenizer is tok and tok.added == ["<im_patch>"] and ctx == 2048
    assert procs == ["proc128", "proc384"] and all(t.is_loaded for t in model.get_vision_tower_aux_list())
    assert model.model.embed_tokens.weight.shape == (71, 64) and model.lm_head.weight.shape == (71, 64)
    assert model.config.vocab_size == 71
    _same(model, state)
    # sharded + .bin + start/end tokens
    d2 = tmp_path / "cambrian-13b-test"
    d2.mkdir()
    cfg.mm_use_im_start_end = True
    cfg.save_pretrained(str(d2))
    keys = sorted(state)
    a, b = {k: state[k] for k in keys[::2]}, {k: state[k] for k in keys[1::2]}
    torch.save(a, str(d2 / "pytorch_model-00001-of-00002.bin"))
    torch.save(b, str(d2 / "pytorch_model-00002-of-00002.bin"))
    wm = {**{k: "pytorch_model-00001-of-00002.bin" for k in a}, **{k: "pytorch_model-00002-of-00002.bin" for k in b}}
    json.dump({"weight_map": wm}, open(d2 / "pytorch_model.bin.index.json", "w"))
    tok2 = _Tok(70)
    _, model2, _, _ = load_pretrained_model(str(d2), None, "cambrian-13b-test", device="cpu", tokenizer=tok2,
                                            torch_dtype=torch.float32)
    assert tok2.added == ["<im_patch>", "<im_start>", "<im_end>"] and model2.lm_head.weight.shape[0] == 73
    _same(model2, state)


def test_base_model_plus_adapter(tmp_path, monkeypatch):
    from safetensors.torch import save_file
    from cambrian_amd.model.builder import load_pretrained_model
    from cambrian_amd.train.checkpoint import save_mm_adapter
    src, cfg = _make(monkeypatch, 3)
    state = {k: v.detach().clone().contiguous() for k, v in src.state_dict().items()}
    base = tmp_path / "llama-base"
    base.mkdir()
    adapter_keys = ("mm_projector", "vision_sampler", "vision_query", "image_newline")
    save_file({k: v for k, v in state.items() if not any(a in k for a in adapter_keys)}, str(base / "model.safetensors"))
    ad = tmp_path / "cambrian-pretrain"
    save_mm_adapter(src, str(ad))
    assert os.path.isfile(ad / "config.json")
    _, model, _, _ = load_pretrained_model(str(ad), str(base), "cambrian-pretrain", device="cpu", tokenizer=_Tok(70),
                                           torch_dtype=torch.float32)
    _same(model, state)
    with pytest.raises(RuntimeError):          # an adapter file that lacks the connector
        torch.save({"model.image_newline": state["model.image_newline"]}, str(ad / "mm_projector.bin"))
        load_pretrained_model(str(ad), str(base), "cambrian-pretrain", device="cpu", tokenizer=_Tok(70), torch_dtype=torch.float32)


def test_refused_branches(tmp_path):
    from cambrian_amd.model.builder import load_hf_state, load_pretrained_model
    for name, kw in (("cambrian-8b", dict(load_8bit=True)), ("vicuna-7b", {}), ("cambrian-lora", {}), ("cambrian-mistral", {})):
        with pytest.raises(NotImplementedError):
            load_pretrained_model(str(tmp_path), None, name, device="cpu", **kw)
    with pytest.raises(FileNotFoundError):
        load_hf_state(str(tmp_path))


def test_missing_tower_weights_raise_without_the_opt_in(monkeypatch):
    """a tower with no local checkpoint must fail like the reference's from_pretrained does, not fall back to
    random weights silently; CAMBRIAN_AMD_RANDOM_INIT=1 is the explicit opt-in the tests / bench use."""
    from types import SimpleNamespace
    import pytest
    from cambrian_amd.model.multimodal_encoder.builder import build_vision_tower
    monkeypatch.setenv("CAMBRIAN_AMD_RANDOM_INIT", "0")
    monkeypatch.delenv("CAMBRIAN_WEIGHTS_DIR", raising=False)
    cfg = SimpleNamespace(mm_vision_tower="openai/clip-vit-large-patch14-336", mm_vision_select_layer=-2,
                          mm_vision_select_feature="patch", unfreeze_mm_vision_tower=False)
    tower = build_vision_tower(cfg, delay_load=True)
    with pytest.raises(FileNotFoundError, match="CAMBRIAN_AMD_RANDOM_INIT"):
        tower.load_model()
