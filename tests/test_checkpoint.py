"""CPU: adapter checkpoint round trip (train_fsdp.py:245-253 key filter -> mm_projector.bin ->
initialize_vision_modules(pretrain_mm_mlp_adapter=...), cambrian_arch.py:183-200)."""
from types import SimpleNamespace

import torch
import torch.nn as nn


class _Tower(nn.Module):
    def __init__(self, hidden, tokens):
        super().__init__()
        self.hidden_size, self.tokens, self.is_loaded = hidden, tokens, True

    def load_model(self, device_map=None):
        pass


def _model(monkeypatch, seed):
    import cambrian_amd.model.cambrian_arch as A
    from cambrian_amd.model.language_model import cambrian_llama as CL
    towers = [_Tower(128, 16), _Tower(384, 64)]
    monkeypatch.setattr(A, "build_vision_tower_aux_list", lambda cfg, **kw: towers)
    cfg = CL.CambrianConfig(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
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


def test_adapter_round_trip(tmp_path, monkeypatch):
    from cambrian_amd.train.checkpoint import ADAPTER_KEYS, mm_adapter_state, save_mm_adapter
    src, cfg = _model(monkeypatch, 1)
    state = mm_adapter_state(src)
    names = {n for n, _ in src.named_parameters()}
    assert state and all(any(k in n for k in ADAPTER_KEYS) for n in state)
    # exactly the pre-training stage's trainable set (train_fsdp.py:1681): nothing of the decoder, everything of the connector
    assert not any(n.startswith(("model.layers.", "lm_head", "model.embed_tokens", "model.norm")) for n in state)
    assert {n for n in names if "vision_sampler" in n or "mm_projector" in n} <= set(state)
    assert "model.vision_query" in state and "model.image_newline" in state
    assert any("pos_embed_1" in n for n in state)
    path = save_mm_adapter(src, str(tmp_path / "ckpt"))
    assert path.endswith("mm_projector.bin")
    dst, _ = _model(monkeypatch, 2)                      # different random init
    assert not torch.equal(dst.model.vision_query, src.model.vision_query)
    args = SimpleNamespace(image_token_len=16, num_query_group=1, query_num_list=[16], connector_depth=2,
                           vision_tower_aux_list=["a", "b"], vision_tower_aux_token_len_list=[16, 64], connector_only=False,
                           unfreeze_mm_vision_tower=False, mm_projector_type="sva", vision_hidden_size=64,
                           mm_vision_select_layer=-2, mm_vision_select_feature="patch", num_of_vision_sampler_layers=2,
                           start_of_vision_sampler_layers=0, stride_of_vision_sampler_layers=1,
                           pretrain_mm_mlp_adapter=path)
    dst.model.initialize_vision_modules(args)
    after = mm_adapter_state(dst)
    assert after.keys() == state.keys()
    for k in state:
        assert torch.equal(after[k], state[k]), k
    # the decoder was not touched
    assert not torch.equal(dst.lm_head.weight, src.lm_head.weight)
    assert set(mm_adapter_state(src, use_im_start_end=True)) - set(state) == {"model.embed_tokens.weight"}
