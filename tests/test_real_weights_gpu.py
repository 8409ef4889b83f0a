"""`-m gpu`: real-weight loading.  A tower whose name resolves to a LOCAL checkpoint loads it through
the key maps of weight_maps.py instead of random-initialising: a small HF ``CLIPVisionModel`` / ``Dinov2Model`` is saved
with ``save_pretrained`` (model.safetensors), the native tower is pointed at that directory, and its output must equal
the HF module's own forward — ``hidden_states[-2]`` without CLS for CLIP (clip_encoder.py:57-68), ``last_hidden_state``
without CLS for DINOv2 (dino_encoder.py:115-126)."""
from types import SimpleNamespace

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _args():
    return SimpleNamespace(mm_vision_select_layer=-2, mm_vision_select_feature="patch", unfreeze_mm_vision_tower=False)


def test_clip_tower_loads_local_hf_checkpoint(dev, tmp_path, monkeypatch):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from cambrian_amd.model.multimodal_encoder import clip_encoder as CE
    torch.manual_seed(0)
    hf_cfg = CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                              image_size=56, patch_size=14, hidden_act="quick_gelu", projection_dim=64)
    hf = CLIPVisionModel(hf_cfg).eval()
    ckpt_dir = str(tmp_path / "clip-tiny")
    hf.save_pretrained(ckpt_dir)
    monkeypatch.setitem(CE.CLIP_ARCH, ckpt_dir, dict(image_size=56, patch_size=14, hidden_size=128, num_layers=3,
                                                      num_heads=2, mlp_dim=256))
    tower = CE.ClipVisionTower(ckpt_dir, _args(), delay_load=True)
    tower._compute_dtype = torch.float32
    tower.load_model()
    tower.vision_tower.to(dev)
    img = torch.randn(2, 3, 56, 56)
    with torch.no_grad():
        ref = hf(img, output_hidden_states=True).hidden_states[-2][:, 1:]
    out = tower(img.to(dev))
    assert out.shape == ref.shape == (2, 16, 128)
    assert rel_err(out, ref) < 1e-4


def test_dino_tower_loads_local_hf_checkpoint(dev, tmp_path, monkeypatch):
    """$CAMBRIAN_WEIGHTS_DIR/<org>--<repo>/model.safetensors, DINOv2-small architecture at its native 518 px (no
    position-embedding interpolation, which the reference pins to the transformers==4.37 arithmetic)."""
    import os
    from transformers import Dinov2Config, Dinov2Model
    from cambrian_amd.model.multimodal_encoder import dino_encoder as DE
    torch.manual_seed(1)
    hf_cfg = Dinov2Config(hidden_size=384, num_hidden_layers=12, num_attention_heads=6, mlp_ratio=4, image_size=518,
                          patch_size=14, use_swiglu_ffn=False, layerscale_value=0.7)
    hf = Dinov2Model(hf_cfg).eval()
    hf.save_pretrained(str(tmp_path / "facebook--dinov2-small"))
    monkeypatch.setenv("CAMBRIAN_WEIGHTS_DIR", str(tmp_path))
    tower = DE.DinoVisionTower("facebook/dinov2-small-res518", _args(), delay_load=True)
    tower._compute_dtype = torch.float32
    tower.load_model()
    tower.vision_tower.to(dev)
    img = torch.randn(1, 3, 518, 518)
    with torch.no_grad():
        ref = hf(img).last_hidden_state[:, 1:]
    out = tower(img.to(dev))
    assert out.shape == ref.shape == (1, 1369, 384)
    assert rel_err(out, ref) < 2e-4
