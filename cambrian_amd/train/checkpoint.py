"""Adapter checkpoints of the pre-training stage (``--tune_mm_mlp_adapter``): what ``safe_save_model_for_hf_trainer``
keeps (``train_fsdp.py:245-253``: parameters whose name contains one of ADAPTER_KEYS) and what
``initialize_vision_modules(pretrain_mm_mlp_adapter=...)`` / ``load_pretrained_model`` read back
(``cambrian_arch.py:183-200``, ``model/builder.py``: ``mm_projector.bin``).  The reference writes one XLA shard per TPU
core and consolidates them offline (``eval/scripts/convert_hf_model.py``); on one MI355X node every rank holds the full
adapter, so rank 0 writes the consolidated file directly.  Key names are the model's ``named_parameters`` names
(``model.mm_projector.0.weight``, ``model.vision_sampler_layers.3.layers.0…``, ``model.vision_query``, …) — the
reference's."""
from __future__ import annotations

import os
from typing import Dict

import torch

ADAPTER_KEYS = ("mm_projector", "pos_emb", "vision_sampler", "vision_sampler_layers", "vision_query", "image_newline")


def mm_adapter_state(model: torch.nn.Module, use_im_start_end: bool = False) -> Dict[str, torch.Tensor]:
    """train_fsdp.py:247-252 (get_mm_adapter_state_maybe_zero_3 without the DeepSpeed gather): CPU copies."""
    keys = ADAPTER_KEYS + (("embed_tokens", "embed_in") if use_im_start_end else ())
    return {n: p.detach().cpu().clone() for n, p in model.named_parameters() if any(k in n for k in keys)}


def save_mm_adapter(model: torch.nn.Module, output_dir: str, use_im_start_end: bool = False, rank: int = 0) -> str:
    """Writes ``<output_dir>/mm_projector.bin`` (rank 0 only) and returns its path."""
    path = os.path.join(output_dir, "mm_projector.bin")
    if rank == 0:
        os.makedirs(output_dir, exist_ok=True)
        torch.save(mm_adapter_state(model, use_im_start_end), path)
        cfg = getattr(model, "config", None)
        if cfg is not None and hasattr(cfg, "save_pretrained"):
            try:
                cfg.save_pretrained(output_dir)                 # train_fsdp.py:254-255
            except Exception:                                   # configs carrying non-JSON test doubles
                pass
    return path
