"""``cambrian/model/builder.py`` — ``load_pretrained_model`` for the Cambrian checkpoints, same
signature and return value (tokenizer, model, image_processor list, context length).

Supported, as in the reference (:52-127): a full Cambrian checkpoint directory (Llama wrapper; the Phi-3 wrapper when
'phi3' is in the model name) and ``model_base`` + ``mm_projector.bin`` (adapter-only checkpoints of the pre-training
stage).  The LoRA / 8-bit / 4-bit / Mistral / plain-LM branches are outside the hot path and raise.
Differences that follow from the MI355X path: weights are loaded from LOCAL files only (``model.safetensors`` or its
shard index, else ``pytorch_model.bin`` / index — there is no hub access here), the model is built directly on
``device`` (``device_map`` other than a single device is ignored), and the compute dtype is bf16 (the kernels' storage
type) where the reference asks for fp16.
"""
from __future__ import annotations

import json
import logging
import os
from typing import Dict, Optional

import torch

from ..constants import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN

logger = logging.getLogger("cambrian_amd")


def load_hf_state(path: str) -> Dict[str, torch.Tensor]:
    """All tensors of a HF checkpoint directory (safetensors / .bin, single file or sharded through the index json)."""
    def read(fn):
        full = os.path.join(path, fn)
        if fn.endswith(".safetensors"):
            from safetensors.torch import load_file
            return load_file(full)
        return torch.load(full, map_location="cpu", weights_only=True)

    for single, index in (("model.safetensors", "model.safetensors.index.json"),
                          ("pytorch_model.bin", "pytorch_model.bin.index.json")):
        if os.path.isfile(os.path.join(path, single)):
            return read(single)
        if os.path.isfile(os.path.join(path, index)):
            shards = sorted(set(json.load(open(os.path.join(path, index)))["weight_map"].values()))
            sd: Dict[str, torch.Tensor] = {}
            for shard in shards:
                sd.update(read(shard))
            return sd
    raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin (or shard index) under {path}")


def _wrapper_for(model_name: str):
    if "phi3" in model_name.lower():
        from .language_model.cambrian_phi3 import CambrianConfig, CambrianPhi3ForCausalLM
        return CambrianConfig, CambrianPhi3ForCausalLM
    from .language_model.cambrian_llama import CambrianConfig, CambrianLlamaForCausalLM
    return CambrianConfig, CambrianLlamaForCausalLM


def _load_into(model, sd: Dict[str, torch.Tensor], what: str, allow_missing=()) -> None:
    res = model.load_state_dict(sd, strict=False)
    missing = [k for k in res.missing_keys if not any(a in k for a in allow_missing)]
    if missing:
        raise RuntimeError(f"{what}: {len(missing)} parameters missing from the checkpoint, e.g. {missing[:4]}")
    if res.unexpected_keys:
        logger.warning(f"{what}: {len(res.unexpected_keys)} checkpoint tensors not used, e.g. {res.unexpected_keys[:4]}")


def load_pretrained_model(model_path, model_base, model_name, load_8bit=False, load_4bit=False, device_map="auto",
                          device="cuda", use_flash_attn=False, tokenizer=None, **kwargs):
    """model/builder.py:29-173.  ``tokenizer``: pass one to skip ``AutoTokenizer.from_pretrained`` (offline tests)."""
    if load_8bit or load_4bit:
        raise NotImplementedError("bitsandbytes quantised loading is outside the MI355X hot path")
    if "cambrian" not in model_name.lower():
        raise NotImplementedError("plain language models load with transformers; this loader is for Cambrian checkpoints")
    if "lora" in model_name.lower():
        raise NotImplementedError("LoRA checkpoints need peft merging (model/builder.py:54-92); merge first, then load")
    if "mistral" in model_name.lower():
        raise NotImplementedError("the Mistral wrapper is stale in the reference (SURVEY.md §2) and not provided")
    dev = torch.device(device if device != "cuda" else f"cuda:{torch.cuda.current_device()}") if device != "cpu" else torch.device("cpu")
    dtype = kwargs.pop("torch_dtype", torch.bfloat16)
    if dtype == torch.float16:
        dtype = torch.bfloat16
    config_cls, model_cls = _wrapper_for(model_name)
    cfg = config_cls.from_pretrained(model_path)
    if tokenizer is None:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(model_base or model_path, use_fast="phi3" in model_name.lower())
    model = model_cls(cfg, device=dev, llm_dtype=dtype)
    adapter = ("mm_projector", "vision_sampler", "vision_query", "image_newline")
    if model_base is not None:
        # adapter-only checkpoint (:93-102): decoder from the base model, connector from mm_projector.bin
        logger.info(f"Loading Cambrian-1 from base model... {model_base}")
        _load_into(model, load_hf_state(model_base), "base model", allow_missing=adapter)
        weights = torch.load(os.path.join(model_path, "mm_projector.bin"), map_location="cpu", weights_only=True)
        res = model.load_state_dict(weights, strict=False)
        if res.unexpected_keys:
            raise RuntimeError(f"mm_projector.bin: unknown tensors {res.unexpected_keys[:4]}")
        still = [k for k in res.missing_keys if any(a in k for a in adapter)]
        if still:
            raise RuntimeError(f"mm_projector.bin: adapter tensors missing, e.g. {still[:4]}")
    else:
        logger.info(f"Loading Cambrian from {model_path}")
        _load_into(model, load_hf_state(model_path), "checkpoint")
    model = model.to(dev)
    # :151-158
    if getattr(model.config, "mm_use_im_patch_token", True):
        tokenizer.add_tokens([DEFAULT_IMAGE_PATCH_TOKEN], special_tokens=True)
    if getattr(model.config, "mm_use_im_start_end", False):
        tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
    model.resize_token_embeddings(len(tokenizer))
    towers = model.get_vision_tower_aux_list()           # :160-167 (the towers place themselves on the current device)
    for t in towers:
        if not t.is_loaded:
            t.load_model(device_map=device_map)
    image_processor = [t.image_processor for t in towers]
    context_len = getattr(model.config, "max_sequence_length", 2048)      # :169-172
    return tokenizer, model, image_processor, context_len
