"""CPU: prompt-side helpers of cambrian/mm_utils.py (tokenizer_image_token*, get_model_name_from_path) against the
reference's own functions where /root/reference is present, and against literal expectations everywhere."""
import os
from types import SimpleNamespace

import pytest
import torch

from cambrian_amd import mm_utils as M

REF = "/root/reference/cambrian/mm_utils.py"


class _Tok:
    """word-level stand-in: BOS (= 1) in front of every call, as the Llama-2 / Vicuna tokenizers do"""
    bos_token_id = 1

    def __init__(self, bos=True):
        self.bos = bos

    def __call__(self, text):
        ids = [10 + (sum(map(ord, w)) % 50) for w in text.split()]
        return SimpleNamespace(input_ids=([1] if self.bos else []) + ids)


PROMPTS = ["<image>\nwhat is this ?", "look <image> and <image> then answer", "no image here", "<image>", "a <image>",
           "<image><image> twice"]


def test_literal_expectations():
    t = _Tok()
    ids = M.tokenizer_image_token("hello <image> world", t)
    assert ids[0] == 1 and ids.count(-200) == 1 and ids.count(1) == 1
    assert ids == [1] + t("hello").input_ids[1:] + [-200] + t("world").input_ids[1:]
    pt = M.tokenizer_image_token("hello <image> world", t, return_tensors="pt")
    assert pt.dtype == torch.long and pt.tolist() == ids
    with pytest.raises(ValueError):
        M.tokenizer_image_token("x", t, return_tensors="np")
    t3 = _Tok(bos=False)
    assert M.tokenizer_image_token_llama3("a <image> b", t3) == t3("a").input_ids + [-200] + t3("b").input_ids
    assert M.get_model_name_from_path("/x/y/cambrian-8b/") == "cambrian-8b"
    assert M.get_model_name_from_path("/x/cambrian-8b/checkpoint-500") == "cambrian-8b_checkpoint-500"


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present (GPU box)")
def test_same_as_reference_functions():
    src = open(REF).read().split("\n")
    ns = {"torch": torch, "IMAGE_TOKEN_INDEX": -200}
    exec("\n".join(src[202:249]), ns)       # tokenizer_image_token, tokenizer_image_token_llama3, get_model_name_from_path
    for bos in (True, False):
        t = _Tok(bos)
        for p in PROMPTS:
            assert M.tokenizer_image_token(p, t) == ns["tokenizer_image_token"](p, t), (bos, p)
            assert M.tokenizer_image_token_llama3(p, t) == ns["tokenizer_image_token_llama3"](p, t), (bos, p)
    for path in ("a/b/c", "/a/b/checkpoint-1/", "m"):
        assert M.get_model_name_from_path(path) == ns["get_model_name_from_path"](path)
