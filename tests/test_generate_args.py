"""CPU: the decoding arguments the reference's eval scripts pass to ``model.generate`` (eval/eval/gqa/gqa_eval.py:108-117:
do_sample, temperature, top_p, num_beams, max_new_tokens, use_cache) are either implemented as HF GenerationMixin defines
them or rejected loudly — never swallowed (VERDICT r4 missing #5)."""
import pytest
import torch

from cambrian_amd.model.language_model.cambrian_llama import _generate, filter_logits_top_k_top_p


@pytest.mark.parametrize("top_k,top_p", [(None, 0.9), (None, 0.5), (50, None), (20, 0.7), (None, 1.0), (0, 0.05)])
def test_top_k_top_p_equal_hf_warpers(top_k, top_p):
    tf = pytest.importorskip("transformers")
    from transformers.generation.logits_process import TopKLogitsWarper, TopPLogitsWarper
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(5, 1000, generator=g) * 3.0
    ref = logits.clone()
    ids = torch.zeros(5, 1, dtype=torch.long)
    if top_k is not None and top_k > 0:
        ref = TopKLogitsWarper(top_k=top_k)(ids, ref)
    if top_p is not None and top_p < 1.0:
        ref = TopPLogitsWarper(top_p=top_p)(ids, ref)
    out = filter_logits_top_k_top_p(logits.clone(), top_k, top_p)
    assert torch.equal(torch.isinf(out), torch.isinf(ref))
    keep = ~torch.isinf(ref)
    assert torch.equal(out[keep], ref[keep])
    assert bool(keep.any(-1).all())            # the most probable token always survives


def test_top_p_keeps_smallest_sufficient_set():
    logits = torch.log(torch.tensor([[0.5, 0.3, 0.15, 0.05]]))
    out = filter_logits_top_k_top_p(logits, None, 0.8)
    assert torch.isinf(out).tolist() == [[False, False, True, True]]
    out = filter_logits_top_k_top_p(logits, None, 0.81)
    assert torch.isinf(out).tolist() == [[False, False, False, True]]


def test_generate_rejects_what_it_does_not_implement():
    with pytest.raises(NotImplementedError, match="num_beams"):
        _generate(None, torch.zeros(1, 4, dtype=torch.long), num_beams=5)
    with pytest.raises(NotImplementedError, match="inputs_embeds"):
        _generate(None, None, inputs_embeds=torch.zeros(1, 4, 8))
    with pytest.raises(ValueError, match="penalty_alpha"):
        _generate(None, torch.zeros(1, 4, dtype=torch.long), penalty_alpha=0.6)
    with pytest.raises(ValueError, match="repetition_penalty"):
        _generate(None, torch.zeros(1, 4, dtype=torch.long), repetition_penalty=1.3)
    with pytest.raises(ValueError, match="top_p"):
        filter_logits_top_k_top_p(torch.zeros(1, 8), None, 0.0)
