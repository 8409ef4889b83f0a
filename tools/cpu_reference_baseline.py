"""CPU baseline from the REFERENCE'S OWN modules (SURVEY.md §8d "CPU baseline plan").

Runs only where /root/reference exists (the build container; the GPU box has no reference).  Times, on this host's
cores, fp32, B = 1, at the release-8B dimensions:

  * the real `cambrian/model/vision_sampler.py::VisionTokenSampler` — the 3-layer connector (q_dim 1024) and ten
    1-layer in-LLM samplers (q_dim 4096) over one image's 10 944 KV tokens, forward + backward;
  * the real `cambrian_arch.py::prepare_inputs_labels_for_multimodal` (static branch, loader shim of
    tests/golden/make_golden.py) with stand-in towers that return seeded features: aux projectors + connector +
    mm_projector + newline/splice, forward + backward (its SVA part is the first bullet's connector, so only the
    NON-sampler remainder is added to the total);
  * the third-party tower arithmetic the reference delegates to, from the installed HF classes with random weights at
    the release dimensions (CLIPVisionModel L/14@336, SiglipVisionModel SO400M/14@384, Dinov2Model giant@378,
    ConvNextModel XXL widths @1024), forward only (frozen, no_grad — as in the reference run).

Writes profiles/<round>_cpu_reference_baseline.json (dated; round 5 re-timed it: r05_cpu_reference_baseline.json); bench.py reports it as cpu_baseline.reference_run.
"""
from __future__ import annotations

import json
import os
import platform
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as G  # noqa: E402  (loader shims only)


def timed(fn, warm=1, reps=2):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def main():
    cores = int(os.environ.get("CPU_BASELINE_THREADS", os.cpu_count() or 1))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    out = {"host": platform.processor() or platform.machine(), "cores": cores, "dtype": "fp32", "batch": 1,
           "torch": torch.__version__, "parts": {}}
    hidden, kv_sizes, qside = 1024, [1, 1, 1, 4], 24
    bq = qside * qside
    vs = G.load_ref_vision_sampler()

    def sva_case(q_dim, layers):
        m = vs.VisionTokenSampler(q_dim, hidden, [hidden] * 4, kv_sizes, hidden, layers).float()
        q = torch.randn(bq, 1, q_dim, requires_grad=True)
        ctx = torch.randn(bq, 1, hidden)
        kvs = [torch.randn(bq, s * s, hidden, requires_grad=True) for s in kv_sizes]
        masks = [torch.ones(bq, s * s, dtype=torch.bool) for s in kv_sizes]

        def run():
            m.zero_grad(set_to_none=True)
            m(q, ctx, *kvs, *masks).sum().backward()
        return timed(run)

    t_conn = sva_case(1024, 3)
    t_llm1 = sva_case(4096, 1)
    out["parts"]["sva_connector_3_layers_fwd_bwd_s"] = t_conn
    out["parts"]["sva_in_llm_1_layer_fwd_bwd_s"] = t_llm1

    # prepare_inputs_labels_for_multimodal (real cambrian_arch.py), release geometry, stand-in towers
    import torch.nn as nn
    A = G.load_ref_arch()
    ns = G.load_ref_collator()
    H, S, V = 4096, 2048, 1024        # a small vocabulary: the embedding table is not part of the hot path
    tower_dims, token_lens = [1152, 1024, 1536, 5760], [576, 576, 576, 9216]

    class FakeTower(nn.Module):
        def __init__(self, hid, tok):
            super().__init__()
            self.hidden_size, self.tokens, self.is_loaded, self.out = hid, tok, True, None

        def load_model(self):
            pass

        def forward(self, images):
            return self.out

    towers = [FakeTower(d, t) for d, t in zip(tower_dims, token_lens)]

    class Cfg:
        pass

    cfg = Cfg()
    cfg.hidden_size, cfg.vision_hidden_size = H, hidden
    cfg.mm_vision_tower_aux_list = ["a", "b", "c", "d"]
    cfg.mm_vision_tower_aux_token_len_list = token_lens
    cfg.mm_projector_type = "sva"
    cfg.num_query_group, cfg.query_num_list, cfg.connector_only, cfg.connector_depth = 1, [576], False, 3
    cfg.image_token_len = 576
    cfg.num_of_vision_sampler_layers, cfg.start_of_vision_sampler_layers, cfg.stride_of_vision_sampler_layers = 10, 0, 3
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
    ids = torch.randint(1, V, (1, S))
    ids[0, 91] = -200
    labels = ids.clone()
    att = torch.ones(1, S, dtype=torch.bool)
    new_ids, new_lab, new_att, new_pos, aux_masks = ns["prepare_multimodal_data"](ids, labels, att, [(336, 336)], 576,
                                                                               token_lens, S)
    feats = [torch.randn(1, t, d) for t, d in zip(token_lens, tower_dims)]
    for t, f in zip(towers, feats):
        t.out = f
    images = [torch.zeros(1, 3, 8, 8) for _ in towers]

    def run_prepare():
        lm.zero_grad(set_to_none=True)
        o = lm.prepare_inputs_labels_for_multimodal(new_ids, new_pos, new_att, None, new_lab, images, aux_masks, [(336, 336)])
        o[4].sum().backward()
    t_prep = timed(run_prepare)
    out["parts"]["prepare_inputs_labels_for_multimodal_fwd_bwd_s"] = t_prep
    out["parts"]["aux_projectors_mm_projector_splice_fwd_bwd_s"] = max(t_prep - t_conn, 0.0)

    # towers: installed HF classes (stand-ins for transformers 4.37 / timm), random weights, release dims
    from transformers import (CLIPVisionConfig, CLIPVisionModel, ConvNextConfig, ConvNextModel, Dinov2Config, Dinov2Model,
                              SiglipVisionConfig, SiglipVisionModel)
    tower_s = {}
    with torch.no_grad():
        m = CLIPVisionModel(CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                                             num_attention_heads=16, image_size=336, patch_size=14)).eval()
        x = torch.randn(1, 3, 336, 336)
        tower_s["clip_l_14_336"] = timed(lambda: m(x, output_hidden_states=True), reps=1)
        m = SiglipVisionModel(SiglipVisionConfig(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27,
                                                 num_attention_heads=16, image_size=384, patch_size=14)).eval()
        x = torch.randn(1, 3, 384, 384)
        tower_s["siglip_so400m_14_384"] = timed(lambda: m(x), reps=1)
        m = Dinov2Model(Dinov2Config(hidden_size=1536, num_hidden_layers=40, num_attention_heads=24, image_size=518,
                                     patch_size=14, use_swiglu_ffn=True, mlp_ratio=4)).eval()
        x = torch.randn(1, 3, 378, 378)
        tower_s["dinov2_giant_378"] = timed(lambda: m(x), reps=1)
        m = ConvNextModel(ConvNextConfig(depths=[3, 4, 30, 3], hidden_sizes=[384, 768, 1536, 3072])).eval()
        x = torch.randn(1, 3, 1024, 1024)
        tower_s["convnext_xxl_1024"] = timed(lambda: m(x, output_hidden_states=True), warm=0, reps=1)
    out["parts"]["towers_fwd_s"] = tower_s

    step_s = sum(tower_s.values()) + t_prep + 10 * t_llm1
    out["seconds_per_image_tower_plus_sva_train_step"] = step_s
    out["images_per_s"] = 1.0 / step_s
    out["what"] = ("towers forward (HF stand-ins) + real prepare_inputs_labels_for_multimodal fwd+bwd (aux projectors, "
                   "3-layer connector, mm_projector, splice) + 10 x real in-LLM VisionTokenSampler layer fwd+bwd; the LLM "
                   "decoder itself is not part of the tower+SVA path")
    out["timed_on"] = time.strftime("%Y-%m-%d %H:%M:%S %Z")
    out["round"] = os.environ.get("CAMBRIAN_ROUND", "r05")
    path = os.path.join(ROOT, "profiles", f"{out['round']}_cpu_reference_baseline.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if not os.path.isdir("/root/reference"):
        raise SystemExit("needs /root/reference (build container only)")
    main()
