"""CPU, build container only (skipped where /root/reference is absent — e.g. on the GPU box): the drop-in surface of
SURVEY.md §8(b), checked MECHANICALLY against the reference's own source.  The reference files are AST-parsed (never
imported through their package, whose __init__ chain needs timm / open_clip) and compared with ``inspect.signature`` of the
package's classes; the one reference module that imports cleanly (vision_sampler.py) is also instantiated and its
``state_dict`` keys / shapes compared with the package's module."""
import ast
import importlib.util
import inspect
import os

import pytest

REF = "/root/reference/cambrian"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference is not present on this machine")


def _classes(path):
    with open(path) as f:
        tree = ast.parse(f.read())
    out = {}
    for node in tree.body:
        if isinstance(node, ast.ClassDef):
            out[node.name] = {n.name: n for n in node.body if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef))}
    funcs = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
    return out, funcs


def _ref_args(fn: ast.FunctionDef):
    """(positional names incl. self, names that have defaults, *args name, **kwargs name, is_property)"""
    a = fn.args
    pos = [x.arg for x in a.posonlyargs + a.args]
    n_def = len(a.defaults)
    with_default = set(pos[len(pos) - n_def:]) if n_def else set()
    is_prop = any((isinstance(d, ast.Name) and d.id == "property") for d in fn.decorator_list)
    return pos, with_default, (a.vararg.arg if a.vararg else None), (a.kwarg.arg if a.kwarg else None), is_prop


def _check_callable(ref_fn, ours, what):
    pos, with_default, vararg, kwarg, _ = _ref_args(ref_fn)
    sig = inspect.signature(ours)
    ours_pos = [p for p in sig.parameters.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    names = [p.name for p in ours_pos]
    # every reference parameter exists here, in the same order (extra trailing parameters with defaults are allowed)
    assert names[:len(pos)] == pos, f"{what}: reference parameters {pos} vs {names}"
    for p in ours_pos[len(pos):]:
        assert p.default is not inspect.Parameter.empty, f"{what}: extra parameter {p.name} has no default"
    for p in ours_pos[:len(pos)]:
        if p.name in with_default:
            assert p.default is not inspect.Parameter.empty, f"{what}: {p.name} has a default in the reference"
    if vararg:
        assert any(p.kind == p.VAR_POSITIONAL for p in sig.parameters.values()), f"{what}: reference takes *{vararg}"
    if kwarg:
        assert any(p.kind == p.VAR_KEYWORD for p in sig.parameters.values()), f"{what}: reference takes **{kwarg}"


def test_vision_sampler_signatures():
    from cambrian_amd.model import vision_sampler as ours
    cls, _ = _classes(f"{REF}/model/vision_sampler.py")
    _check_callable(cls["VisionTokenSampler"]["__init__"], ours.VisionTokenSampler.__init__, "VisionTokenSampler.__init__")
    _check_callable(cls["VisionTokenSampler"]["forward"], ours.VisionTokenSampler.forward, "VisionTokenSampler.forward")


def test_vision_sampler_state_dict_equals_the_reference_modules():
    spec = importlib.util.spec_from_file_location("ref_vs_conf", f"{REF}/model/vision_sampler.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    from cambrian_amd.model.vision_sampler import VisionTokenSampler
    for q_dim, sizes, layers in ((1024, [1, 1, 1, 4], 3), (4096, [1, 1, 2, 4], 1), (3072, [1], 2)):
        kv_dims = [1024] * len(sizes)
        r = ref.VisionTokenSampler(q_dim, 1024, kv_dims, sizes, 1024, layers)
        o = VisionTokenSampler(q_dim, 1024, kv_dims, sizes, 1024, layers)
        rs, os_ = r.state_dict(), o.state_dict()
        assert list(rs.keys()) == list(os_.keys()), set(rs) ^ set(os_)
        for k in rs:
            assert rs[k].shape == os_[k].shape and rs[k].dtype == os_[k].dtype, k
        o.load_state_dict(rs, strict=True)     # a reference checkpoint loads
        r.load_state_dict(o.state_dict(), strict=True)   # and the other way round


ARCH_METHODS = ["encode_images", "rearrange_vision_tower_features_train", "rearrange_vision_tower_features_inference",
                "prepare_inputs_labels_for_multimodal", "initialize_vision_tokenizer", "get_vision_tower_aux_list"]


def test_cambrian_arch_surface():
    from cambrian_amd.model import cambrian_arch as ours
    cls, funcs = _classes(f"{REF}/model/cambrian_arch.py")
    meta, causal = cls["CambrianMetaModel"], cls["CambrianMetaForCausalLM"]
    _check_callable(meta["__init__"], ours.CambrianMetaModel.__init__, "CambrianMetaModel.__init__")
    for name in ("get_vision_tower_aux_list", "initialize_vision_modules"):
        _check_callable(meta[name], getattr(ours.CambrianMetaModel, name), f"CambrianMetaModel.{name}")
    for name in ARCH_METHODS:
        if name in causal:
            _check_callable(causal[name], getattr(ours.CambrianMetaForCausalLM, name), f"CambrianMetaForCausalLM.{name}")
    assert "get_model" in causal and hasattr(ours.CambrianMetaForCausalLM, "get_model")
    for fn in ("unmask_attention_mask", "unpad_image"):
        if fn in funcs:
            _check_callable(funcs[fn], getattr(ours, fn), fn)
    # the 10-tuple: count the elements of the static branch's return statement in the reference
    src = open(f"{REF}/model/cambrian_arch.py").read()
    tree = ast.parse(src)
    lens = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "prepare_inputs_labels_for_multimodal":
            for r in ast.walk(node):
                if isinstance(r, ast.Return) and isinstance(r.value, ast.Tuple):
                    lens.add(len(r.value.elts))
    assert lens == {10}, lens
    ours_src = inspect.getsource(ours.CambrianMetaForCausalLM.prepare_inputs_labels_for_multimodal)
    ours_lens = {len(r.value.elts) for r in ast.walk(ast.parse("class _X:\n" + ours_src if ours_src.startswith("    ") else ours_src))
                 if isinstance(r, ast.Return) and isinstance(r.value, ast.Tuple)}
    assert ours_lens <= {10} and ours_lens, ours_lens


def test_tower_protocol_and_builders():
    from cambrian_amd.model.multimodal_encoder import base_encoder as ours_base
    from cambrian_amd.model.multimodal_encoder import builder as ours_builder
    cls, _ = _classes(f"{REF}/model/multimodal_encoder/base_encoder.py")
    ref_base = cls["BaseVisionTower"]
    for name, fn in ref_base.items():
        if name.startswith("__") and name != "__init__":
            continue
        assert hasattr(ours_base.BaseVisionTower, name), f"BaseVisionTower.{name} missing"
        _, _, _, _, is_prop = _ref_args(fn)
        attr = inspect.getattr_static(ours_base.BaseVisionTower, name)
        if is_prop:
            assert isinstance(attr, property), f"BaseVisionTower.{name} is a property in the reference"
        elif name != "__init__":
            _check_callable(fn, attr, f"BaseVisionTower.{name}")
    _, funcs = _classes(f"{REF}/model/multimodal_encoder/builder.py")
    for name in ("build_vision_tower", "build_vision_tower_aux_list"):
        _check_callable(funcs[name], getattr(ours_builder, name), name)
    # the four release wrappers keep their class names and load_model(device_map=None)
    for mod, cname in (("clip_encoder", "ClipVisionTower"), ("siglip_encoder", "SiglipVisionTower"),
                       ("dino_encoder", "DinoVisionTower"), ("clip_convnext_encoder", "CLIPConvNextTower")):
        rcls, _ = _classes(f"{REF}/model/multimodal_encoder/{mod}.py")
        assert cname in rcls, (mod, list(rcls))
        ours_mod = importlib.import_module(f"cambrian_amd.model.multimodal_encoder.{mod}")
        ocls = getattr(ours_mod, cname)
        _check_callable(rcls[cname]["__init__"], ocls.__init__, f"{cname}.__init__")
        if "load_model" in rcls[cname]:
            _check_callable(rcls[cname]["load_model"], ocls.load_model, f"{cname}.load_model")


def test_projector_builder_and_language_model_surface():
    from cambrian_amd.model.multimodal_projector import builder as ours_proj
    _, funcs = _classes(f"{REF}/model/multimodal_projector/builder.py")
    _check_callable(funcs["build_vision_projector"], ours_proj.build_vision_projector, "build_vision_projector")
    from cambrian_amd.model.language_model import cambrian_llama as ours_llama
    cls, _ = _classes(f"{REF}/model/language_model/cambrian_llama.py")
    for cname in ("CambrianConfig", "CambrianLlamaModel", "CambrianLlamaForCausalLM"):
        assert hasattr(ours_llama, cname), cname
    ref_fwd = cls["CambrianLlamaForCausalLM"]["forward"]
    pos, *_ = _ref_args(ref_fwd)
    ours_params = set(inspect.signature(ours_llama.CambrianLlamaForCausalLM.forward).parameters)
    # every keyword the HF Trainer / the collator passes (train_fsdp.py:1177-1236) is accepted by name
    for name in ("input_ids", "attention_mask", "position_ids", "labels", "images", "image_aux_attention_masks_list", "image_sizes"):
        assert name in pos, name
        assert name in ours_params or any(p.kind == p.VAR_KEYWORD for p in
                                          inspect.signature(ours_llama.CambrianLlamaForCausalLM.forward).parameters.values()), name
    pos_g, _, _, kw_g, _ = _ref_args(cls["CambrianLlamaForCausalLM"]["generate"])
    ours_g = inspect.signature(ours_llama.CambrianLlamaForCausalLM.generate)
    assert list(ours_g.parameters)[:len(pos_g)] == pos_g, (pos_g, list(ours_g.parameters))
    assert kw_g and any(p.kind == p.VAR_KEYWORD for p in ours_g.parameters.values())
