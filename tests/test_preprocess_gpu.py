"""GPU: the image pre-processing kernels (csrc/preprocess.hip) through the C-ABI against the oracle, the committed
reference fixtures and Pillow itself — bit-exact (integer resample; table-exact float stage)."""
import os

import numpy as np
import pytest
import torch

from cambrian_amd import lib as L
from cambrian_amd.train import image_pipeline as IP
from oracle import preprocess as OP

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "preprocess_cases.npz")
OPENAI_MEAN, OPENAI_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)

RELEASE_SPECS = [IP.TowerPixelSpec(384, OPENAI_MEAN, (0.5,) * 3, (0.5,) * 3, "torchvision"),      # SigLIP-SO400M
                 IP.TowerPixelSpec(336, OPENAI_MEAN, OPENAI_MEAN, OPENAI_STD, "hf"),               # CLIP-L
                 IP.TowerPixelSpec(378, IMAGENET_MEAN, IMAGENET_MEAN, IMAGENET_STD, "hf"),         # DINOv2-g
                 IP.TowerPixelSpec(1024, OPENAI_MEAN, OPENAI_MEAN, OPENAI_STD, "torchvision")]     # ConvNeXt-XXL


def _pil_levels(img, R, image_mean):
    """the reference expression (train_fsdp.py:1006) with Pillow -> uint8 [R,R,3]"""
    from PIL import Image
    pil = Image.fromarray(img)
    w, h = pil.size
    if w != h:
        s = max(w, h)
        sq = Image.new("RGB", (s, s), tuple(int(x * 255) for x in image_mean))
        sq.paste(pil, (0, (w - h) // 2) if w > h else ((h - w) // 2, 0))
        pil = sq
    return np.asarray(pil.resize((R, R)))


def _want(img, sp, dtype):
    u8 = _pil_levels(img, sp.out_side, sp.pad_mean)
    lut = OP.pixel_lut(sp.mean, sp.std, sp.flavour)
    px = np.stack([lut[c][u8[:, :, c]] for c in range(3)])
    return torch.from_numpy(px).to(dtype)


def _equal(a, b):
    return torch.equal(a.cpu().view(torch.uint8), b.contiguous().view(torch.uint8))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_fixtures_bit_exact(dtype):
    g = np.load(GOLDEN)
    specs = [IP.TowerPixelSpec(int(s), tuple(p), tuple(m), tuple(d), str(f)) for s, f, p, m, d in
             zip(g["tower_sides"], g["tower_flavours"], g["tower_pad_mean"], g["tower_mean"], g["tower_std"])]
    images = [g[f"img{ci}"] for ci in range(int(g["n_cases"]))]
    outs = IP.GpuImagePreprocessor(specs, "cuda:0", dtype)(images)
    torch.cuda.synchronize()
    for ti, (name, sp) in enumerate(zip(g["tower_names"], specs)):
        lut = OP.pixel_lut(sp.mean, sp.std, sp.flavour)
        for ci in range(len(images)):
            u8 = g[f"u8_{ci}_{name}"]
            want = torch.from_numpy(np.stack([lut[c][u8[:, :, c]] for c in range(3)]))
            if ci < 2:
                assert torch.equal(want, torch.from_numpy(g[f"px_{ci}_{name}"]))
            assert _equal(outs[ti][ci], want.to(dtype)), (ci, str(name))


def test_release_towers_ragged_batch_vs_pillow():
    rng = np.random.default_rng(11)
    shapes = [(480, 640), (640, 480), (336, 336), (1024, 1024), (800, 1333), (50, 50), (1024, 700), (377, 378),
              (1, 1), (2000, 3)]
    images = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    pre = IP.GpuImagePreprocessor(RELEASE_SPECS, "cuda:0", torch.bfloat16)
    outs = pre(images)
    torch.cuda.synchronize()
    for ti, sp in enumerate(RELEASE_SPECS):
        assert outs[ti].shape == (len(images), 3, sp.out_side, sp.out_side)
        for bi, im in enumerate(images):
            assert _equal(outs[ti][bi], _want(im, sp, torch.bfloat16)), (ti, shapes[bi])


def test_large_downscale_and_constant_image():
    rng = np.random.default_rng(3)
    big = rng.integers(0, 256, (3000, 4000, 3), dtype=np.uint8)
    flat = np.full((700, 500, 3), (13, 200, 77), np.uint8)
    pre = IP.GpuImagePreprocessor(RELEASE_SPECS[:2], "cuda:0", torch.float32)
    outs = pre([big, flat])
    torch.cuda.synchronize()
    for ti, sp in enumerate(RELEASE_SPECS[:2]):
        assert _equal(outs[ti][0], _want(big, sp, torch.float32))
        assert _equal(outs[ti][1], _want(flat, sp, torch.float32))
    # a constant image letter-boxed with its own colour stays constant (coefficients sum to 2^22 +- rounding)
    sp = IP.TowerPixelSpec(336, (13 / 255 + 1e-4, 200 / 255 + 1e-4, 77 / 255 + 1e-4), (0.0,) * 3, (1.0,) * 3, "hf")
    o = IP.GpuImagePreprocessor([sp], "cuda:0", torch.float32)([flat])[0][0]
    lut = sp.lut()
    for c, lv in enumerate((13, 200, 77)):
        assert (o[c] == float(lut[c][lv])).all()


def test_slot_reuse_keeps_earlier_batches_intact():
    rng = np.random.default_rng(7)
    pre = IP.GpuImagePreprocessor(RELEASE_SPECS[1:3], "cuda:0", torch.bfloat16, slots=2)
    batches = [[rng.integers(0, 256, (rng.integers(20, 500), rng.integers(20, 500), 3), dtype=np.uint8)
                for _ in range(3)] for _ in range(5)]
    outs = [pre(b) for b in batches]          # no synchronisation in between
    torch.cuda.synchronize()
    for b, o in zip(batches, outs):
        for ti, sp in enumerate(RELEASE_SPECS[1:3]):
            for bi, im in enumerate(b):
                assert _equal(o[ti][bi], _want(im, sp, torch.bfloat16))


def test_prefetcher_matches_direct_and_moves_tensors():
    rng = np.random.default_rng(9)
    specs = RELEASE_SPECS[1:3]
    host_batches = []
    for k in range(4):
        imgs = [rng.integers(0, 256, (100 + 37 * k, 150 + i * 11, 3), dtype=np.uint8) for i in range(2)]
        host_batches.append(dict(raw_images=imgs, input_ids=torch.arange(8).view(2, 4) + k,
                                 image_aux_attention_masks_list=[torch.ones(2, 4, dtype=torch.bool)],
                                 image_sizes=[(im.shape[1], im.shape[0]) for im in imgs]))
    pre = IP.GpuImagePreprocessor(specs, "cuda:0", torch.bfloat16)
    got = list(IP.DevicePrefetcher(host_batches, pre))
    torch.cuda.synchronize()
    assert len(got) == 4
    for k, (hb, db) in enumerate(zip(host_batches, got)):
        assert "raw_images" not in db and db["input_ids"].is_cuda
        assert torch.equal(db["input_ids"].cpu(), hb["input_ids"])
        assert db["image_aux_attention_masks_list"][0].is_cuda and db["image_sizes"] == hb["image_sizes"]
        for ti, sp in enumerate(specs):
            for bi, im in enumerate(hb["raw_images"]):
                assert _equal(db["images"][ti][bi], _want(im, sp, torch.bfloat16))


def test_process_images_surface_matches_reference_expression():
    from PIL import Image
    from cambrian_amd.mm_utils import process_images
    from cambrian_amd.model.multimodal_encoder.base_encoder import ProcessorWrapper, SimpleImageTransform
    procs = [ProcessorWrapper(SimpleImageTransform(48, [0.5] * 3, [0.5] * 3), height=48, width=48),
             ProcessorWrapper(SimpleImageTransform(64, list(IMAGENET_MEAN), list(IMAGENET_STD), flavour="hf"), height=64,
                              width=64, image_mean=list(IMAGENET_MEAN))]
    rng = np.random.default_rng(1)
    pil = [Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in ((30, 70), (90, 41), (64, 64))]
    got = process_images(pil, procs, None)
    torch.cuda.synchronize()
    from cambrian_amd.mm_utils import expand2square
    for ti, p in enumerate(procs):
        assert got[ti].dtype == torch.float16 and got[ti].is_cuda
        for bi, im in enumerate(pil):
            R = p.crop_size["height"]
            aux = expand2square(im, tuple(int(x * 255) for x in p.image_mean)).resize((R, R))    # mm_utils.py:193
            want = p.preprocess(aux, return_tensors="pt")["pixel_values"][0].half()               # :194, :200
            assert _equal(got[ti][bi], want)


def test_errors():
    with pytest.raises(L.CambrianAmdError):
        IP.GpuImagePreprocessor(RELEASE_SPECS, "cpu")
    pre = IP.GpuImagePreprocessor(RELEASE_SPECS[:1], "cuda:0")
    with pytest.raises(ValueError):
        pre([np.zeros((4, 4), np.uint8)])
    lib = L.load()
    assert lib.cmb_image_preprocess(None, None, 1, None, None, None, None, 0, None, None, None) == -1
    assert lib.cmb_image_preprocess(None, None, 0, None, None, None, None, 0, None, None, None) == 0
