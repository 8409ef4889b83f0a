"""CPU: image pre-processing.

* the oracle (oracle/preprocess.py) against Pillow itself on seeded images and against the committed fixtures the
  reference's own expression produced (tests/golden/make_preprocess_golden.py);
* the library's HOST function cmb_resize_coeffs against the oracle's coefficient rows;
* the product's job table (cambrian_amd/train/image_pipeline.py) + the per-thread kernel code
  (cambrian_amd/csrc/preprocess_core.h) run thread by thread on the CPU (tests/csrc/preprocess_sim.cpp, g++)
  against the oracle — bit-exact; no GPU involved."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from cambrian_amd.train import image_pipeline as IP
from oracle import preprocess as OP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "preprocess_cases.npz")
OPENAI_MEAN = [0.48145466, 0.4578275, 0.40821073]


def _pil_reference(img, R, image_mean):
    """the reference expression (train_fsdp.py:1006), with Pillow"""
    from PIL import Image
    pil = Image.fromarray(img)
    w, h = pil.size
    bg = tuple(int(x * 255) for x in image_mean)
    if w != h:
        s = max(w, h)
        sq = Image.new("RGB", (s, s), bg)
        sq.paste(pil, (0, (w - h) // 2) if w > h else ((h - w) // 2, 0))
        pil = sq
    return np.asarray(pil.resize((R, R)))


SHAPES = [(50, 37, 64), (37, 50, 24), (96, 96, 96), (31, 31, 336), (640, 480, 384), (333, 500, 378), (1200, 900, 336),
          (17, 300, 48), (1, 1, 8), (2, 7, 5), (300, 200, 1024)]


@pytest.mark.parametrize("w,h,R", SHAPES)
def test_oracle_matches_pillow(w, h, R):
    rng = np.random.default_rng(w * 7919 + h * 31 + R)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = OP.resize_square(OP.expand2square(img, OP.background_of(OPENAI_MEAN)), R)
    assert np.array_equal(got, _pil_reference(img, R, OPENAI_MEAN))


def _golden():
    g = np.load(GOLDEN)
    towers = [dict(name=str(n), side=int(s), flavour=str(f), pad_mean=list(p), mean=list(m), std=list(d))
              for n, s, f, p, m, d in zip(g["tower_names"], g["tower_sides"], g["tower_flavours"], g["tower_pad_mean"],
                                          g["tower_mean"], g["tower_std"])]
    return g, towers


def test_oracle_matches_reference_fixtures():
    g, towers = _golden()
    for ci in range(int(g["n_cases"])):
        img = g[f"img{ci}"]
        for t in towers:
            sq = OP.resize_square(OP.expand2square(img, OP.background_of(t["pad_mean"])), t["side"])
            assert np.array_equal(sq, g[f"u8_{ci}_{t['name']}"]), (ci, t["name"])
            if ci < 2:
                px = OP.preprocess(img, t["side"], t["pad_mean"], t["mean"], t["std"], t["flavour"])
                assert px.dtype == np.float32
                assert np.array_equal(px, g[f"px_{ci}_{t['name']}"]), (ci, t["name"])


def test_product_lut_equals_oracle_lut():
    for flavour in ("hf", "torchvision"):
        for mean, std in ((OPENAI_MEAN, [0.26862954, 0.26130258, 0.27577711]), ([0.5] * 3, [0.5] * 3),
                          ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])):
            sp = IP.TowerPixelSpec(32, tuple(OPENAI_MEAN), tuple(mean), tuple(std), flavour)
            assert np.array_equal(sp.lut(), OP.pixel_lut(mean, std, flavour))
            assert sp.background == sum(int(x * 255) << (8 * i) for i, x in enumerate(OPENAI_MEAN))


@pytest.mark.parametrize("n_in,n_out", [(50, 64), (64, 50), (336, 1024), (1200, 336), (7, 5), (1, 8), (4000, 378),
                                        (385, 384), (383, 384)])
def test_library_host_coefficients_match_oracle(n_in, n_out):
    bounds, coefs, ksize = IP.library_coefficients(n_in, n_out)        # host function of libcambrian_amd.so
    ob, oc, ok = OP.resize_coeffs(n_in, n_out)
    assert ksize == ok
    assert np.array_equal(bounds, ob)
    assert np.array_equal(coefs, oc.T)                                   # the library stores them tap-major


def test_job_table_layout():
    specs = [IP.TowerPixelSpec(24, tuple(OPENAI_MEAN), (0.5,) * 3, (0.5,) * 3, "torchvision"),
             IP.TowerPixelSpec(64, tuple(OPENAI_MEAN), tuple(OPENAI_MEAN), (0.25,) * 3, "hf")]
    plan = IP.build_plan([(40, 61), (90, 37), (24, 24), (48, 64)], specs, lambda a, b: _oracle_coefs(a, b))
    j = plan.jobs
    assert len(j) == 8 and plan.n_images == 4
    assert (j["side"] == np.repeat([61, 90, 24, 64], 2)).all()
    assert tuple(j[0][["off_x", "off_y"]]) == (0, 10) and tuple(j[2][["off_x", "off_y"]]) == (26, 0)
    assert j[4]["ksize"] == 0 and j[7]["ksize"] == 0                    # same-size: copy
    assert j[5]["ksize"] > 0 and j[6]["ksize"] > 0
    assert (j["tmp_off"] % 4 == 0).all() and (j["src_off"] % 16 == 0).all()
    # every tower's block is [B,3,R,R] contiguous and 16-byte aligned for 2-byte elements
    for t, (off, side) in enumerate(plan.tower_out):
        assert off % 8 == 0
        assert (j["dst_off"][t::2] == off + np.arange(4) * 3 * side * side).all()
    # the same (S, R) pair shares one coefficient block
    plan2 = IP.build_plan([(50, 50), (50, 30)], specs, lambda a, b: _oracle_coefs(a, b))
    assert plan2.jobs["coef_off"][0] == plan2.jobs["coef_off"][2]
    with pytest.raises(ValueError):
        IP.build_plan([(0, 5)], specs, lambda a, b: _oracle_coefs(a, b))


def _oracle_coefs(n_in, n_out):
    b, c, k = OP.resize_coeffs(n_in, n_out)
    return b, np.ascontiguousarray(c.T), k


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    d = tmp_path_factory.mktemp("ppsim")
    so = str(d / "libppsim.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tests", "csrc", "preprocess_sim.cpp"),
                    "-o", so], check=True)
    lib = C.CDLL(so)
    lib.sim_resize_coeffs.restype = C.c_int
    lib.sim_resize_coeffs.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.sim_image_preprocess.restype = C.c_int
    lib.sim_image_preprocess.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 6
    return lib


def _sim_coefs(lib):
    def fn(n_in, n_out):
        k = lib.sim_resize_coeffs(n_in, n_out, None, None)
        b = np.empty((n_out, 2), np.int32)
        c = np.empty((k, n_out), np.int32)
        assert lib.sim_resize_coeffs(n_in, n_out, b.ctypes.data, c.ctypes.data) == k
        return b, c, k
    return fn


def _run_sim(lib, images, specs):
    plan = IP.build_plan([im.shape[:2] for im in images], specs, _sim_coefs(lib))
    src = np.zeros(max(plan.src_bytes, 1), np.uint8)
    for im, off in zip(images, plan.src_offsets):
        src[off:off + im.size] = im.reshape(-1)
    tmp = np.full(max(plan.tmp_bytes, 4), 0xAB, np.uint8)
    dst = np.full(plan.out_elems, np.nan, np.float32)
    assert lib.sim_image_preprocess(plan.jobs.ctypes.data, len(plan.jobs), src.ctypes.data, plan.bounds.ctypes.data,
                                    plan.coefs.ctypes.data, plan.lut.ctypes.data, tmp.ctypes.data, dst.ctypes.data) == 0
    B = len(images)
    return [dst[o:o + B * 3 * r * r].reshape(B, 3, r, r) for o, r in plan.tower_out]


def test_kernel_code_on_cpu_matches_oracle_and_fixtures(sim):
    g, towers = _golden()
    specs = [IP.TowerPixelSpec(t["side"], tuple(t["pad_mean"]), tuple(t["mean"]), tuple(t["std"]), t["flavour"])
             for t in towers]
    images = [g[f"img{ci}"] for ci in range(int(g["n_cases"]))]
    outs = _run_sim(sim, images, specs)
    for ti, t in enumerate(towers):
        lut = OP.pixel_lut(t["mean"], t["std"], t["flavour"])
        for ci in range(len(images)):
            u8 = g[f"u8_{ci}_{t['name']}"]
            want = np.stack([lut[c][u8[:, :, c]] for c in range(3)])
            assert np.array_equal(outs[ti][ci], want), (ci, t["name"])
            if ci < 2:
                assert np.array_equal(outs[ti][ci], g[f"px_{ci}_{t['name']}"])


def test_kernel_code_on_cpu_ragged_batch(sim):
    rng = np.random.default_rng(5)
    specs = [IP.TowerPixelSpec(42, tuple(OPENAI_MEAN), (0.5,) * 3, (0.5,) * 3, "torchvision"),
             IP.TowerPixelSpec(47, (0.485, 0.456, 0.406), (0.485, 0.456, 0.406), (0.229, 0.224, 0.225), "hf"),
             IP.TowerPixelSpec(128, tuple(OPENAI_MEAN), tuple(OPENAI_MEAN), (0.26862954, 0.26130258, 0.27577711), "torchvision")]
    shapes = [(33, 80), (80, 33), (47, 47), (128, 100), (1, 1), (3, 200), (150, 149)]
    images = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    outs = _run_sim(sim, images, specs)
    for ti, sp in enumerate(specs):
        for bi, im in enumerate(images):
            want = OP.preprocess(im, sp.out_side, sp.pad_mean, sp.mean, sp.std, sp.flavour)
            assert np.array_equal(outs[ti][bi], want), (ti, bi)
    assert not np.isnan(np.concatenate([o.reshape(-1) for o in outs])).any()
