"""The committed fixtures ARE what the committed recipe produces from the reference.

Runs every ``golden_*`` recipe of tests/golden/make_golden.py against /root/reference into a scratch directory and compares the
result with the committed file tensor for tensor (bit-exact: same torch build, same seeds, CPU).  Skipped where the reference is
absent (the GPU box); about a minute on the build container."""
import os
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
RECIPES = {  # recipe -> file it writes
    "sva": "sva_small.pt", "sva_k1024": "sva_k1024.pt", "towers": "towers_small.pt", "collator": "collator_cases.pt",
    "arch": "arch_small.pt", "arch_groups": "arch_groups_small.pt", "llama": "llama_small.pt",
    "arch_dynamic": "arch_dynamic_small.pt", "config0": "config0_small.pt", "sampler": "sampler_cases.pt",
}

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/cambrian"), reason="needs the reference checkout")


def same(a, b, path="fx"):
    """Recursive equality: tensors bit for bit (NaN == NaN), containers element by element; returns the first difference."""
    if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
        if not (isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor)):
            return f"{path}: tensor vs {type(b).__name__}"
        if a.dtype != b.dtype or a.shape != b.shape:
            return f"{path}: {a.dtype}{tuple(a.shape)} vs {b.dtype}{tuple(b.shape)}"
        return None if torch.equal(a, b) or torch.equal(a.nan_to_num(0.0), b.nan_to_num(0.0)) and torch.equal(a.isnan(), b.isnan()) \
            else f"{path}: values differ (max abs {float((a.double() - b.double()).abs().max()):.3e})"
    if isinstance(a, dict):
        if not isinstance(b, dict) or sorted(map(str, a)) != sorted(map(str, b)):
            return f"{path}: keys differ: {sorted(map(str, a))[:12]} vs {sorted(map(str, b))[:12] if isinstance(b, dict) else type(b)}"
        for k in a:
            d = same(a[k], b[k], f"{path}[{k!r}]")
            if d:
                return d
        return None
    if isinstance(a, (list, tuple)):
        if not isinstance(b, (list, tuple)) or len(a) != len(b):
            return f"{path}: sequence lengths differ"
        for i, (x, y) in enumerate(zip(a, b)):
            d = same(x, y, f"{path}[{i}]")
            if d:
                return d
        return None
    if isinstance(a, float) and isinstance(b, float):
        return None if (a == b or (a != a and b != b)) else f"{path}: {a} vs {b}"
    return None if a == b else f"{path}: {a!r} vs {b!r}"


@pytest.mark.parametrize("recipe", sorted(RECIPES))
def test_recipe_reproduces_the_committed_fixture(recipe, tmp_path):
    env = dict(os.environ, CAMBRIAN_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py"), recipe], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, f"recipe golden_{recipe} does not run:\n{r.stderr[-2000:]}"
    new = torch.load(tmp_path / RECIPES[recipe], weights_only=False)
    old = torch.load(os.path.join(GOLDEN, RECIPES[recipe]), weights_only=False)
    diff = same(old, new)
    assert diff is None, f"{RECIPES[recipe]} differs from what the committed recipe produces: {diff}"
