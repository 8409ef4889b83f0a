"""Pre-tuned hipBLASLt / rocBLAS solution choices for the frozen LLM's projection GEMMs (stock PyTorch-ROCm linears).

The decoder's GEMMs are not part of the hand-written hot path (north star: LLM decoder stays stock), but they are half
of the step, and PyTorch's default heuristic picks a forward (TN) solution that runs at ~1.28 PFLOP/s where the library
holds one at 1.6-1.9 PFLOP/s.  ``tools/tune_llm_gemms.py`` runs PyTorch's own TunableOp search once on an MI355X for
the ten shapes of Llama-3-8B at 16 images x 2048 tokens per GPU and stores the winners in
``tools/llm_gemm_tuning/llama3_8b_b16_gfx950.csv``; this module only LOADS that file (tuning disabled: nothing is
searched at run time, unknown shapes fall back to the default heuristic, a file from another ROCm / hipBLASLt build is
rejected by TunableOp's validators and ignored).

A round-1 experiment kept under tools/: measured neutral in the step (the default heuristic's picks are as fast under
sustained load), so no product code imports it."""
from __future__ import annotations

import os
import shutil
import tempfile

import torch

_CSV = os.path.join(os.path.dirname(os.path.abspath(__file__)), "llama3_8b_b16_gfx950.csv")


def load_tuned_llm_gemms(path: str = _CSV) -> bool:
    """Returns True when the pre-tuned table was handed to TunableOp."""
    if not torch.cuda.is_available() or not os.path.exists(path):
        return False
    try:
        t = torch.cuda.tunable
        # TunableOp rewrites its results file at process exit: give every process a private copy
        private = os.path.join(tempfile.mkdtemp(prefix="cambrian_tunableop_"), os.path.basename(path))
        shutil.copyfile(path, private)
        t.enable(True)
        t.tuning_enable(False)
        t.set_filename(private, insert_device_ordinal=False)
        return bool(t.read_file(private))
    except Exception:  # an optional speed-up of stock code must never take the step down
        try:
            torch.cuda.tunable.enable(False)
        except Exception:
            pass
        return False
