"""Per-step kernel time by family from a `tools/rocpd_summary.py` table (profiles/r06*_bench_b24_kernel_stats.md).

    python tools/kernel_families.py profiles/r06b_bench_b24_kernel_stats.md 30 > profiles/r06b_kernel_families_per_step.md

The second argument is the number of steps in the trace (warm-up + timed + per-launch-event + masked-case steps)."""
import re
import sys

FAMILIES = [   # (label, regex on the kernel name) — first match wins
    ("hipBLASLt (frozen decoder)", r"^Custom_Cijk|^Cijk_"),
    ("own NT GEMM, pair launches (cmb_gemm_pair)", r"gemm_nt_p5_kernel<\d+, true>"),
    ("own NT GEMM, persistent 4-wave kernel (single launches)", r"gemm_nt_p5_kernel"),
    ("own NT GEMM, 8-wave kernel", r"gemm_nt_256_kernel"),
    ("own NT GEMM, 128 x 128 kernel", r"gemm_nt_kernel"),
    ("gemm_k64 / gemm_small_m", r"gemm_k64|gemm_small_m"),
    ("gemm_tn + splitk_reduce", r"gemm_tn_kernel|splitk_reduce"),
    ("dwconv7x7_col", r"dwconv7x7"),
    ("decoder attention (flash_*)", r"flash_"),
    ("decoder elementwise (swiglu_bwd, act_mul, rmsnorm*, qkv_rope, ce_*)", r"swiglu_bwd|act_mul|rmsnorm|qkv_rope|ce_fwd|ce_bwd|rope_"),
    ("LayerNorm forward, multi-layer pass (cmb_layernorm_fwd_multi)", r"layernorm_fwd_multi"),
    ("LayerNorm forward (per-launch kernels)", r"layernorm_fwd|row_stats"),
    ("LayerNorm backward (multi + single)", r"layernorm_bwd"),
    ("vit_attn_dma", r"vit_attn"),
    ("sva_abs fwd + bwd", r"sva_abs|sva_fwd|sva_bwd"),
    ("copy_rows, colsum, resample, transpose, patchify, act_bwd, fold_kv, cast, weight_prep, token_mean, splice", r"copy_rows|colsum|resample|transpose|patchify|act_bwd|fold_kv|cast_kernel|weight_prep|token_mean|embed_splice|bcast_rows|scatter|gather"),
    ("ATen fills / adds / copies / optimizer (FillFunctor, CUDAFunctor_add, copyBuffer, multi_tensor_apply, elementwise)", r"at::native|__amd_rocclr"),
]


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    acc = {lab: [0, 0.0] for lab, _ in FAMILIES}
    other = [0, 0.0]
    for line in open(path):
        m = re.match(r"\| `(.*?)` \| (\d+) \| ([\d.]+) \|", line)
        if not m:
            continue
        name, calls, total = m.group(1), int(m.group(2)), float(m.group(3))
        for lab, rx in FAMILIES:
            if re.search(rx, name):
                acc[lab][0] += calls
                acc[lab][1] += total
                break
        else:
            other[0] += calls
            other[1] += total
    print(f"# kernel time per step by family ({path}, {int(steps)} steps in the trace; tools/kernel_families.py)\n")
    print("| family | launches per step | ms per step |\n|---|---:|---:|")
    for lab, _ in FAMILIES:
        n, t = acc[lab]
        print(f"| {lab} | {n / steps:.0f} | {t / steps:.2f} |")
    print(f"| other listed kernels | {other[0] / steps:.0f} | {other[1] / steps:.2f} |")


if __name__ == "__main__":
    main()
