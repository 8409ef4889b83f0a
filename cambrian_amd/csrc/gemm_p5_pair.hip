// gemm_p5_pair.hip — the PAIR = true instantiations of gemm_nt_p5_kernel (cmb_gemm_pair: two problems, one launch) as their own
// translation unit: gemm_p5.hip compiled with CMB_P5_PAIR_TU emits only those kernels and p5_pair_set_attr / p5_pair_launch.
#define CMB_P5_PAIR_TU 1
#include "gemm_p5.hip"
