// hipapi.hpp -- maps the host templates on T to the _f32 / _f64 entry points of prost_hip.h.
#pragma once
#include "prost_hip.h"

namespace prost {

template <typename T> struct Api;

#define PROST_API_STRUCT(T, S)                                                    \
  template <> struct Api<T> {                                                     \
    static constexpr auto grad2d_fwd = prost_hip_grad2d_fwd_##S;                  \
    static constexpr auto grad2d_adj = prost_hip_grad2d_adj_##S;                  \
    static constexpr auto grad3d_fwd = prost_hip_grad3d_fwd_##S;                  \
    static constexpr auto grad3d_adj = prost_hip_grad3d_adj_##S;                  \
    static constexpr auto diags_fwd = prost_hip_diags_fwd_##S;                    \
    static constexpr auto diags_adj = prost_hip_diags_adj_##S;                    \
    static constexpr auto csr_spmv_acc = prost_hip_csr_spmv_acc_##S;              \
    static constexpr auto csr_spmv = prost_hip_csr_spmv_##S;                      \
    static constexpr auto pattern_spmv = prost_hip_pattern_spmv_tab_##S;          \
    static constexpr auto pattern_spmv_anchored = prost_hip_pattern_spmv_anchored_##S; \
    static constexpr auto fill = prost_hip_fill_##S;                              \
    static constexpr auto scale = prost_hip_scale_##S;                            \
    static constexpr auto sparse_kron_id_acc = prost_hip_sparse_kron_id_acc_##S;  \
    static constexpr auto id_kron_sparse_acc = prost_hip_id_kron_sparse_acc_##S;  \
    static constexpr auto sparse_kron_id = prost_hip_sparse_kron_id_##S;          \
    static constexpr auto id_kron_sparse = prost_hip_id_kron_sparse_##S;          \
    static constexpr auto prox_elem = prost_hip_prox_elem_##S;                    \
    static constexpr auto prox_elem_moreau = prost_hip_prox_elem_moreau_##S;      \
    static constexpr auto prox_elem_arg = prost_hip_prox_elem_arg_##S;            \
    static constexpr auto prox_epi_quad = prost_hip_prox_epi_quad_##S;            \
    static constexpr auto prox_elem_ind_sum = prost_hip_prox_elem_ind_sum_##S;    \
    static constexpr auto prox_elem_ind_simplex = prost_hip_prox_elem_ind_simplex_##S; \
    static constexpr auto transform_prescale = prost_hip_transform_prescale_##S;  \
    static constexpr auto transform_postscale = prost_hip_transform_postscale_##S; \
    static constexpr auto permute = prost_hip_permute_##S;                        \
    static constexpr auto prox_ind_halfspace = prost_hip_prox_ind_halfspace_##S;  \
    static constexpr auto prox_ind_soc = prost_hip_prox_ind_soc_##S;              \
    static constexpr auto prox_ind_sum = prost_hip_prox_ind_sum_##S;              \
    static constexpr auto moreau_prescale = prost_hip_moreau_prescale_##S;        \
    static constexpr auto moreau_postscale = prost_hip_moreau_postscale_##S;      \
    static constexpr auto pdhg_primal_arg = prost_hip_pdhg_primal_arg_##S;        \
    static constexpr auto pdhg_dual_arg = prost_hip_pdhg_dual_arg_##S;            \
    static constexpr auto pdhg_residual_primal = prost_hip_pdhg_residual_primal_##S; \
    static constexpr auto pdhg_residual_dual = prost_hip_pdhg_residual_dual_##S;  \
    static constexpr auto pdhg_w_variable = prost_hip_pdhg_w_variable_##S;        \
    static constexpr auto pdhg_z_variable = prost_hip_pdhg_z_variable_##S;        \
    static constexpr auto fused_primal = prost_hip_fused_primal_##S;              \
    static constexpr auto fused_dual = prost_hip_fused_dual_##S;                  \
    static constexpr auto fused_iteration = prost_hip_fused_iteration_##S;        \
    static constexpr auto fused_iteration2 = prost_hip_fused_iteration2_##S;      \
    static constexpr auto fused_iteration_rec = prost_hip_fused_iteration_rec_##S; \
    static constexpr auto fused_iteration_mc_rec = prost_hip_fused_iteration_mc_rec_##S; \
    static constexpr auto fused_iteration_mc_x2_rec = prost_hip_fused_iteration_mc_x2_rec_##S; \
    static constexpr auto fused_iteration3d_rec = prost_hip_fused_iteration3d_rec_##S; \
    static constexpr auto fused_iteration3d_pw_rec = prost_hip_fused_iteration3d_pw_rec_##S; \
    static constexpr auto fused_iteration3d_x2_rec = prost_hip_fused_iteration3d_x2_rec_##S; \
    static constexpr auto fused_iteration2_rec = prost_hip_fused_iteration2_rec_##S; \
    static constexpr auto fused_iterationk = prost_hip_fused_iterationk_##S;      \
    static constexpr auto fused_iterationk_rec = prost_hip_fused_iterationk_rec_##S; \
    static constexpr auto pdhg_rule_begin = prost_hip_pdhg_rule_begin_##S;        \
    static constexpr auto pdhg_rule_apply = prost_hip_pdhg_rule_apply_##S;        \
    static constexpr auto pdhg_residuals = prost_hip_pdhg_residuals_##S;          \
    static constexpr auto pdhg_fold_sums = prost_hip_pdhg_fold_sums_##S;          \
    static constexpr auto fused_iteration3d = prost_hip_fused_iteration3d_##S;    \
    static constexpr auto fused_iteration_mc = prost_hip_fused_iteration_mc_##S;  \
    static constexpr auto fused_iteration_mc_x2 = prost_hip_fused_iteration_mc_x2_##S; \
    static constexpr auto fused_iteration3d_pw = prost_hip_fused_iteration3d_pw_##S; \
    static constexpr auto fused_iteration3d_x2 = prost_hip_fused_iteration3d_x2_##S; \
    static constexpr auto compare = prost_hip_compare_##S;                        \
    static constexpr auto nrm2 = prost_hip_nrm2_##S;                              \
    static constexpr auto axpy = prost_hip_axpy_##S;                              \
    static constexpr auto admm_elem = prost_hip_admm_elem_##S;                    \
    static constexpr auto mask_merge = prost_hip_mask_merge_##S;                  \
    static constexpr auto cgls_stage = prost_hip_cgls_stage_##S;                  \
    static constexpr auto cgls_round = prost_hip_cgls_round_##S;                  \
    static constexpr auto cgls_round_timed = prost_hip_cgls_round_timed_##S;      \
    static constexpr auto cgls_init_fused = prost_hip_cgls_init_fused_##S;        \
    static constexpr auto cgls_pixel_round = prost_hip_cgls_pixel_round_##S;      \
    static constexpr auto cgls_pixel_round_timed = prost_hip_cgls_pixel_round_timed_##S; \
    static constexpr auto cgls_pixel_close = prost_hip_cgls_pixel_close_##S;      \
    static constexpr auto admm_fused_stage = prost_hip_admm_fused_stage_##S;      \
    static constexpr auto admm_stage = prost_hip_admm_stage_##S;                  \
    static constexpr auto normest_stage = prost_hip_normest_stage_##S;            \
    static constexpr auto normest_grad_round = prost_hip_normest_grad_round_##S;  \
  };

PROST_API_STRUCT(float, f32)
PROST_API_STRUCT(double, f64)
#undef PROST_API_STRUCT

inline int negate(float* x, size_t n, bool, void* s) { return prost_hip_negate_f32(x, n, s); }
inline int negate(double* x, size_t n, bool via_float, void* s) { return prost_hip_negate_f64(x, n, via_float ? 1 : 0, s); }
template <typename T> inline int dtype_id();
template <> inline int dtype_id<float>() { return 0; }
template <> inline int dtype_id<double>() { return 1; }

}  // namespace prost
