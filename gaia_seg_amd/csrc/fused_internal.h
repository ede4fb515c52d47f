// Internal (C++ linkage) interfaces between the conv, BatchNorm and fused-layer translation units.
#pragma once
#include "common.h"

namespace gs {

struct ConvFwdInfo {
  int mode;          // 0: y complete; 1: per-tile BN partials written; 2: split-K slabs left to reduce
  int splits, tiles_m, bm;
  const float* tile_part;   // mode 1: where the per-tile partials are (inside the workspace)
  // in: the BatchNorm behind this conv (gs_conv_bn_forward), so that a launch with few row tiles can
  // merge its tile partials and write `fin_coeffs` itself (igemm_core.h column_finalize_stats);
  // out: finalized = it did, no bn_tile_finalize launch is needed
  const gs_bn_args* fin_bn;
  float* fin_coeffs;
  bool finalized;
  float* slab;       // mode 2: [splits][M][Co]
  size_t slab_bytes;
  bool timed;        // a K3 timer interval is open (mode 2: the caller closes it)
  double flops;
};

// igemm_fwd.hip
int conv2d_forward_impl(const gs_conv_desc* d, const float* x, const float* w, const float* bias,
                        const float* addend, float* y, void* workspace, size_t workspace_bytes,
                        void* stream, bool want_stats, ConvFwdInfo* info);

// Arrival counters of the split-K launches that combine their slabs inside the launch (igemm_core.h
// splitk_publish): one zero-at-rest buffer per (device, stream), so that launches on different streams
// never share a counter and launches on one stream reuse it in order.  NULL = not available (switched
// off with GS_SPLITK_INKERNEL=0 / gs_debug_set_splitk_inkernel, too many tiles, or no memory): the
// caller then keeps the separate reduce launch.
unsigned* splitk_tickets(hipStream_t st, long ntiles);
// Arrival counters per COLUMN tile of the launches that merge their per-tile partials themselves
// (igemm_core.h column_arrive); NULL when switched off (the default; GS_COL_FINALIZE=1 /
// gs_debug_set_col_finalize turn it on),
// when the launch has more than GS_COL_FINALIZE_MAX (160) row tiles, or without memory.
unsigned* column_tickets(hipStream_t st, long tiles_m, long tiles_n, int kind /* 1 fwd, 2 dgrad */);

// norm.hip
size_t bn_fused_reduce_bytes(long rows, int C);
int bn_reduce_stats_finalize(const float* slab, int splits, long rows, int C, float* y, int ldy,
                             const float* gamma, const float* beta, float eps, float momentum,
                             float* running_mean, float* running_var, float* coeffs, float* part,
                             size_t part_bytes, hipStream_t st, int role, bool timed, double flops);
int bn_tile_finalize(const float* part, int np, int bm, long rows, int C, const float* gamma,
                     const float* beta, float eps, float momentum, float* running_mean,
                     float* running_var, float* coeffs, hipStream_t st);

// stem.hip: the 7x7 stride-2 stem convolution of the NCHW image (forward with BatchNorm tile partials,
// weight gradient); stem_conv_ok says whether a descriptor takes these kernels
bool stem_conv_ok(const gs_conv_desc* d);
bool stem_wgrad_on();   // GS_STEM_WGRAD=0 puts the weight gradient back on the generic kernel
size_t stem_wgrad_slab_bytes(const gs_conv_desc* d);
int stem_forward(const gs_conv_desc* d, const float* x, const float* w, float* y, float* tile_stats,
                 int* np, hipStream_t st);
int stem_wgrad(const gs_conv_desc* d, const float* x, const float* dy, float* dw, void* workspace,
               size_t workspace_bytes, hipStream_t st);

// norm.hip: z = relu?((y - mean) * scale + beta + ((residual - rmean) * rscale + rbeta)), optional mask
int bn_apply_resaff(const float* x, int64_t rows, int32_t C, int32_t ldx, const float* coeffs,
                    const float* residual, int32_t ld_res, const float* rcoeffs, int32_t relu, float* y,
                    int32_t ldy, uint8_t* mask, void* stream);

// norm.hip: out[0..width) = fixed-order sum over nparts partial rows (quad-major layout)
int bn_sum_partials(const float* part, int nparts, int width, float* out, hipStream_t st);

size_t bn_reduce_bnbwd_bytes(long rows, int C);
int bn_reduce_bnbwd(const float* slab, int splits, long rows, int C, float* dx, int ld_dx,
                    int accumulate, const gs_bn_bwd_fuse* bw, float* part, size_t part_bytes,
                    hipStream_t st);

// igemm_dgrad.hip: data gradient, optionally with the producer BatchNorm's backward reduction folded
// into the epilogue (gs_bn_bwd_fuse); *fused tells whether that happened
int conv2d_dgrad_impl(const gs_conv_desc* d, const float* dy, const float* w, float* dx,
                      int accumulate, void* workspace, size_t workspace_bytes, void* stream,
                      const gs_bn_bwd_fuse* bw, int* fused);
size_t dgrad_bnbwd_part_bytes(const gs_conv_desc* d);

// fused_layers.hip: live timer of the role-1 (K3) forward launches
bool k3_prof_on();
void k3_prof_begin(hipStream_t st);
void k3_prof_end(hipStream_t st, double flops);

}  // namespace gs
