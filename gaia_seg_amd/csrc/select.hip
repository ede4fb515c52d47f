// OHEM pixel sampling support on gfx950: the k-th smallest
// ground-truth-class probability over the valid pixels, and the 0/1 pixel weights derived from it.
//
// Replaces, for mmseg's OHEMPixelSampler.sample (call sites
// gaiaseg/models/decode_heads/dynamic_fcn_head.py:70-71,147-148):
//   sort_prob, _ = seg_prob[valid_mask].sort();  min_threshold = sort_prob[min(batch_kept, n-1)]
//   threshold = max(min_threshold, thresh);       weight = (seg_prob < threshold) & valid
// A full sort of ~1 M floats is replaced by a 3-pass radix select (11 + 11 + 10 bits) on the float
// bit patterns (probabilities are >= 0, so bit order == numeric order; ignored pixels carry 2.0 and
// sort last).  Only integer atomics are used: the result is exact and bit-reproducible.
#include "common.h"

namespace gs {

constexpr int kBins = 2048;

struct SelectState {
  unsigned prefix;     // bits decided so far
  unsigned mask;       // which bits are decided
  long long k;         // rank still to find inside the current prefix class
  long long n_valid;   // number of values < 2.0 (set by pass 0)
  float value;         // final answer
  int done;
};

__device__ __forceinline__ unsigned digit_of(unsigned bits, int pass) {
  return pass == 0 ? (bits >> 21) : pass == 1 ? ((bits >> 10) & 2047u) : (bits & 1023u);
}

__global__ __launch_bounds__(256) void select_hist_kernel(const float* __restrict__ v, long n,
                                                          int pass, const SelectState* st,
                                                          unsigned* __restrict__ hist) {
  __shared__ unsigned sh[kBins];
  for (int i = threadIdx.x; i < kBins; i += 256) sh[i] = 0;
  __syncthreads();
  const unsigned prefix = st->prefix, mask = st->mask;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    const unsigned b = __float_as_uint(v[i]);
    if ((b & mask) == prefix) atomicAdd(&sh[digit_of(b, pass)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kBins; i += 256)
    if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

// one block: find the digit whose cumulative count crosses k; update the state
__global__ __launch_bounds__(256) void select_scan_kernel(unsigned* __restrict__ hist, int pass,
                                                          long long k_req, SelectState* st) {
  __shared__ long long cum[kBins];
  for (int i = threadIdx.x; i < kBins; i += 256) cum[i] = hist[i];
  __syncthreads();
  if (threadIdx.x == 0) {  // 2048-element serial prefix sum: ~2 us, runs three times per call
    long long run = 0;
    for (int i = 0; i < kBins; ++i) { run += cum[i]; cum[i] = run; }
    long long k = st->k;
    if (pass == 0) {
      // values < 2.0 have top-11 bits < (bits(2.0) >> 21) = 0x200
      const long long nv = cum[0x200 - 1];
      st->n_valid = nv;
      k = nv > 0 ? (k_req < nv - 1 ? k_req : nv - 1) : -1;
      if (k < 0) { st->done = 1; st->value = 0.f; }
    }
    if (!st->done) {
      int d = 0;
      while (d < kBins - 1 && cum[d] <= k) ++d;
      const long long below = d > 0 ? cum[d - 1] : 0;
      const int shift = pass == 0 ? 21 : pass == 1 ? 10 : 0;
      const unsigned m = pass == 0 ? 0xFFE00000u : pass == 1 ? 0x001FFC00u : 0x000003FFu;
      st->prefix |= (unsigned)d << shift;
      st->mask |= m;
      st->k = k - below;
      if (pass == 2) st->value = __uint_as_float(st->prefix);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kBins; i += 256) hist[i] = 0;  // ready for the next pass
}

// weight = 1 where prob < max(kth, thresh) for valid pixels (prob <= 1), else 0
__global__ __launch_bounds__(256) void ohem_weight_kernel(const float* __restrict__ prob, long n,
                                                          const SelectState* st, float thresh,
                                                          int use_thresh, float* __restrict__ w) {
  const float kth = st->value;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    const float p = prob[i];
    const bool valid = p <= 1.5f;
    bool keep;
    if (use_thresh) keep = p < fmaxf(kth, thresh);
    else keep = p <= kth;  // the batch_kept smallest probabilities == largest losses
    w[i] = (valid && keep) ? 1.f : 0.f;
  }
}

}  // namespace gs

using namespace gs;

extern "C" size_t gs_ohem_workspace_bytes(void) {
  return sizeof(SelectState) + 64 + kBins * sizeof(unsigned);
}

extern "C" int gs_ohem_weights(const float* prob, int64_t n, int64_t batch_kept, float thresh,
                               int32_t use_thresh, float* weight, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (!prob || !weight || !workspace) return GS_E_NULL;
  if (n <= 0 || batch_kept < 0) return GS_E_BADARG;
  if (workspace_bytes < gs_ohem_workspace_bytes()) return GS_E_WORKSPACE;
  if (!aligned16(workspace)) return GS_E_ALIGN;
  hipStream_t st = as_stream(stream);
  SelectState* state = static_cast<SelectState*>(workspace);
  unsigned* hist = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + 64);
  hipError_t e = hipMemsetAsync(workspace, 0, gs_ohem_workspace_bytes(), st);
  if (e != hipSuccess) return static_cast<int>(e);
  const int grid = stream_grid(n, 256);
  // thresh is None in mmseg => keep exactly the batch_kept hardest pixels: rank batch_kept - 1
  const long long k_req = use_thresh ? batch_kept : (batch_kept > 0 ? batch_kept - 1 : 0);
  for (int pass = 0; pass < 3; ++pass) {
    hipLaunchKernelGGL(select_hist_kernel, dim3(grid), dim3(256), 0, st, prob, (long)n, pass, state,
                       hist);
    hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(256), 0, st, hist, pass, k_req, state);
  }
  hipLaunchKernelGGL(ohem_weight_kernel, dim3(grid), dim3(256), 0, st, prob, (long)n, state, thresh,
                     use_thresh, weight);
  return launch_status();
}

// ------------------------------------------------------------------------------------------
// mIoU support: confusion matrix of predictions vs labels (mmseg `intersect_and_union` /
// dataset.evaluate(metric='mIoU'), driven by gaiaseg/core/evaluation/cross_arch_eval_hooks.py:85-92
// through mmseg's multi_gpu_test).  conf[label * C + pred] += 1 for labels != ignore_index.
// LDS-privatised integer atomics, exact and order-independent.
// ------------------------------------------------------------------------------------------
namespace gs {
constexpr int kMaxLdsBins = 8192;
__global__ __launch_bounds__(256) void confusion_kernel(const int64_t* __restrict__ pred,
                                                        const int64_t* __restrict__ label, long n,
                                                        int C, int ignore,
                                                        unsigned long long* __restrict__ conf) {
  __shared__ unsigned sh[kMaxLdsBins];
  const int bins = C * C;
  const bool use_lds = bins <= kMaxLdsBins;
  if (use_lds) {
    for (int i = threadIdx.x; i < bins; i += 256) sh[i] = 0;
    __syncthreads();
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    const long l = label[i], p = pred[i];
    if (l == ignore || l < 0 || l >= C || p < 0 || p >= C) continue;
    const int b = (int)l * C + (int)p;
    if (use_lds) atomicAdd(&sh[b], 1u);
    else atomicAdd(&conf[b], 1ull);
  }
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += 256)
      if (sh[i]) atomicAdd(&conf[i], (unsigned long long)sh[i]);
  }
}
}  // namespace gs

extern "C" int gs_confusion_matrix(const int64_t* pred, const int64_t* label, int64_t n,
                                   int32_t num_classes, int32_t ignore_index, uint64_t* conf,
                                   void* stream) {
  if (!pred || !label || !conf) return GS_E_NULL;
  if (n <= 0 || num_classes <= 0 || num_classes > 4096) return GS_E_BADARG;
  hipLaunchKernelGGL(gs::confusion_kernel, dim3(gs::stream_grid(n, 256)), dim3(256), 0,
                     gs::as_stream(stream), pred, label, (long)n, num_classes, ignore_index,
                     reinterpret_cast<unsigned long long*>(conf));
  return gs::launch_status();
}
