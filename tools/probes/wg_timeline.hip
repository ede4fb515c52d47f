// Per-workgroup timeline of the fast forward conv kernel (diagnostic build ABL = 9): entry, start of
// the K loop, loop length, end of the epilogue -- 100 MHz s_memrealtime stamps per workgroup.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "igemm_core.h"
using namespace gs;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// pad: dynamic LDS bytes added to the launch (caps the workgroups a CU can hold); the placement
// histogram comes from HW_REG_HW_ID / HW_REG_XCC_ID read by every wave of the stamped build.
template <int KS, bool PAIR = false>
void run(const char* name, int H, int W, int Ci, int Co, int pad = 0) {
  const int N = 2; const long M = (long)N * H * W;
  float *x, *w, *y; unsigned long long* dbg;
  CK(hipMalloc(&x, M * Ci * 4)); CK(hipMalloc(&w, (long)KS * KS * Ci * Co * 4)); CK(hipMalloc(&y, M * Co * 4));
  CK(hipMemset(x, 0, M * Ci * 4)); CK(hipMemset(w, 0, (long)KS * KS * Ci * Co * 4));
  IgemmArgs a{};
  a.src = x; a.dense = w; a.out = y;
  a.s_c = 1; a.s_w = Ci; a.s_h = (long)W * Ci; a.s_n = (long)H * W * Ci;
  a.Hs = H; a.Ws = W; a.Cs = Ci; a.Hp = H; a.Wp = W; a.npix = (int)M; a.KW = KS; a.taps = KS * KS;
  a.mul_h = a.mul_w = 1; a.base_h = a.base_w = -(KS / 2); a.step_h = a.step_w = 1; a.div_h = a.div_w = 1;
  a.d_tap = (long)Ci * Co; a.d_row = Co; a.n_lim = Co; a.M = (int)M; a.Nn = Co; a.Ktot = KS * KS * Ci;
  a.ld_out = Co; a.nk_total = a.Ktot / 16; a.nk_per_split = a.nk_total;
  a.tiles_m = (int)(M / 64); a.tiles_n = Co / 64; a.nsplits = 1; a.tile_order = 0;
  a.kh_n = KS; a.kw_n = KS; a.d_tap_h = (long)KS * a.d_tap; a.d_tap_w = a.d_tap;
  a.src_bytes = (unsigned)(M * Ci * 4); a.dense_bytes = (unsigned)((long)KS * KS * Ci * Co * 4);
  const int tiles = a.tiles_m * a.tiles_n;
  CK(hipMalloc(&dbg, (long)tiles * 4 * 64)); CK(hipMemset(dbg, 0, (long)tiles * 4 * 64));
  // timing of the production build
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((igemm_rows_fast_kernel<64, 64, false, KS, 0, 0, true, PAIR, false>), dim3(tiles), dim3(256), pad, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((igemm_rows_fast_kernel<64, 64, false, KS, 0, 0, true, PAIR, false>), dim3(tiles), dim3(256), pad, 0, a);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  IgemmArgs b = a; b.slab = reinterpret_cast<float*>(dbg);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((igemm_rows_fast_kernel<64, 64, false, KS, 9, 0, true, PAIR, false>), dim3(tiles), dim3(256), pad, 0, b);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h((size_t)tiles * 4 * 8);
  CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int g = 0; g < tiles; ++g) { t0 = std::min(t0, h[(size_t)g * 32 + 2]); t1 = std::max(t1, h[(size_t)g * 32 + 3]); }
  std::vector<double> ent, pre, loop, tot, endt;
  for (int g = 0; g < tiles; ++g) {
    const unsigned long long* o = &h[(size_t)g * 32];
    ent.push_back((o[2] - t0) * 0.01); pre.push_back((o[0] - o[2]) * 0.01);
    tot.push_back((o[3] - o[2]) * 0.01); endt.push_back((o[3] - t0) * 0.01);
    loop.push_back((double)o[1]);
  }
  auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
  // placement: workgroups per CU (xcc, se, cu) and waves per SIMD
  {
    std::vector<int> per_cu(8 * 8 * 16, 0), per_simd(8 * 8 * 16 * 4, 0);
    int split_wg = 0;
    for (int g = 0; g < tiles; ++g) {
      int cu0 = -1; bool same = true;
      for (int wv = 0; wv < 4; ++wv) {
        const unsigned long long v = h[(size_t)g * 32 + wv * 8 + 7];
        const unsigned hw = (unsigned)v, xcc = (unsigned)(v >> 32) & 15;
        const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, se = (hw >> 13) & 7;
        const int id = (xcc * 8 + se) * 16 + cu;
        if (wv == 0) { cu0 = id; per_cu[id]++; } else if (id != cu0) same = false;
        per_simd[id * 4 + simd]++;
      }
      if (!same) ++split_wg;
    }
    int hist[16] = {0}, hs[16] = {0}, used = 0;
    for (size_t i = 0; i < per_cu.size(); ++i) if (per_cu[i]) { ++used; hist[std::min(per_cu[i], 15)]++; }
    for (size_t i = 0; i < per_simd.size(); ++i) if (per_cu[i / 4]) hs[std::min(per_simd[i], 15)]++;
    printf("%s [pad %d B]\n  placement: %d CUs used;", name, pad, used);
    for (int k = 1; k < 16; ++k) if (hist[k]) printf(" %d CUs x %d WGs;", hist[k], k);
    printf("  waves per SIMD:");
    for (int k = 0; k < 16; ++k) if (hs[k]) printf(" %d x %d;", hs[k], k);
    printf("  (workgroups spanning CUs: %d)\n", split_wg);
    // mean K-loop ticks by the number of workgroups on the CU
    double sum[16] = {0}; int cnt[16] = {0};
    for (int g = 0; g < tiles; ++g) {
      const unsigned long long v = h[(size_t)g * 32 + 7];
      const unsigned hw = (unsigned)v, xcc = (unsigned)(v >> 32) & 15;
      const int id = (xcc * 8 + ((hw >> 13) & 7)) * 16 + ((hw >> 8) & 15);
      const int k = std::min(per_cu[id], 15);
      sum[k] += (double)h[(size_t)g * 32 + 1]; cnt[k]++;
    }
    printf("  mean K-loop ticks by WGs on the CU:");
    for (int k = 1; k < 16; ++k) if (cnt[k]) printf(" %d: %.0f;", k, sum[k] / cnt[k]);
    printf("\n");
  }
  printf("%s: %d workgroups, production build %.1f us per launch; stamped build first entry -> last end %.1f us\n",
         name, tiles, ms * 1e3 / 20, (t1 - t0) * 0.01);
  printf("  entry after first entry   p10 %.2f  p50 %.2f  p90 %.2f  max %.2f us\n", pct(ent, .1), pct(ent, .5), pct(ent, .9), pct(ent, 1));
  printf("  entry -> K loop (setup)   p10 %.2f  p50 %.2f  p90 %.2f us\n", pct(pre, .1), pct(pre, .5), pct(pre, .9));
  printf("  K loop (s_memtime ticks)  p10 %.0f  p50 %.0f  p90 %.0f\n", pct(loop, .1), pct(loop, .5), pct(loop, .9));
  printf("  entry -> end of epilogue  p10 %.2f  p50 %.2f  p90 %.2f us\n", pct(tot, .1), pct(tot, .5), pct(tot, .9));
  printf("  end after first entry     p10 %.2f  p50 %.2f  p90 %.2f  max %.2f us\n", pct(endt, .1), pct(endt, .5), pct(endt, .9), pct(endt, 1));
  CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(y)); CK(hipFree(dbg));
}

int main(int argc, char** argv) {
  if (argc > 1) {   // pad sweep on the K3 shapes
    for (int i = 1; i < argc; ++i) {
      const int pad = atoi(argv[i]);
      run<3>("3x3 64->64 at 128x256 (K3 stage 1)", 128, 256, 64, 64, pad);
      run<3>("3x3 128->128 at 64x128 (K3 stage 2)", 64, 128, 128, 128, pad);
      run<1>("1x1 256->64 at 128x256 (conv1 stage 1)", 128, 256, 256, 64, pad);
      run<3, true>("3x3 128->128 at 64x128 (K3 stage 2), paired loop", 64, 128, 128, 128, pad);
      run<3, true>("3x3 64->64 at 128x256 (K3 stage 1), paired loop", 128, 256, 64, 64, pad);
    }
    return 0;
  }
  run<3>("3x3 64->64 at 128x256 (K3 stage 1)", 128, 256, 64, 64);
  run<3>("3x3 128->128 at 64x128 (K3 stage 2)", 64, 128, 128, 128);
  run<1>("1x1 64->64 at 128x256", 128, 256, 64, 64);
  run<1>("1x1 256->64 at 128x256 (conv1 stage 1)", 128, 256, 256, 64);
  run<1>("1x1 64->256 at 128x256 (conv3 stage 1)", 128, 256, 64, 256);
  run<1>("1x1 128->512 at 64x128 (conv3 stage 2)", 64, 128, 128, 512);
  return 0;
}
