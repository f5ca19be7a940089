// Skinny MFMA GEMM for the per-frame gate MLPs of the adapter (gfx950, bf16 operands).
//
//   D[m][n] = epi( sum_k A[m][k] * B[n][k] )        M <= 256 rows (one per frame: BT = 160 at the benchmark), both operands K-major
//
// Every adapter call runs eight of these on or next to its dependency chain -- aq1 / aq2 / q / ch forward, dq / dm1 / da (x2)
// backward (reference net_trans.py:593-597 and their autograd): [BT, C] x [C, C] products whose cost is the WEIGHT read, not the
// math.  On the tiled engine (64 x 64 tiles, 3 x 8 = 24 workgroups, a two-barrier k-loop of 8-16 tiles) they took 9-18 us each,
// 6.7 ms of serial kernel time per step in 576 launches; even a 160 x 96 x 48 product took 9.7 us -- the loop's own latency.
//
// Here the unit of work is one 32 x 32 output tile per workgroup and the CONTRACTION is split over its four waves: wave w owns
// k in [w K/4, (w+1) K/4), loads its operand fragments straight from global memory / L2 into registers (16 bytes per lane per
// fragment, no LDS staging, no barrier in the loop, the next four k-steps already in flight while four are multiplied), and
// the four partial tiles meet once in LDS.  A workgroup reads (32 + 32) x K x 2 bytes -- 64 KB at K = 512 -- so the ~40 GB/s a CU
// can pull are spread over M/32 x N/32 = 40-160 CUs instead of 24.  Epilogue: bias, ReLU / sigmoid, ReLU mask of
// another tensor, residual; fp32 or bf16 out as 16- / 8-byte row pieces.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include "prims.h"
#include "device_util.h"
#include "err.h"
#include "gemm_int.h"

namespace dgsct {

typedef __attribute__((ext_vector_type(8))) __bf16 sk_bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float sk_f32x16_t;

struct SkArgs {
  int M, N, K;
  const unsigned short* A; long lda;
  const unsigned short* B; long ldb;
  char* D; int ddt; long ldd;
  const float* bias_n; int act;
  const char* R; int rdt; long ldr; float beta;
  const unsigned short* mask; long ldmask;
};

__global__ __launch_bounds__(256) void gemm_skinny_k(const SkArgs p) {
  __shared__ float part[4][32][33];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  // fragment of a 16-deep k-step: lane l holds X[row0 + (l & 31)][k0 + 8 (l >> 5) .. + 7]; rows past the edge are clamped (their
  // products land in accumulator rows / columns that are never stored)
  int ra = m0 + (lane & 31); ra = ra < p.M ? ra : p.M - 1;
  int rb = n0 + (lane & 31); rb = rb < p.N ? rb : p.N - 1;
  const int kq = p.K >> 2;                                    // this wave's share of the contraction (a multiple of 16)
  const unsigned short* pa = p.A + (long)ra * p.lda + wave * kq + 8 * (lane >> 5);
  const unsigned short* pb = p.B + (long)rb * p.ldb + wave * kq + 8 * (lane >> 5);
  const int nks = kq >> 4;

  sk_f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int CH = 4;                                       // k-steps per chunk; two chunks of loads in flight
  uint4 fa[2][CH], fb[2][CH];
  auto load = [&](int s, int ks0) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      int ks = ks0 + i; ks = ks < nks ? ks : nks - 1;         // unconditional, clamped; masked at the MFMA
      fa[s][i] = *reinterpret_cast<const uint4*>(pa + ks * 16);
      fb[s][i] = *reinterpret_cast<const uint4*>(pb + ks * 16);
    }
  };
  auto mma = [&](int s, int ks0) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      uint4 a = fa[s][i];
      if (ks0 + i >= nks) a = make_uint4(0, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sk_bf16x8_t, a), __builtin_bit_cast(sk_bf16x8_t, fb[s][i]), acc, 0, 0, 0);
    }
  };
  // straight-line body: every load is issued (k-steps past the end re-read the last one, their MFMAs get a zero operand) -- a
  // load under a condition makes hipcc wait for ALL outstanding loads at the merge point, i.e. the two chunks would not overlap
  load(0, 0);
  for (int ks0 = 0; ks0 < nks; ks0 += 2 * CH) {
    load(1, ks0 + CH);
    mma(0, ks0);
    load(0, ks0 + 2 * CH);
    mma(1, ks0 + CH);
  }
  // accumulator element r: row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), col = lane & 31
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][lane & 31] = acc[r];
  __syncthreads();
  // thread t finishes row t / 8, columns 4 (t % 8) .. + 3
  const int row = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
  const int m = m0 + row, n = n0 + c0;
  if (m >= p.M || n >= p.N) return;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = part[0][row][c0 + e] + part[1][row][c0 + e] + part[2][row][c0 + e] + part[3][row][c0 + e];
  const bool full = n + 4 <= p.N;
  // epilogue operands: unconditional loads from clamped columns (N % 4 == 0 is host-checked, so `full` holds for every stored piece)
  float bn[4] = {0.f, 0.f, 0.f, 0.f}, rv[4] = {0.f, 0.f, 0.f, 0.f}, mk[4] = {1.f, 1.f, 1.f, 1.f};
  if (p.bias_n) ldv<DT_F32, 4>(p.bias_n, n, bn);
  if (p.R) {
    if (p.rdt == DT_F32) ldv<DT_F32, 4>(p.R, (long)m * p.ldr + n, rv);
    else ldv<DT_BF16, 4>(p.R, (long)m * p.ldr + n, rv);
  }
  if (p.mask) ldv<DT_BF16, 4>(p.mask, (long)m * p.ldmask + n, mk);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float x = v[e] + bn[e];
    if (p.act == ACT_RELU) x = fmaxf(x, 0.f);
    else if (p.act == ACT_SIGMOID) x = 1.f / (1.f + __expf(-x));
    if (p.mask && !(mk[e] > 0.f)) x = 0.f;
    v[e] = x + p.beta * rv[e];
  }
  (void)full;
  if (p.ddt == DT_F32) stv<DT_F32, 4>(p.D, (long)m * p.ldd + n, v);
  else stv<DT_BF16, 4>(p.D, (long)m * p.ldd + n, v);
}

// ---- the same product with the neighbouring elementwise launches of the gate-MLP chain folded in (SkFuse, prims.h) -------------
// Every [BT, C] elementwise kernel next to these products was a 5-6 us slot on a dependency chain that is launch-latency-bound at
// stages 2-3 (tools/call_overlap.py: ~95 chain launches per adapter call, ~15 us per slot of a pair): the operand transform
// (m1 = aq1 * mean_N vq1;  dpre = dch * ch (1 - ch)) is applied to the A fragment as it arrives and stored once by the first column of
// workgroups, the two `da` products accumulate into one tile, and dm1's two consumers (dpa1, coef) are written from the epilogue.
template <int AMODE>
__device__ __forceinline__ uint4 sk_a_frag(const SkFuse& p, int row, int k) {
  if (AMODE == 0) return *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p.A) + (long)row * p.lda + k);
  float a[8], m[8];
  if (AMODE == 1) ldv<DT_BF16, 8>(p.A, (long)row * p.lda + k, a);
  else { ldv<DT_F32, 4>(p.A, (long)row * p.lda + k, *reinterpret_cast<float(*)[4]>(a)); ldv<DT_F32, 4>(p.A, (long)row * p.lda + k + 4, *reinterpret_cast<float(*)[4]>(a + 4)); }
  ldv<DT_F32, 4>(p.a_mul, (long)row * p.ld_mul + k, *reinterpret_cast<float(*)[4]>(m));
  ldv<DT_F32, 4>(p.a_mul, (long)row * p.ld_mul + k + 4, *reinterpret_cast<float(*)[4]>(m + 4));
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = AMODE == 1 ? a[e] * m[e] : a[e] * m[e] * (1.f - m[e]);
  uint4 r;
  r.x = f2bf2(a[0], a[1]); r.y = f2bf2(a[2], a[3]); r.z = f2bf2(a[4], a[5]); r.w = f2bf2(a[6], a[7]);
  return r;
}

template <int AMODE>
__global__ __launch_bounds__(256) void gemm_skinny_fused_k(const SkFuse p) {
  __shared__ float part[4][32][33];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  int ra = m0 + (lane & 31); const bool ra_ok = ra < p.M; ra = ra_ok ? ra : p.M - 1;
  int rb = n0 + (lane & 31); rb = rb < p.N ? rb : p.N - 1;
  sk_f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  {
    const int kq = p.K >> 2, nks = kq >> 4;
    const int kbase = wave * kq + 8 * (lane >> 5);
    const unsigned short* pb = reinterpret_cast<const unsigned short*>(p.B) + (long)rb * p.ldb + kbase;
    constexpr int CH = 2;
    uint4 fa[2][CH], fb[2][CH];
    auto load = [&](int s, int ks0) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        int ks = ks0 + i; ks = ks < nks ? ks : nks - 1;
        fa[s][i] = sk_a_frag<AMODE>(p, ra, kbase + ks * 16);
        fb[s][i] = *reinterpret_cast<const uint4*>(pb + ks * 16);
      }
    };
    auto mma = [&](int s, int ks0) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        uint4 a = fa[s][i];
        if (AMODE != 0 && p.a_store && blockIdx.x == 0 && ra_ok && ks0 + i < nks)      // the transformed operand, stored once
          *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p.a_store) + (long)ra * p.ld_store + kbase + (ks0 + i) * 16) = a;
        if (ks0 + i >= nks) a = make_uint4(0, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sk_bf16x8_t, a), __builtin_bit_cast(sk_bf16x8_t, fb[s][i]), acc, 0, 0, 0);
      }
    };
    load(0, 0);
    for (int ks0 = 0; ks0 < nks; ks0 += 2 * CH) {
      load(1, ks0 + CH);
      mma(0, ks0);
      load(0, ks0 + 2 * CH);
      mma(1, ks0 + CH);
    }
  }
  if (p.K2 > 0) {                                             // second product into the same tile (plain bf16 operands)
    const int kq = p.K2 >> 2, nks = kq >> 4;
    const unsigned short* pa = reinterpret_cast<const unsigned short*>(p.A2) + (long)ra * p.lda2 + wave * kq + 8 * (lane >> 5);
    const unsigned short* pb = reinterpret_cast<const unsigned short*>(p.B2) + (long)rb * p.ldb2 + wave * kq + 8 * (lane >> 5);
    for (int ks0 = 0; ks0 < nks; ks0 += 4) {
      uint4 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int ks = ks0 + i; ks = ks < nks ? ks : nks - 1;
        fa[i] = *reinterpret_cast<const uint4*>(pa + ks * 16);
        fb[i] = *reinterpret_cast<const uint4*>(pb + ks * 16);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint4 a = fa[i];
        if (ks0 + i >= nks) a = make_uint4(0, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sk_bf16x8_t, a), __builtin_bit_cast(sk_bf16x8_t, fb[i]), acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][lane & 31] = acc[r];
  __syncthreads();
  const int row = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
  const int m = m0 + row, n = n0 + c0;
  if (m >= p.M || n >= p.N) return;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = part[0][row][c0 + e] + part[1][row][c0 + e] + part[2][row][c0 + e] + part[3][row][c0 + e];
  float bn[4] = {0.f, 0.f, 0.f, 0.f}, mk[4] = {1.f, 1.f, 1.f, 1.f};
  if (p.bias_n) ldv<DT_F32, 4>(p.bias_n, n, bn);
  if (p.mask) ldv<DT_BF16, 4>(p.mask, (long)m * p.ldmask + n, mk);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float x = v[e] + bn[e];
    if (p.act == ACT_RELU) x = fmaxf(x, 0.f);
    else if (p.act == ACT_SIGMOID) x = 1.f / (1.f + __expf(-x));
    if (p.mask && !(mk[e] > 0.f)) x = 0.f;
    v[e] = x;
  }
  if (p.epi == 1) {                                           // D = v * e_mul * (e_q > 0)  (E);  D2 = v * e_q  (fp32)
    float em[4], eq[4], o1[4], o2[4];
    ldv<DT_F32, 4>(p.e_mul, (long)m * p.ld_emul + n, em);
    ldv<DT_BF16, 4>(p.e_q, (long)m * p.ld_eq + n, eq);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o1[e] = eq[e] > 0.f ? v[e] * em[e] : 0.f; o2[e] = v[e] * eq[e]; }
    stv<DT_BF16, 4>(p.D, (long)m * p.ldd + n, o1);
    stv<DT_F32, 4>(p.D2, (long)m * p.ldd2 + n, o2);
    return;
  }
  if (p.ddt == DT_F32) stv<DT_F32, 4>(p.D, (long)m * p.ldd + n, v);
  else stv<DT_BF16, 4>(p.D, (long)m * p.ldd + n, v);
}

int skfuse_mode(int set) {
  static std::atomic<int> mode{getenv("DGSCT_NO_SKFUSE") ? 0 : 1};
  const int old = mode.load(std::memory_order_relaxed);
  if (set >= 0) mode.store(set ? 1 : 0, std::memory_order_relaxed);
  return old;
}

bool skinny_fused_supported(const Ctx& ctx, int M, int N, int K, int K2) {
  if (!skfuse_mode(-1) || !gemm_skinny_mode(-1) || ctx.mode != DT_BF16) return false;
  if (M < 1 || M > 256 || N < 32 || N % 4) return false;
  if (K < 64 || K % 64 || K > 4096) return false;
  if (K2 && (K2 < 64 || K2 % 64 || K2 > 4096)) return false;
  return true;
}

void skinny_fused(const Ctx& ctx, const SkFuse& p) {
  auto al = [](const void* q, int a) { return (reinterpret_cast<uintptr_t>(q) & (uintptr_t)(a - 1)) == 0; };
  bool ok = skinny_fused_supported(ctx, p.M, p.N, p.K, p.K2) && p.b_kmajor && (!p.K2 || p.b2_kmajor) && al(p.A, 16) && al(p.B, 16) && p.lda % 8 == 0 && p.ldb % 8 == 0 && al(p.D, 8) &&
            p.ldd % 4 == 0;
  if (p.a_mode) ok = ok && al(p.a_mul, 16) && p.ld_mul % 4 == 0 && (!p.a_store || (al(p.a_store, 16) && p.ld_store % 8 == 0));
  if (p.K2) ok = ok && al(p.A2, 16) && al(p.B2, 16) && p.lda2 % 8 == 0 && p.ldb2 % 8 == 0;
  if (p.mask) ok = ok && al(p.mask, 8) && p.ldmask % 4 == 0;
  if (p.bias_n) ok = ok && al(p.bias_n, 16);
  if (p.epi == 1) ok = ok && al(p.e_mul, 16) && p.ld_emul % 4 == 0 && al(p.e_q, 8) && p.ld_eq % 4 == 0 && al(p.D2, 16) && p.ldd2 % 4 == 0;
  if (!ok) { set_error("skinny_fused: unsupported shape / alignment (check skinny_fused_supported first)"); return; }
  dim3 grid((p.N + 31) / 32, (p.M + 31) / 32);
  hipStream_t s = (hipStream_t)ctx.stream;
  GemmProfShape shp{p.M, p.N, p.K + p.K2, 1, 1, 1, 10, 1, 1, 0, 1, 0.0};
  shp.bytes = ((double)p.M * (p.K + p.K2) + (double)p.N * (p.K + p.K2)) * 2 + (double)p.M * p.N * 4;
  void* rec = gemm_prof_begin(s, 2.0 * p.M * (double)p.N * (p.K + p.K2), shp);
  if (p.a_mode == 0) hipLaunchKernelGGL(gemm_skinny_fused_k<0>, grid, dim3(256), 0, s, p);
  else if (p.a_mode == 1) hipLaunchKernelGGL(gemm_skinny_fused_k<1>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(gemm_skinny_fused_k<2>, grid, dim3(256), 0, s, p);
  gemm_prof_end(rec, s);
}

static inline bool sk_al(const void* p, int a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

int gemm_skinny_mode(int set) {     // process-wide, test-only switch (like gemm8_mode / rowfuse_mode): atomic, DataParallel replicas run on threads
  static std::atomic<int> mode{getenv("DGSCT_GEMM_SKINNY") ? atoi(getenv("DGSCT_GEMM_SKINNY")) : 1};
  const int old = mode.load(std::memory_order_relaxed);
  if (set >= 0) mode.store(set, std::memory_order_relaxed);
  return old;
}

bool gemm_skinny_try(const Ctx& ctx, const Gemm& g) {
  if (!gemm_skinny_mode(-1) || ctx.mode != DT_BF16) return false;
  if (g.batch != 1 || g.KB != 1 || g.atomic || g.splitk > 1) return false;
  if (!g.A.kmajor || !g.B.kmajor) return false;
  if (g.M < 1 || g.M > 256 || g.N < 32 || g.N % 4) return false;
  if (g.K < 64 || g.K % 64 || g.K > 4096) return false;
  if (g.act != ACT_NONE && g.act != ACT_RELU && g.act != ACT_SIGMOID) return false;
  if (g.alpha != 1.f || g.alpha_ptr || g.bias_m || g.r1_m || g.r1_n || g.R2 || g.sm_scale || g.sm_dot || g.bias_n_bs) return false;
  if (!sk_al(g.A.p, 16) || !sk_al(g.B.p, 16) || g.A.ld % 8 || g.B.ld % 8) return false;
  const int des = g.ddt == DT_F32 ? 4 : 2;
  if (!sk_al(g.D, 4 * des) || g.ldd % 4) return false;
  if (g.R && (!sk_al(g.R, g.rdt == DT_F32 ? 16 : 8) || g.ldr % 4)) return false;
  if (g.mask && (!sk_al(g.mask, 8) || g.ldmask % 4)) return false;
  if (g.bias_n && !sk_al(g.bias_n, 16)) return false;
  SkArgs a;
  a.M = g.M; a.N = g.N; a.K = g.K;
  a.A = (const unsigned short*)g.A.p; a.lda = g.A.ld;
  a.B = (const unsigned short*)g.B.p; a.ldb = g.B.ld;
  a.D = (char*)g.D; a.ddt = g.ddt; a.ldd = g.ldd;
  a.bias_n = g.bias_n; a.act = g.act;
  a.R = (const char*)g.R; a.rdt = g.rdt; a.ldr = g.ldr; a.beta = g.R ? g.beta : 0.f;
  a.mask = (const unsigned short*)g.mask; a.ldmask = g.ldmask;
  dim3 grid((g.N + 31) / 32, (g.M + 31) / 32);
  hipStream_t s = (hipStream_t)ctx.stream;
  GemmProfShape shp{g.M, g.N, g.K, 1, 1, 1, 10, 1, 1, 0, 1, 0.0};
  shp.bytes = ((double)g.M * g.K + (double)g.N * g.K) * 2 + (double)g.M * g.N * des;
  void* rec = gemm_prof_begin(s, 2.0 * g.M * (double)g.N * g.K, shp);
  hipLaunchKernelGGL(gemm_skinny_k, grid, dim3(256), 0, s, a);
  gemm_prof_end(rec, s);
  return true;
}

}  // namespace dgsct
