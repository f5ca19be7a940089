// Producer / consumer fusion hooks of the tiled MFMA engine (gfx950, bf16; round 5).
//
//   D[b][m][n] = epi( sum_k pro(A)[b][m][k] * B[b][n][k] )          both operands K-major, E (bf16) output      (prims.h: GemmFx)
//
// The late stages of the adapter stack (C = 384 .. 1024: 36 of the 48 adapter calls) run the backward of the gate / bottleneck chain
// as  elementwise launch -> GEMM -> elementwise launch.  Measured inside the step with the launches switched off one class at a
// time (dgsct_test_tune "skip", tools/call_overlap.py; round 5): relu_bwd_scale 1.1 ms, bn_bwd_apply 0.8 ms, xc_bwd 0.65 ms of the
// 51.7 ms step -- the forward-side candidates (scale_cols + rowdot 0.15, bn_stats + affine 0.2) are not worth a kernel.  Here:
//   * the transform of the A operand (ReLU mask x row scale x per-frame column scale; BatchNorm backward of two tensors) is applied to
//     the 16-byte chunks of the operand tile on their way from registers to LDS -- frame = row / rpf, the per-frame vectors of the
//     (<= 5) frames a tile touches sit in LDS as one fp32 table, so the weights stay frame-independent and stream like any GEMM's;
//     the transformed operand is written once (by the n-tile-0 workgroups) for the weight-gradient product on the aux stream;
//   * the channel-gate backward (dX1 += dXc (1 + ch_b), dch_b += sum_n dXc X1) runs on the staged accumulator rows; the per-frame
//     column sums are collected per workgroup in LDS and leave as one fp32 atomic per (frame, column).
// Kernel skeleton = gemm.hip's FAST path (unconditional clamped 16-byte loads, two register sets in flight, [rows][64 + 8] LDS
// tiles, v_mfma_f32_32x32x16_bf16, XCD-chunked work list, rows through LDS as 16-byte stores).
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include "prims.h"
#include "device_util.h"
#include "err.h"
#include "gemm_int.h"

namespace dgsct {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 fx_bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float fx_f32x16_t;

constexpr int FX_BKT = 64;                 // k-tile depth
constexpr int FX_PITCH = (FX_BKT + 8) * 2; // LDS row pitch (bytes): conflict-free ds_read_b128 fragments

struct FxK {
  int M, N, K, batch;
  int tiles_m, tiles_n, kt_total, nfast;
  const char* A; long lda, a_bs;
  const char* B; long ldb, b_bs;
  char* D; long ldd, dbs;
  const float* bias_n; int act;
  const char* R; long ldr, rbs;
  // prologue
  int rpf, nframes, nb, kpad; unsigned rinv;          // frame = umulhi(m, rinv); nb = frame buckets a tile can touch; kpad = K rounded up to 64
  const float* a_rs; const char* a_cs; int a_cs_dt; long a_cs_ld; const float* a_cs2; float a_scale;
  const char* a2; const float* bn_mean; const float* bn_rstd; const float* bn_sc; const float* bn_sh; const float* bn_sums;
  float bn_inv; int bn_C, bn_relu, bn_training;
  char* a_store;
  // epilogue
  const float* e_cs; const char* e_x; float* e_acc; long e_ld;
  float* e_acc2; float e_scale;          // COLSTATS: sum of squares; COLSUM: positive counts, scale of the sums
};

__device__ __forceinline__ int fx_frame(int m, unsigned rinv) { return (int)__umulhi((unsigned)m, rinv); }

__device__ __forceinline__ void fx_unpack(const uint4& r, float (&v)[8]) {
  const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
}
__device__ __forceinline__ uint4 fx_pack(const float (&v)[8]) {
  return make_uint4(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7]));
}

template <int WGM, int WGN, int TM, int TN, int PRO, int EPI>
__global__ __launch_bounds__(256, (TM * TN >= 4 || PRO == APRO_BNBWD) ? 2 : 3)
void gemm_fx_kernel(const FxK p) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  constexpr int NLA = BM / 32, NLB = BN / 32;               // 16-byte chunks per thread and k-tile (chunk = row r_i, k offset kq)
  constexpr int A_BYTES = BM * FX_PITCH, B_BYTES = BN * FX_PITCH;
  constexpr int SP = TN * 32 + 4;                           // epilogue staging pitch (floats)
  constexpr int STG_BYTES = 4 * 32 * SP * 4;
  constexpr int MAIN_BYTES = (A_BYTES + B_BYTES) > STG_BYTES ? (A_BYTES + B_BYTES) : STG_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsA = smem;
  char* ldsB = smem + A_BYTES;
  float* tab = reinterpret_cast<float*>(smem + MAIN_BYTES);                   // prologue table
  const int tab_floats = PRO == APRO_MASKSCALE ? p.nb * p.kpad : (PRO == APRO_BNBWD ? 5 * p.kpad : 0);
  float* chacc = tab + tab_floats;                                            // epilogue column sums: [nb][NQ][BN] (NQ = 2 but for EPI_XCBWD)
  constexpr int NQ = (EPI == EPI_COLSTATS || EPI == EPI_COLSUM) ? 2 : 1;
  constexpr bool COLRED = EPI != EPI_NONE;                                    // the epilogue ends in a column reduction

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  // XCD-chunked work list over (batch, tile): see gemm.hip
  int tm, tn, b;
  {
    const unsigned ntile = (unsigned)(p.tiles_m * p.tiles_n);
    const unsigned total = gridDim.x, L = blockIdx.x;
    const unsigned q = total >> 3, r = total & 7, x = L & 7, y = L >> 3;
    const unsigned pp = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    b = __builtin_amdgcn_readfirstlane((int)(pp / ntile));
    const int t = (int)(pp - (unsigned)b * ntile);
    if (p.nfast) { tm = __builtin_amdgcn_readfirstlane(t / p.tiles_n); tn = __builtin_amdgcn_readfirstlane(t - tm * p.tiles_n); }
    else { tn = __builtin_amdgcn_readfirstlane(t / p.tiles_m); tm = __builtin_amdgcn_readfirstlane(t - tn * p.tiles_m); }
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int f0 = (PRO == APRO_MASKSCALE || EPI == EPI_XCBWD || EPI == EPI_COLSUM) ? fx_frame(m0, p.rinv) : 0;
  const char* Ab = p.A + (long)b * p.a_bs * 2;
  const char* A2b = PRO == APRO_BNBWD ? p.a2 + (long)b * p.a_bs * 2 : nullptr;
  const char* Bb = p.B + (long)b * p.b_bs * 2;

  // ---- per-thread chunk geometry: rows r_i = (tid >> 3) + 32 i, k offset kq inside the k-tile
  const int kq = (tid & 7) * 8, rq = tid >> 3;
  long offA[NLA], offB[NLB];
  bool okA[NLA], okB[NLB];
#pragma unroll
  for (int i = 0; i < NLA; ++i) { const int m = m0 + rq + 32 * i; okA[i] = m < p.M; offA[i] = (long)(okA[i] ? m : p.M - 1) * p.lda; }
#pragma unroll
  for (int i = 0; i < NLB; ++i) { const int n = n0 + rq + 32 * i; okB[i] = n < p.N; offB[i] = (long)(okB[i] ? n : p.N - 1) * p.ldb; }

  // ---- prologue tables
  float rsv[NLA];                 // MASKSCALE: row factor of chunk i
  int ftab[NLA];                  // MASKSCALE: table row (frame - f0) * kpad of chunk i
  if constexpr (PRO == APRO_MASKSCALE) {
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      int m = m0 + rq + 32 * i; m = m < p.M ? m : p.M - 1;
      rsv[i] = (p.a_rs ? p.a_rs[m] : 1.f) * p.a_scale;
      ftab[i] = (fx_frame(m, p.rinv) - f0) * p.kpad;
    }
    for (int idx = tid; idx < p.nb * p.kpad; idx += 256) {
      const int f = idx / p.kpad, k = idx - f * p.kpad;
      const int fr = f0 + f;
      float v = 0.f;
      if (fr < p.nframes && k < p.K) {
        const long o = (long)fr * p.a_cs_ld + k;
        v = p.a_cs_dt == DT_F32 ? reinterpret_cast<const float*>(p.a_cs)[o] : bf2f(reinterpret_cast<const unsigned short*>(p.a_cs)[o]);
        if (p.a_cs2) v *= p.a_cs2[k];
      }
      tab[idx] = v;
    }
  }
  if constexpr (PRO == APRO_BNBWD) {
    // per channel c = b * K + k:  k1 | k2 | k3 | sc | sh   (bn_bwd_apply_k's arithmetic, prims_strip.hip)
    for (int k = tid; k < p.kpad; k += 256) {
      float k1 = 0.f, k2 = 0.f, k3 = 0.f, a = 0.f, sh = 0.f;
      if (k < p.K) {
        const int c = b * p.K + k;
        a = p.bn_sc[c]; sh = p.bn_sh[c]; k1 = a;
        if (p.bn_training) { k3 = a * p.bn_rstd[c] * p.bn_sums[p.bn_C + c] * p.bn_inv; k2 = a * p.bn_sums[c] * p.bn_inv - p.bn_mean[c] * k3; }
      }
      tab[k] = k1; tab[p.kpad + k] = k2; tab[2 * p.kpad + k] = k3; tab[3 * p.kpad + k] = a; tab[4 * p.kpad + k] = sh;
    }
  }
  if constexpr (COLRED) {
    for (int idx = tid; idx < p.nb * NQ * BN; idx += 256) chacc[idx] = 0.f;
  }
  // (the first __syncthreads of the k-loop orders these LDS writes before their first use)

  fx_f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  constexpr int NL2 = PRO == APRO_BNBWD ? NLA : 1;
  uint4 ra0[NLA], rb0[NLB], ra1[NLA], rb1[NLB], rx0[NL2], rx1[NL2];
  auto prefetch = [&](uint4 (&ra)[NLA], uint4 (&rb)[NLB], uint4 (&rx)[NL2], int kt) {
    int kf = kt * FX_BKT + kq;
    kf = kf < p.K ? kf : p.K - 8;                           // unconditional, clamped (masked at the LDS store)
#pragma unroll
    for (int i = 0; i < NLA; ++i) ra[i] = *reinterpret_cast<const uint4*>(Ab + (offA[i] + kf) * 2);
    if constexpr (PRO == APRO_BNBWD) {
#pragma unroll
      for (int i = 0; i < NLA; ++i) rx[i] = *reinterpret_cast<const uint4*>(A2b + (offA[i] + kf) * 2);
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) rb[i] = *reinterpret_cast<const uint4*>(Bb + (offB[i] + kf) * 2);
  };
  auto store = [&](const uint4 (&ra)[NLA], const uint4 (&rb)[NLB], const uint4 (&rx)[NL2], int kt) {
    const int kf = kt * FX_BKT + kq;
    const bool kok = kf < p.K;
    const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      uint4 v = ra[i];
      if constexpr (PRO == APRO_MASKSCALE) {
        const float4 t0 = *reinterpret_cast<const float4*>(tab + ftab[i] + kf);
        const float4 t1 = *reinterpret_cast<const float4*>(tab + ftab[i] + kf + 4);
        const float s = rsv[i];
        const uint4 w = make_uint4(f2bf2(s * t0.x, s * t0.y), f2bf2(s * t0.z, s * t0.w), f2bf2(s * t1.x, s * t1.y), f2bf2(s * t1.z, s * t1.w));
        // A holds ReLU outputs (>= 0, no NaNs): element > 0  <=>  its magnitude bits are not all zero.  Per 16-bit half: m = min(bits & 0x7fff, 1)
        // is 1 or 0 and w * m keeps or clears the scaled value -- three packed integer instructions per PAIR of elements
        // (inline asm: hipcc rewrites the vector-type formulation into per-half compares + selects + byte permutes, 3 x the instructions)
        auto sel = [](unsigned x, unsigned y) {
          unsigned m, r;
          asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(x & 0x7fff7fffu), "v"(0x00010001u));
          asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(y), "v"(m));
          return r;
        };
        v = make_uint4(sel(v.x, w.x), sel(v.y, w.y), sel(v.z, w.z), sel(v.w, w.w));
      }
      if constexpr (PRO == APRO_BNBWD) {
        float g[8], x[8], o[8];
        fx_unpack(ra[i], g);
        fx_unpack(rx[i], x);
        const float* t = tab + (kok ? kf : 0);
        const float4 k1a = *reinterpret_cast<const float4*>(t), k1b = *reinterpret_cast<const float4*>(t + 4);
        const float4 k2a = *reinterpret_cast<const float4*>(t + p.kpad), k2b = *reinterpret_cast<const float4*>(t + p.kpad + 4);
        const float4 k3a = *reinterpret_cast<const float4*>(t + 2 * p.kpad), k3b = *reinterpret_cast<const float4*>(t + 2 * p.kpad + 4);
        const float k1[8] = {k1a.x, k1a.y, k1a.z, k1a.w, k1b.x, k1b.y, k1b.z, k1b.w};
        const float k2[8] = {k2a.x, k2a.y, k2a.z, k2a.w, k2b.x, k2b.y, k2b.z, k2b.w};
        const float k3[8] = {k3a.x, k3a.y, k3a.z, k3a.w, k3b.x, k3b.y, k3b.z, k3b.w};
        if (p.bn_relu) {
          const float4 sa = *reinterpret_cast<const float4*>(t + 3 * p.kpad), sb = *reinterpret_cast<const float4*>(t + 3 * p.kpad + 4);
          const float4 ha = *reinterpret_cast<const float4*>(t + 4 * p.kpad), hb = *reinterpret_cast<const float4*>(t + 4 * p.kpad + 4);
          const float sc[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
          const float sh[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = (x[e] * sc[e] + sh[e] > 0.f) ? g[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = k1[e] * g[e] - k2[e] - x[e] * k3[e];
        v = fx_pack(o);
      }
      const bool ok = okA[i] && kok;
      if (!ok) v = z;
      *reinterpret_cast<uint4*>(ldsA + (rq + 32 * i) * FX_PITCH + kq * 2) = v;
      if constexpr (PRO != APRO_NONE) {
        if (p.a_store && tn == 0 && ok) *reinterpret_cast<uint4*>(p.a_store + ((long)b * p.a_bs + offA[i] + kf) * 2) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      uint4 v = rb[i];
      if (!(okB[i] && kok)) v = z;
      *reinterpret_cast<uint4*>(ldsB + (rq + 32 * i) * FX_PITCH + kq * 2) = v;
    }
  };
  auto compute = [&]() {
#pragma unroll
    for (int kk = 0; kk < FX_BKT / 16; ++kk) {
      fx_bf16x8_t af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const fx_bf16x8_t*>(ldsA + ((wm * TM + i) * 32 + (lane & 31)) * FX_PITCH + (kk * 16 + (lane >> 5) * 8) * 2);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[j] = *reinterpret_cast<const fx_bf16x8_t*>(ldsB + ((wn * TN + j) * 32 + (lane & 31)) * FX_PITCH + (kk * 16 + (lane >> 5) * 8) * 2);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };
  const int nk = p.kt_total;
  if (nk > 0) prefetch(ra0, rb0, rx0, 0);
  if (nk > 1) prefetch(ra1, rb1, rx1, 1);
  for (int kt = 0; kt < nk; kt += 2) {
    __syncthreads();
    store(ra0, rb0, rx0, kt);
    __syncthreads();
    if (kt + 2 < nk) prefetch(ra0, rb0, rx0, kt + 2);
    compute();
    if (kt + 1 < nk) {
      __syncthreads();
      store(ra1, rb1, rx1, kt + 1);
      __syncthreads();
      if (kt + 3 < nk) prefetch(ra1, rb1, rx1, kt + 3);
      compute();
    }
  }

  // ---- epilogue: accumulator element r of tile (i, j): row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), col = lane & 31.
  // Each wave stages its 32 x (TN * 32) block through LDS (fp32) and leaves token rows as 16-byte stores.
  constexpr int CPR = TN * 4;                               // 8-column chunks per staged row
  constexpr int NIT = (32 * CPR) / 64;
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * SP);
  char* Db = p.D + (long)b * p.dbs * 2;
  const char* Rb = p.R ? p.R + (long)b * p.rbs * 2 : nullptr;
  const int ncol0 = n0 + wn * TN * 32;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int mrow0 = m0 + (wm * TM + i) * 32;
    // loads of this block's epilogue up front, unconditionally, from clamped indices
    uint4 rpre[NIT], xpre[EPI == EPI_XCBWD ? NIT : 1];
    float4 cpre[EPI == EPI_XCBWD ? NIT : 1][2];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 64 + lane;
      int m = mrow0 + c / CPR, n = ncol0 + (c % CPR) * 8;
      m = m < p.M ? m : p.M - 1;
      n = n + 8 <= p.N ? n : (p.N >= 8 ? p.N - 8 : 0);
      if (Rb) rpre[it] = *reinterpret_cast<const uint4*>(Rb + ((long)m * p.ldr + n) * 2);
      if constexpr (EPI == EPI_XCBWD) {
        xpre[it] = *reinterpret_cast<const uint4*>(p.e_x + ((long)m * p.ldd + n) * 2);
        const float* cs = p.e_cs + (long)fx_frame(m, p.rinv) * p.e_ld + n;
        cpre[it][0] = *reinterpret_cast<const float4*>(cs);
        cpre[it][1] = *reinterpret_cast<const float4*>(cs + 4);
      }
    }
    __syncthreads();                                        // operand tiles / previous block fully consumed
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = ncol0 + j * 32 + (lane & 31);
      const float bn = (p.bias_n && n < p.N) ? p.bias_n[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float av = acc[i][j][r] + bn;
        if (p.act == ACT_RELU) av = fmaxf(av, 0.f);
        stg[row * SP + j * 32 + (lane & 31)] = av;
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 64 + lane;
      const int row = c / CPR, cc = c % CPR;
      const int m = mrow0 + row, n = ncol0 + cc * 8;
      const bool ok = m < p.M && n + 8 <= p.N;              // (N is a multiple of 8: host-checked)
      float v[8];
      {
        const float4 a = *reinterpret_cast<const float4*>(stg + row * SP + cc * 8);
        const float4 c4 = *reinterpret_cast<const float4*>(stg + row * SP + cc * 8 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c4.x; v[5] = c4.y; v[6] = c4.z; v[7] = c4.w;
      }
      if constexpr (EPI == EPI_XCBWD) {
        // v = E(acc) (what the unfused path stored as dXc); D = R + v (1 + ch); products v * X1 back into the staging block
        float x1[8], rv[8], pr[8];
        fx_unpack(xpre[it], x1);
        fx_unpack(rpre[it], rv);
        const float cs[8] = {cpre[it][0].x, cpre[it][0].y, cpre[it][0].z, cpre[it][0].w, cpre[it][1].x, cpre[it][1].y, cpre[it][1].z, cpre[it][1].w};
        float vr[8];
        {
          const uint4 w = fx_pack(v);
          fx_unpack(w, vr);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { pr[e] = ok ? vr[e] * x1[e] : 0.f; v[e] = rv[e] + vr[e] * (1.f + cs[e]); }
        *reinterpret_cast<float4*>(stg + row * SP + cc * 8) = make_float4(pr[0], pr[1], pr[2], pr[3]);
        *reinterpret_cast<float4*>(stg + row * SP + cc * 8 + 4) = make_float4(pr[4], pr[5], pr[6], pr[7]);
      } else if (Rb) {
        float rv[8];
        fx_unpack(rpre[it], rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      const uint4 packed = fx_pack(v);
      if (ok) *reinterpret_cast<uint4*>(Db + ((long)m * p.ldd + n) * 2) = packed;
      if constexpr (EPI == EPI_COLSTATS || EPI == EPI_COLSUM) {          // the values AS STORED back into the staging block (0 for rows / columns outside)
        float vr[8];
        fx_unpack(packed, vr);
#pragma unroll
        for (int e = 0; e < 8; ++e) vr[e] = ok ? vr[e] : 0.f;
        *reinterpret_cast<float4*>(stg + row * SP + cc * 8) = make_float4(vr[0], vr[1], vr[2], vr[3]);
        *reinterpret_cast<float4*>(stg + row * SP + cc * 8 + 4) = make_float4(vr[4], vr[5], vr[6], vr[7]);
      }
    }
    if constexpr (COLRED) {
      __syncthreads();
      // column reduction over this block's rows, split at the (single: rpf >= 32) frame boundary inside the block.
      //   XCBWD: sum of the products;  COLSUM: sum and number of positive entries (per frame);  COLSTATS: sum and sum of squares (no frames)
      constexpr int NCOL = TN * 32, LPC = 64 / NCOL, RPL = 32 / LPC;       // lanes per column, rows per lane
      const int col = lane % NCOL, half = lane / NCOL;
      int nb1 = 32, bk = 0;
      if constexpr (EPI != EPI_COLSTATS) {
        const int fb = fx_frame(mrow0 < p.M ? mrow0 : p.M - 1, p.rinv);
        nb1 = (fb + 1) * p.rpf - mrow0;                                   // rows of the block that belong to frame fb
        bk = fb - f0;
      }
      float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
      for (int rr = 0; rr < RPL; ++rr) {
        const int row = half * RPL + rr;
        const float x = stg[row * SP + col];
        const float y = EPI == EPI_COLSTATS ? x * x : (x > 0.f ? 1.f : 0.f);
        if (row < nb1) { s0 += x; q0 += y; } else { s1 += x; q1 += y; }
      }
      float* dst = chacc + (bk * NQ) * BN + wn * NCOL + col;
      atomicAdd(dst, s0);
      if (NQ == 2) atomicAdd(dst + BN, q0);
      if (nb1 < 32 && bk + 1 < p.nb) {
        atomicAdd(dst + NQ * BN, s1);
        if (NQ == 2) atomicAdd(dst + NQ * BN + BN, q1);
      }
    }
  }
  if constexpr (COLRED) {
    __syncthreads();
    for (int idx = tid; idx < p.nb * NQ * BN; idx += 256) {
      const int f = idx / (NQ * BN), rem = idx - f * (NQ * BN), q = rem / BN, cn = rem - q * BN;
      const int fr = f0 + f, n = n0 + cn;
      const float sv = chacc[idx];
      if (n >= p.N || sv == 0.f) continue;
      if constexpr (EPI == EPI_COLSTATS) {                                // channel of column n in group b: b * N + n
        unsafeAtomicAdd((q ? p.e_acc2 : p.e_acc) + (long)b * p.N + n, sv);
      } else {
        if (fr >= p.nframes) continue;
        if (q == 0) unsafeAtomicAdd(p.e_acc + (long)fr * p.e_ld + n, EPI == EPI_COLSUM ? sv * p.e_scale : sv);
        else unsafeAtomicAdd(p.e_acc2 + (long)fr * p.e_ld + n, sv);
      }
    }
  }
}

std::atomic<int> g_fx_mode{-1};

template <int WGM, int WGN, int TM, int TN, int PRO, int EPI>
void fx_launch_one(const FxK& k, dim3 grid, size_t shmem, hipStream_t s) {
  auto kern = gemm_fx_kernel<WGM, WGN, TM, TN, PRO, EPI>;
  if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  hipLaunchKernelGGL(kern, grid, dim3(256), shmem, s, k);
}
template <int WGM, int WGN, int TM, int TN>
void fx_launch_cfg(const FxK& k, int pro, int epi, dim3 grid, size_t shmem, hipStream_t s) {
  if (pro == APRO_MASKSCALE && epi == EPI_XCBWD) fx_launch_one<WGM, WGN, TM, TN, APRO_MASKSCALE, EPI_XCBWD>(k, grid, shmem, s);
  else if (pro == APRO_MASKSCALE) fx_launch_one<WGM, WGN, TM, TN, APRO_MASKSCALE, EPI_NONE>(k, grid, shmem, s);
  else if (pro == APRO_BNBWD) fx_launch_one<WGM, WGN, TM, TN, APRO_BNBWD, EPI_NONE>(k, grid, shmem, s);
  else if (epi == EPI_XCBWD) fx_launch_one<WGM, WGN, TM, TN, APRO_NONE, EPI_XCBWD>(k, grid, shmem, s);
  else if (epi == EPI_COLSTATS) fx_launch_one<WGM, WGN, TM, TN, APRO_NONE, EPI_COLSTATS>(k, grid, shmem, s);
  else if (epi == EPI_COLSUM) fx_launch_one<WGM, WGN, TM, TN, APRO_NONE, EPI_COLSUM>(k, grid, shmem, s);
  else fx_launch_one<WGM, WGN, TM, TN, APRO_NONE, EPI_NONE>(k, grid, shmem, s);
}

// tile configuration: 0 = 128 x 128, 1 = 64 x 64, 2 = 128 x 32, 3 = 128 x 64
int fx_cfg(const Gemm& g) {
  if (g.N <= 32) return 2;
  if (g.N <= 64) return 3;
  const long w0 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * g.batch;
  const long last = w0 % 512;
  const bool fills = w0 >= 192 && (last == 0 || last >= 384 || w0 >= 4096);
  return (g.K > 256 && g.M >= 128 && g.N >= 128 && fills) ? 0 : 1;
}
constexpr int FX_BM[4] = {128, 64, 128, 128}, FX_BN[4] = {128, 64, 32, 64};

size_t fx_shmem(int cfg, const Gemm& g, const GemmFx& fx, int* nb_out, int* kpad_out) {
  const int BM = FX_BM[cfg], BN = FX_BN[cfg];
  const int TNv = cfg == 0 ? 2 : (cfg == 3 ? 2 : 1);
  const size_t opnd = (size_t)(BM + BN) * FX_PITCH, stg = (size_t)4 * 32 * (TNv * 32 + 4) * 4;
  const int kpad = (g.K + 63) / 64 * 64;
  int nb = 1;
  if (fx.rpf > 0) nb = (BM + fx.rpf - 2) / fx.rpf + 1;
  size_t extra = 0;
  if (fx.a_pro == APRO_MASKSCALE) extra += (size_t)nb * kpad * 4;
  if (fx.a_pro == APRO_BNBWD) extra += (size_t)5 * kpad * 4;
  if (fx.epi == EPI_XCBWD) extra += (size_t)nb * BN * 4;
  if (fx.epi == EPI_COLSTATS || fx.epi == EPI_COLSUM) extra += (size_t)nb * 2 * BN * 4;
  if (nb_out) *nb_out = nb;
  if (kpad_out) *kpad_out = kpad;
  return (opnd > stg ? opnd : stg) + extra;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int gemmfx_mode(int set) {
  // bit mask of the call sites (plan.cpp): 1 dZ (BN2 backward), 2 dX3 (ReLU + BN1 backward), 4 dXc (ReLU backward + channel-gate backward),
  // 8 dX1 += dvq1 Wv1 (ReLU backward); 15 = all (default); 16: the A-operand prologues of sites 4 / 8 at every width (default: C <= 256);
  // 32: site 4 never with its prologue (tests); FORWARD sites: 64 vq1 product + per-frame column sums / positive counts, 128 the two
  // bottleneck products + their BatchNorm sums; default = 15 + 64 + 128
  if (g_fx_mode.load(std::memory_order_relaxed) < 0) g_fx_mode.store(getenv("DGSCT_NO_GEMMFX") ? 0 : 15 + 64 + 128, std::memory_order_relaxed);
  const int old = g_fx_mode.load(std::memory_order_relaxed);
  if (set >= 0) g_fx_mode.store(set & 255, std::memory_order_relaxed);
  return old;
}

bool gemm_fx_supported(const Ctx& ctx, const Gemm& g, const GemmFx& fx) {
  if (!gemmfx_mode(-1) || ctx.mode != DT_BF16) return false;
  if (!g.A.kmajor || !g.B.kmajor || g.KB != 1 || g.atomic || g.splitk > 1 || g.ddt != DT_BF16) return false;
  if ((g.act != ACT_NONE && g.act != ACT_RELU) || g.mask || g.R2 || g.bias_m || g.r1_m || g.r1_n || g.alpha_ptr || g.alpha != 1.f || g.sm_scale || g.sm_dot) return false;
  if (g.bias_n_bs != 0 || (g.R && (g.rdt != DT_BF16 || g.beta != 1.f))) return false;
  if (g.M < 8 || g.N < 8 || g.K < 8 || g.K % 8 || g.N % 8) return false;
  if (!al16(g.A.p) || !al16(g.B.p) || !al16(g.D) || g.A.ld % 8 || g.B.ld % 8 || g.ldd % 8 || g.A.bs % 8 || g.B.bs % 8 || g.dbs % 8) return false;
  if (g.R && (!al16(g.R) || g.ldr % 8 || g.rbs % 8)) return false;
  if (g.A.kbs || g.B.kbs) return false;
  const bool frames = fx.a_pro == APRO_MASKSCALE || fx.epi == EPI_XCBWD || fx.epi == EPI_COLSUM;
  if (frames) {
    if (fx.rpf < 32 || g.batch != 1 || g.M % fx.rpf) return false;          // a 32-row block touches <= 2 frames; whole frames
    if ((unsigned long long)g.M * (unsigned long long)fx.rpf >= 0x100000000ULL) return false;    // frame split by multiply-high
  }
  if (fx.a_pro == APRO_MASKSCALE) {
    if (!fx.a_cs || fx.a_cs_ld < g.K) return false;
    if (fx.a_store && (!al16(fx.a_store) || fx.a_store == g.A.p)) return false;     // (in place: other n-tiles still read A)
  }
  if (fx.a_pro == APRO_BNBWD) {
    if (!fx.a2 || !al16(fx.a2) || !fx.bn_sc || !fx.bn_sh || fx.bn_C < g.K * g.batch) return false;
    if (fx.bn_training && (!fx.bn_mean || !fx.bn_rstd || !fx.bn_sums || fx.bn_rows <= 0)) return false;
    if (fx.a_store && !al16(fx.a_store)) return false;
    if (fx.a_store == g.A.p && g.N > FX_BN[fx_cfg(g)]) return false;                // in place only when ONE n-tile reads every A element
  }
  if (fx.epi == EPI_COLSTATS && (!fx.e_acc || !fx.e_acc2 || fx.a_pro != APRO_NONE || g.R)) return false;
  if (fx.epi == EPI_COLSUM && (!fx.e_acc || !fx.e_acc2 || fx.e_ld < g.N || fx.a_pro != APRO_NONE || g.R)) return false;
  if (fx.epi == EPI_XCBWD) {
    if (!g.R || g.ldr != g.ldd || g.rbs != g.dbs || !fx.e_cs || !fx.e_x || !fx.e_acc || !al16(fx.e_x) || !al16(fx.e_cs) || fx.e_ld % 4 || fx.e_ld < g.N) return false;
  }
  int nb = 1, kpad = 0;
  if (fx_shmem(fx_cfg(g), g, fx, &nb, &kpad) > 96 * 1024) return false;
  return true;
}

void gemm_fx(const Ctx& ctx, const Gemm& g, const GemmFx& fx) {
  if (!gemm_fx_supported(ctx, g, fx)) { set_error("gemm_fx: unsupported call (the plan must test gemm_fx_supported first)"); return; }
  const int cfg = fx_cfg(g);
  FxK k{};
  k.M = g.M; k.N = g.N; k.K = g.K; k.batch = g.batch;
  k.tiles_m = (g.M + FX_BM[cfg] - 1) / FX_BM[cfg];
  k.tiles_n = (g.N + FX_BN[cfg] - 1) / FX_BN[cfg];
  k.kt_total = (g.K + FX_BKT - 1) / FX_BKT;
  k.nfast = k.tiles_n < k.tiles_m;
  k.A = (const char*)g.A.p; k.lda = g.A.ld; k.a_bs = g.A.bs;
  k.B = (const char*)g.B.p; k.ldb = g.B.ld; k.b_bs = g.B.bs;
  k.D = (char*)g.D; k.ldd = g.ldd; k.dbs = g.dbs;
  k.bias_n = g.bias_n; k.act = g.act;
  k.R = (const char*)g.R; k.ldr = g.ldr; k.rbs = g.rbs;
  k.rpf = fx.rpf > 0 ? fx.rpf : g.M;
  k.nframes = fx.rpf > 0 ? g.M / fx.rpf : 1;
  k.rinv = fx.rpf > 0 ? (unsigned)((0x100000000ULL + (unsigned long long)fx.rpf - 1) / (unsigned long long)fx.rpf) : 0u;
  int nb = 1, kpad = 0;
  const size_t shmem = fx_shmem(cfg, g, fx, &nb, &kpad);
  k.nb = nb; k.kpad = kpad;
  k.a_rs = fx.a_rs; k.a_cs = (const char*)fx.a_cs; k.a_cs_dt = fx.a_cs_dt; k.a_cs_ld = fx.a_cs_ld; k.a_cs2 = fx.a_cs2; k.a_scale = fx.a_scale;
  k.a2 = (const char*)fx.a2; k.bn_mean = fx.bn_mean; k.bn_rstd = fx.bn_rstd; k.bn_sc = fx.bn_sc; k.bn_sh = fx.bn_sh; k.bn_sums = fx.bn_sums;
  k.bn_inv = fx.bn_rows > 0 ? 1.f / (float)fx.bn_rows : 0.f; k.bn_C = fx.bn_C; k.bn_relu = fx.bn_relu; k.bn_training = fx.bn_training;
  k.a_store = (char*)fx.a_store;
  k.e_cs = fx.e_cs; k.e_x = (const char*)fx.e_x; k.e_acc = fx.e_acc; k.e_ld = fx.e_ld; k.e_acc2 = fx.e_acc2; k.e_scale = fx.e_scale;
  const dim3 grid((unsigned)(k.tiles_m * k.tiles_n * g.batch));
  hipStream_t s = (hipStream_t)ctx.stream;
  GemmProfShape shp{g.M, g.N, g.K, 1, g.batch, 1, 20 + cfg, 1, 1, 0, 1, 0.0};
  shp.bytes = ((double)g.M * g.K * (fx.a_pro == APRO_BNBWD ? 2 : 1) * g.batch + (double)g.N * g.K * (g.B.bs ? g.batch : 1)) * 2 +
              (double)g.M * g.N * g.batch * 2 * (g.R ? 2 : 1) * (fx.epi == EPI_XCBWD ? 1.5 : 1.0);
  void* rec = gemm_prof_begin(s, 2.0 * g.M * (double)g.N * (double)g.K * g.batch, shp);
  switch (cfg) {
    case 0: fx_launch_cfg<2, 2, 2, 2>(k, fx.a_pro, fx.epi, grid, shmem, s); break;
    case 1: fx_launch_cfg<2, 2, 1, 1>(k, fx.a_pro, fx.epi, grid, shmem, s); break;
    case 2: fx_launch_cfg<4, 1, 1, 1>(k, fx.a_pro, fx.epi, grid, shmem, s); break;
    default: fx_launch_cfg<4, 1, 1, 2>(k, fx.a_pro, fx.epi, grid, shmem, s); break;
  }
  gemm_prof_end(rec, s);
}

}  // namespace dgsct
