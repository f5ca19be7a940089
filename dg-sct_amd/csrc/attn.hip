// Fused (flash-style) latent-token attention of the DG-SCT adapter for gfx950 -- reference net_trans.py:572-589 and its
// autograd.  Four kernels replace 8 MFMA-GEMM launches with a 32-wide dimension, 4 softmax launches and 3 helpers:
//
//   tokattn_fwd : tok = T0 + softmax_N(T0 . Yp^T) . Yp, a = mean_N Yp      one pass over Yp, split over N (+ tiny combine)
//   xattn_fwd   : X1  = X + gate_av * softmax_tk(X . tok^T) . tok           one pass over X
//   xattn_bwd   : dX, dtok, dgate_av from dX1 (P2 recomputed)               one pass over X and dX1
//   tokattn_bwd : dYp, dT0 from dtok (P1 recomputed from the saved log-sum-exp)   one pass over Yp
//
// Nothing of size [tokens x latent tokens] ever reaches HBM (the logits, P1, P2, dS1, dS2 of the multi-launch schedule
// did, in fp32 and in E), and the logits are computed from fp32 latent tokens split into bf16 hi + lo operands: the
// softmaxes are UN-SCALED (logits ~ sqrt(C)), so operand rounding of `tok` / `my_tokens` is what they amplify.
//
// Structure of every kernel: a workgroup (4 wavefronts) owns 128 token rows of one frame.  Phase A streams 64-channel
// slabs of the token-major activation through LDS and accumulates the [tokens x 32] logits with one 32x32 MFMA tile per
// wave; the softmax runs along the accumulator registers (the latent-token / token axis is laid along registers by
// choosing which operand is "A"); the probabilities go to LDS in the element type.  Phase B streams the slabs again
// (L2 / MALL hits) for the products with the probabilities; token-major outputs are assembled in LDS and leave as
// 16-byte row stores.  Both arithmetic modes of the library share the code (mma_tile.h).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdlib>
#include "prims.h"
#include "device_util.h"
#include "mma_tile.h"
#include "err.h"

namespace dgsct {

namespace {
constexpr int NCH_ROWS = 128;      // token rows per workgroup
constexpr int MAX_CHUNKS = 64;     // chunks of one frame (N <= 8192)

__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }
// the fp32 mode is the parity path (1e-3 against the reference): accurate exp there, the hardware approximation in bf16
template <int MODE> __device__ __forceinline__ float mexp(float x) { return MODE == DT_F32 ? expf(x) : __expf(x); }

// wave-local copy of rows [row0, row0 + 32) x TC columns of an LDS image to a token-major global tensor (16-byte stores);
// `add` (optional, same layout as dst) is added on the way out.
template <int MODE, int TC>
__device__ __forceinline__ void copy_out_rows(const char* img, int pitch, int row0, void* dst, const void* add, long ld, long grow0,
                                              int rows_valid_from_row0, int c0, int cols_valid, int lane) {
  constexpr int ES = MT<MODE>::ES, VE = MT<MODE>::VE, CPR = TC / VE;
#pragma unroll 2
  for (int i = lane; i < 32 * CPR; i += 64) {
    const int r = i / CPR, c = (i % CPR) * VE;
    if (r >= rows_valid_from_row0 || c0 + c >= cols_valid) continue;
    uint4 v = *reinterpret_cast<const uint4*>(img + (row0 + r) * pitch + c * ES);
    const long o = ((grow0 + r) * ld + c0 + c) * ES;
    if (add) {
      float x[VE], y[VE];
      unpack<MODE, VE>(v, x);
      unpack<MODE, VE>(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(add) + o), y);
#pragma unroll
      for (int e = 0; e < VE; ++e) x[e] += y[e];
      stv<MODE, VE>(dst, o / ES, x);
    } else {
      *reinterpret_cast<uint4*>(reinterpret_cast<char*>(dst) + o) = v;
    }
  }
}
}  // namespace

// ====================================================================================================================
// tokattn_fwd: per (frame, 128-row chunk): m[t] = max_n S, l[t] = sum_n exp(S - m), O[t][:] = sum_n exp(S - m) Yp[n][:]
// ====================================================================================================================
struct TokFwdArgs {
  const void* Yp; const float* T0; int N, C, tk, nch;
  float* partO;      // [B][nch][32][C]
  float* partML;     // [B][nch][2][32]
  float* a;          // [B][C] += column sums (pre-zeroed)
};
template <int MODE>
__global__ __launch_bounds__(256) void tokattn_fwd_k(const TokFwdArgs p) {
  using M = MT<MODE>;
  constexpr int ES = M::ES, CSA = 64, CSB = 128;
  constexpr int PYA = M::km_pitch(CSA), PT = M::km_pitch(CSA), PYB = M::mn_pitch(CSB), PP = M::mn_pitch(32);
  constexpr int A_BYTES = NCH_ROWS * PYA + 2 * 32 * PT, B_BYTES = NCH_ROWS * PYB;
  __shared__ __attribute__((aligned(16))) char smem[(A_BYTES > B_BYTES ? A_BYTES : B_BYTES) + NCH_ROWS * PP];
  __shared__ float red[2][4][32];
  char* sY = smem;
  char* sTh = smem + NCH_ROWS * PYA;
  char* sTl = sTh + 32 * PT;
  char* sP = smem + (A_BYTES > B_BYTES ? A_BYTES : B_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk = blockIdx.x, b = blockIdx.y, n0 = chunk * NCH_ROWS;
  const char* Yb = reinterpret_cast<const char*>(p.Yp) + (long)b * p.N * p.C * ES;

  // ---- phase A: S^T[n][t] = Yp[n][:] . T0[t][:]   (rows n along the accumulator registers)
  mt_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int cs = 0; cs < p.C; cs += CSA) {
    __syncthreads();
    stage_tile<MODE, NCH_ROWS, CSA>(sY, PYA, Yb, p.C, n0, p.N, cs, p.C, tid);
    stage_tile_f32<MODE, 32, CSA, true>(sTh, sTl, PT, p.T0, p.C, 0, p.tk, cs, p.C, tid);
    __syncthreads();
    mma_tile<MODE, true, true>(acc, sY, PYA, 32 * wave, sTh, PT, 0, CSA, lane);
    if (MODE == DT_BF16) mma_tile<MODE, true, true>(acc, sY, PYA, 32 * wave, sTl, PT, 0, CSA, lane);
  }
  const int t = lane & 31;
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (n0 + 32 * wave + mt_row(r, lane) < p.N) mx = fmaxf(mx, acc[r]);
  mx = fmaxf(mx, xor32(mx));
  if (lane < 32) red[0][wave][t] = mx;
  __syncthreads();
  const float m = fmaxf(fmaxf(red[0][0][t], red[0][1][t]), fmaxf(red[0][2][t], red[0][3][t]));
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int nl = 32 * wave + mt_row(r, lane);
    const float pv = (n0 + nl < p.N) ? mexp<MODE>(acc[r] - m) : 0.f;
    sum += pv;
    ste<MODE>(sP + nl * PP, t, pv);
  }
  sum += xor32(sum);
  if (lane < 32) red[1][wave][t] = sum;
  __syncthreads();
  if (wave == 0 && lane < 32) {
    float* ml = p.partML + ((long)b * p.nch + chunk) * 64;
    ml[t] = m;
    ml[32 + t] = red[1][0][t] + red[1][1][t] + red[1][2][t] + red[1][3][t];
  }
  // ---- phase B: O[t][c] = sum_n P[n][t] Yp[n][c]   (contraction across the rows of both LDS images)
  float* Ob = p.partO + ((long)b * p.nch + chunk) * 32 * p.C;
  for (int cs = 0; cs < p.C; cs += CSB) {
    __syncthreads();
    stage_tile<MODE, NCH_ROWS, CSB>(sY, PYB, Yb, p.C, n0, p.N, cs, p.C, tid);
    __syncthreads();
    if (cs + 32 * wave < p.C) {
      mt_f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
      mma_tile<MODE, false, false>(o, sP, PP, 0, sY, PYB, 32 * wave, NCH_ROWS, lane);
      const int c = cs + 32 * wave + (lane & 31);
      if (c < p.C) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int tt = mt_row(r, lane);
          if (tt < p.tk) Ob[(long)tt * p.C + c] = o[r];
        }
      }
    }
    {   // column sums of the slab (a = mean_N Yp): thread -> (column, half of the rows)
      const int c = tid & (CSB - 1), half = tid >> 7;
      if (cs + c < p.C) {
        float s = 0.f;
        const char* q = sY + (half * 64) * PYB + c * ES;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) s += lde<MODE>(q + r * PYB, 0);
        unsafeAtomicAdd(p.a + (long)b * p.C + cs + c, s);
      }
    }
  }
}

// tok[b][t][c] = T0[t][c] + sum_k w_k[t] O_k[t][c],  w_k[t] = exp(m_k - m*) / sum_k exp(m_k - m*) l_k;  lse = m* + log(L);
// a *= 1/N (in place) and its copy in the element type.
struct TokCombArgs {
  const float* partO; const float* partML; const float* T0; int C, tk, nch; float invN;
  float* tok; float* lse; float* a; void* aE; int edt;
};
// grid (ceil(C / 64), B, tk / 8): a thread owns one channel and 2 latent tokens (64 channels x 4 token pairs per workgroup;
// the first version -- one thread per channel walking all 32 tokens x nch chunks as one dependent chain, C / 256 workgroups
// per frame -- took 80 us for 47 MB at N = 2304, C = 128).
__global__ __launch_bounds__(256) void tokattn_combine_k(const TokCombArgs p) {
  __shared__ float w[MAX_CHUNKS][8];
  const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.z * 8;
  const float* ml = p.partML + (long)b * p.nch * 64;
  if (tid < 8 && t0 + tid < p.tk) {
    const int t = t0 + tid;
    float ms = -INFINITY;
    for (int k = 0; k < p.nch; ++k) ms = fmaxf(ms, ml[k * 64 + t]);
    float L = 0.f;
    for (int k = 0; k < p.nch; ++k) L += expf(ml[k * 64 + t] - ms) * ml[k * 64 + 32 + t];
    const float inv = 1.f / L;
    for (int k = 0; k < p.nch; ++k) w[k][tid] = expf(ml[k * 64 + t] - ms) * inv;
    if (blockIdx.x == 0) p.lse[(long)b * p.tk + t] = ms + logf(L);
  }
  __syncthreads();
  const int c = blockIdx.x * 64 + (tid & 63), tq = tid >> 6;
  if (c >= p.C) return;
  const float* Ob = p.partO + (long)b * p.nch * 32 * p.C;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int tl = tq * 2 + j, t = t0 + tl;
    if (t >= p.tk) break;
    float o0 = 0.f, o1 = 0.f;
    int k = 0;
    for (; k + 1 < p.nch; k += 2) {
      o0 += w[k][tl] * Ob[((long)k * 32 + t) * p.C + c];
      o1 += w[k + 1][tl] * Ob[((long)(k + 1) * 32 + t) * p.C + c];
    }
    if (k < p.nch) o0 += w[k][tl] * Ob[((long)k * 32 + t) * p.C + c];
    p.tok[((long)b * p.tk + t) * p.C + c] = p.T0[(long)t * p.C + c] + (o0 + o1);
  }
  if (blockIdx.z == 0 && tq == 0) {
    const float av = p.a[(long)b * p.C + c] * p.invN;
    p.a[(long)b * p.C + c] = av;
    if (p.aE) ste_rt(p.aE, p.edt, (long)b * p.C + c, av);
  }
}

// ====================================================================================================================
// xattn_fwd: X1 = X + gate_av * softmax_t(X . tok^T) . tok
// ====================================================================================================================
struct XFwdArgs { const void* X; const float* tok; const float* gate_av; int N, C, tk; void* X1; };
template <int MODE>
__global__ __launch_bounds__(256) void xattn_fwd_k(const XFwdArgs p) {
  using M = MT<MODE>;
  constexpr int ES = M::ES, CSA = 64, CSB = 128;
  constexpr int PXA = M::km_pitch(CSA), PTA = M::km_pitch(CSA), PXB = M::km_pitch(CSB), PTB = M::mn_pitch(CSB), PP = M::km_pitch(32);
  constexpr int A_BYTES = NCH_ROWS * PXA + 2 * 32 * PTA, B_BYTES = NCH_ROWS * PXB + 2 * 32 * PTB;
  __shared__ __attribute__((aligned(16))) char smem[(A_BYTES > B_BYTES ? A_BYTES : B_BYTES) + NCH_ROWS * PP];
  char* sP = smem + (A_BYTES > B_BYTES ? A_BYTES : B_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, n0 = blockIdx.x * NCH_ROWS;
  const char* Xb = reinterpret_cast<const char*>(p.X) + (long)b * p.N * p.C * ES;
  const float* tokb = p.tok + (long)b * p.tk * p.C;
  const float g = *p.gate_av;
  {   // ---- phase A: S[t][n] = tok[t][:] . X[n][:]   (latent tokens along the accumulator registers)
    char* sX = smem; char* sTh = smem + NCH_ROWS * PXA; char* sTl = sTh + 32 * PTA;
    mt_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int cs = 0; cs < p.C; cs += CSA) {
      __syncthreads();
      stage_tile<MODE, NCH_ROWS, CSA>(sX, PXA, Xb, p.C, n0, p.N, cs, p.C, tid);
      stage_tile_f32<MODE, 32, CSA, true>(sTh, sTl, PTA, tokb, p.C, 0, p.tk, cs, p.C, tid);
      __syncthreads();
      mma_tile<MODE, true, true>(acc, sTh, PTA, 0, sX, PXA, 32 * wave, CSA, lane);
      if (MODE == DT_BF16) mma_tile<MODE, true, true>(acc, sTl, PTA, 0, sX, PXA, 32 * wave, CSA, lane);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (mt_row(r, lane) < p.tk) mx = fmaxf(mx, acc[r]);
    mx = fmaxf(mx, xor32(mx));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = mt_row(r, lane) < p.tk ? mexp<MODE>(acc[r] - mx) : 0.f; sum += acc[r]; }
    sum += xor32(sum);
    const float inv = 1.f / sum;
    char* prow = sP + (32 * wave + (lane & 31)) * PP;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float v[4] = {acc[4 * q] * inv, acc[4 * q + 1] * inv, acc[4 * q + 2] * inv, acc[4 * q + 3] * inv};
      stv<MODE, 4>(prow, 8 * q + 4 * (lane >> 5), v);
    }
  }
  {   // ---- phase B: X1[n][c] = X[n][c] + g * sum_t P[n][t] tok[t][c]
    char* sX = smem; char* sTh = smem + NCH_ROWS * PXB; char* sTl = sTh + 32 * PTB;
    for (int cs = 0; cs < p.C; cs += CSB) {
      __syncthreads();
      stage_tile<MODE, NCH_ROWS, CSB>(sX, PXB, Xb, p.C, n0, p.N, cs, p.C, tid);
      stage_tile_f32<MODE, 32, CSB, true>(sTh, sTl, PTB, tokb, p.C, 0, p.tk, cs, p.C, tid);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < CSB / 32; ++j) {
        if (cs + 32 * j >= p.C) break;
        mt_f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
        mma_tile<MODE, true, false>(o, sP, PP, 32 * wave, sTh, PTB, 32 * j, 32, lane);
        if (MODE == DT_BF16) mma_tile<MODE, true, false>(o, sP, PP, 32 * wave, sTl, PTB, 32 * j, 32, lane);
        const int c = 32 * j + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          char* q = sX + (32 * wave + mt_row(r, lane)) * PXB;
          ste<MODE>(q, c, lde<MODE>(q, c) + g * o[r]);
        }
      }
      copy_out_rows<MODE, CSB>(sX, PXB, 32 * wave, p.X1, nullptr, p.C, (long)b * p.N + n0 + 32 * wave, p.N - n0 - 32 * wave, cs,
                               p.C, lane);
    }
  }
}

// ====================================================================================================================
// xattn_bwd: P = softmax_t(X tok^T) recomputed; U = dX1 tok^T; dgate += sum P U; dS = g P (U - sum_t P U);
//            dX = dX1 + dS tok (+ R2);  dtok[b] += g P^T dX1 + dS^T X
// ====================================================================================================================
struct XBwdArgs {
  const void* X; const void* dX1; const float* tok; const float* gate_av; int N, C, tk;
  void* dX; const void* R2; float* dtok; float* dgate;
};
template <int MODE>
__global__ __launch_bounds__(256) void xattn_bwd_k(const XBwdArgs p) {
  using M = MT<MODE>;
  constexpr int ES = M::ES, CS = 64;
  constexpr int PXA = M::km_pitch(CS), PTA = M::km_pitch(CS), PS = M::mn_pitch(CS), PTB = M::mn_pitch(CS), PP = M::km_pitch(32);
  constexpr int A_BYTES = 2 * NCH_ROWS * PXA + 2 * 32 * PTA, B_BYTES = NCH_ROWS * PS + 32 * PTB;
  __shared__ __attribute__((aligned(16))) char smem[(A_BYTES > B_BYTES ? A_BYTES : B_BYTES) + 2 * NCH_ROWS * PP];
  __shared__ float red[4];
  char* sP = smem + (A_BYTES > B_BYTES ? A_BYTES : B_BYTES);
  char* sdS = sP + NCH_ROWS * PP;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, n0 = blockIdx.x * NCH_ROWS;
  const char* Xb = reinterpret_cast<const char*>(p.X) + (long)b * p.N * p.C * ES;
  const char* Gb = reinterpret_cast<const char*>(p.dX1) + (long)b * p.N * p.C * ES;
  const float* tokb = p.tok + (long)b * p.tk * p.C;
  const float g = *p.gate_av;
  {   // ---- phase A: S[t][n], U[t][n]
    char* sX = smem; char* sG = smem + NCH_ROWS * PXA; char* sTh = sG + NCH_ROWS * PXA; char* sTl = sTh + 32 * PTA;
    mt_f32x16 aS, aU;
#pragma unroll
    for (int r = 0; r < 16; ++r) { aS[r] = 0.f; aU[r] = 0.f; }
    for (int cs = 0; cs < p.C; cs += CS) {
      __syncthreads();
      stage_tile<MODE, NCH_ROWS, CS>(sX, PXA, Xb, p.C, n0, p.N, cs, p.C, tid);
      stage_tile<MODE, NCH_ROWS, CS>(sG, PXA, Gb, p.C, n0, p.N, cs, p.C, tid);
      stage_tile_f32<MODE, 32, CS, true>(sTh, sTl, PTA, tokb, p.C, 0, p.tk, cs, p.C, tid);
      __syncthreads();
      mma_tile<MODE, true, true>(aS, sTh, PTA, 0, sX, PXA, 32 * wave, CS, lane);
      mma_tile<MODE, true, true>(aU, sTh, PTA, 0, sG, PXA, 32 * wave, CS, lane);
      if (MODE == DT_BF16) {
        mma_tile<MODE, true, true>(aS, sTl, PTA, 0, sX, PXA, 32 * wave, CS, lane);
        mma_tile<MODE, true, true>(aU, sTl, PTA, 0, sG, PXA, 32 * wave, CS, lane);
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (mt_row(r, lane) < p.tk) mx = fmaxf(mx, aS[r]);
    mx = fmaxf(mx, xor32(mx));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { aS[r] = mt_row(r, lane) < p.tk ? mexp<MODE>(aS[r] - mx) : 0.f; sum += aS[r]; }
    sum += xor32(sum);
    const float inv = 1.f / sum;
    float dot = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { aS[r] *= inv; dot += aS[r] * aU[r]; }
    dot += xor32(dot);
    const bool nvalid = n0 + 32 * wave + (lane & 31) < p.N;
    char* prow = sP + (32 * wave + (lane & 31)) * PP;
    char* drow = sdS + (32 * wave + (lane & 31)) * PP;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float pv[4], dv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pv[e] = nvalid ? aS[4 * q + e] : 0.f;
        dv[e] = nvalid ? g * aS[4 * q + e] * (aU[4 * q + e] - dot) : 0.f;
      }
      stv<MODE, 4>(prow, 8 * q + 4 * (lane >> 5), pv);
      stv<MODE, 4>(drow, 8 * q + 4 * (lane >> 5), dv);
    }
    if (p.dgate) {
      float part = (nvalid && lane < 32) ? dot : 0.f;
      part = group_sum(part, 64);
      if (lane == 0) red[wave] = part;
      __syncthreads();
      if (tid == 0) unsafeAtomicAdd(p.dgate, red[0] + red[1] + red[2] + red[3]);
    }
  }
  {   // ---- phase B
    char* sD = smem; char* sT = smem + NCH_ROWS * PS;
    const int j = wave & 1, kh = wave >> 1;
    for (int cs = 0; cs < p.C; cs += CS) {
      __syncthreads();
      stage_tile<MODE, NCH_ROWS, CS>(sD, PS, Gb, p.C, n0, p.N, cs, p.C, tid);
      stage_tile_f32<MODE, 32, CS, false>(sT, nullptr, PTB, tokb, p.C, 0, p.tk, cs, p.C, tid);
      __syncthreads();
      mt_f32x16 a1, a2;
#pragma unroll
      for (int r = 0; r < 16; ++r) { a1[r] = 0.f; a2[r] = 0.f; }
      const bool tile_on = cs + 32 * j < p.C;
      if (tile_on) mma_tile<MODE, false, false>(a1, sP + kh * 64 * PP, PP, 0, sD + kh * 64 * PS, PS, 32 * j, 64, lane);   // P^T dX1
      __syncthreads();
      // dX rows of this wave: dX1 + dS . tok, assembled in the LDS slab
#pragma unroll
      for (int jj = 0; jj < CS / 32; ++jj) {
        if (cs + 32 * jj >= p.C) break;
        mt_f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
        mma_tile<MODE, true, false>(o, sdS, PP, 32 * wave, sT, PTB, 32 * jj, 32, lane);
        const int c = 32 * jj + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          char* q = sD + (32 * wave + mt_row(r, lane)) * PS;
          ste<MODE>(q, c, lde<MODE>(q, c) + o[r]);
        }
      }
      copy_out_rows<MODE, CS>(sD, PS, 32 * wave, p.dX, p.R2, p.C, (long)b * p.N + n0 + 32 * wave, p.N - n0 - 32 * wave, cs, p.C,
                              lane);
      __syncthreads();
      stage_tile<MODE, NCH_ROWS, CS>(sD, PS, Xb, p.C, n0, p.N, cs, p.C, tid);
      __syncthreads();
      if (tile_on) {
        mma_tile<MODE, false, false>(a2, sdS + kh * 64 * PP, PP, 0, sD + kh * 64 * PS, PS, 32 * j, 64, lane);           // dS^T X
        const int c = cs + 32 * j + (lane & 31);
        if (c < p.C) {
          float* dt = p.dtok + (long)b * p.tk * p.C + c;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int t = mt_row(r, lane);
            if (t < p.tk) unsafeAtomicAdd(dt + (long)t * p.C, g * a1[r] + a2[r]);
          }
        }
      }
    }
  }
}

// ====================================================================================================================
// tokattn_bwd: P1 = exp(T0 Yp^T - lse) recomputed; dP = dtok Yp^T; dS1 = P1 (dP - D), D[t] = dtok[t] . (tok[t] - T0[t]);
//              dYp = P1^T dtok + dS1^T T0 + da / N;   dT0b[b] += dS1 . Yp
// ====================================================================================================================
struct TokBwdArgs {
  const void* Yp; const float* T0; const float* lse; const float* D; const float* dtok; const float* da; float invN;
  int N, C, tk; void* dYp; float* dT0b;
};
template <int MODE>
__global__ __launch_bounds__(256) void tokattn_bwd_k(const TokBwdArgs p) {
  using M = MT<MODE>;
  constexpr int ES = M::ES, CS = 64;
  constexpr int PYA = M::km_pitch(CS), PTA = M::km_pitch(CS), PS = M::mn_pitch(CS), PTB = M::mn_pitch(CS), PP = M::km_pitch(32);
  constexpr int A_BYTES = NCH_ROWS * PYA + 3 * 32 * PTA, B_BYTES = NCH_ROWS * PS + 2 * 32 * PTB;
  __shared__ __attribute__((aligned(16))) char smem[(A_BYTES > B_BYTES ? A_BYTES : B_BYTES) + 2 * NCH_ROWS * PP];
  char* sP = smem + (A_BYTES > B_BYTES ? A_BYTES : B_BYTES);
  char* sdS = sP + NCH_ROWS * PP;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, n0 = blockIdx.x * NCH_ROWS;
  const char* Yb = reinterpret_cast<const char*>(p.Yp) + (long)b * p.N * p.C * ES;
  const float* dtokb = p.dtok + (long)b * p.tk * p.C;
  {   // ---- phase A: S[t][n] = T0[t] . Yp[n], dP[t][n] = dtok[t] . Yp[n]
    char* sY = smem; char* sTh = smem + NCH_ROWS * PYA; char* sTl = sTh + 32 * PTA; char* sG = sTl + 32 * PTA;
    mt_f32x16 aS, aD;
#pragma unroll
    for (int r = 0; r < 16; ++r) { aS[r] = 0.f; aD[r] = 0.f; }
    for (int cs = 0; cs < p.C; cs += CS) {
      __syncthreads();
      stage_tile<MODE, NCH_ROWS, CS>(sY, PYA, Yb, p.C, n0, p.N, cs, p.C, tid);
      stage_tile_f32<MODE, 32, CS, true>(sTh, sTl, PTA, p.T0, p.C, 0, p.tk, cs, p.C, tid);
      stage_tile_f32<MODE, 32, CS, false>(sG, nullptr, PTA, dtokb, p.C, 0, p.tk, cs, p.C, tid);
      __syncthreads();
      mma_tile<MODE, true, true>(aS, sTh, PTA, 0, sY, PYA, 32 * wave, CS, lane);
      if (MODE == DT_BF16) mma_tile<MODE, true, true>(aS, sTl, PTA, 0, sY, PYA, 32 * wave, CS, lane);
      mma_tile<MODE, true, true>(aD, sG, PTA, 0, sY, PYA, 32 * wave, CS, lane);
    }
    const bool nvalid = n0 + 32 * wave + (lane & 31) < p.N;
    char* prow = sP + (32 * wave + (lane & 31)) * PP;
    char* drow = sdS + (32 * wave + (lane & 31)) * PP;
    float lser[16], Dr[16];            // statistics first, unconditionally (a load inside `ok ? .. : 0` is a serialised round trip each)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), tc = t < p.tk ? t : 0;
      lser[r] = p.lse[(long)b * p.tk + tc];
      Dr[r] = p.D[(long)b * p.tk + tc];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float pv[4], dv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int t = 8 * q + 4 * (lane >> 5) + e;
        const bool ok = nvalid && t < p.tk;
        const float pr = ok ? mexp<MODE>(aS[4 * q + e] - lser[4 * q + e]) : 0.f;
        pv[e] = pr;
        dv[e] = ok ? pr * (aD[4 * q + e] - Dr[4 * q + e]) : 0.f;
      }
      stv<MODE, 4>(prow, 8 * q + 4 * (lane >> 5), pv);
      stv<MODE, 4>(drow, 8 * q + 4 * (lane >> 5), dv);
    }
  }
  {   // ---- phase B
    char* sY = smem; char* sG = smem + NCH_ROWS * PS; char* sT = sG + 32 * PTB;
    const int j = wave & 1, kh = wave >> 1;
    for (int cs = 0; cs < p.C; cs += CS) {
      __syncthreads();
      stage_tile<MODE, NCH_ROWS, CS>(sY, PS, Yb, p.C, n0, p.N, cs, p.C, tid);
      stage_tile_f32<MODE, 32, CS, false>(sG, nullptr, PTB, dtokb, p.C, 0, p.tk, cs, p.C, tid);
      stage_tile_f32<MODE, 32, CS, false>(sT, nullptr, PTB, p.T0, p.C, 0, p.tk, cs, p.C, tid);
      __syncthreads();
      if (cs + 32 * j < p.C) {                                         // dT0b += dS1^T . Yp
        mt_f32x16 a2;
#pragma unroll
        for (int r = 0; r < 16; ++r) a2[r] = 0.f;
        mma_tile<MODE, false, false>(a2, sdS + kh * 64 * PP, PP, 0, sY + kh * 64 * PS, PS, 32 * j, 64, lane);
        const int c = cs + 32 * j + (lane & 31);
        if (c < p.C) {
          float* dt = p.dT0b + (long)b * p.tk * p.C + c;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int t = mt_row(r, lane);
            if (t < p.tk) unsafeAtomicAdd(dt + (long)t * p.C, a2[r]);
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int jj = 0; jj < CS / 32; ++jj) {                            // dYp rows of this wave, assembled over the Yp slab
        if (cs + 32 * jj >= p.C) break;
        mt_f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
        mma_tile<MODE, true, false>(o, sP, PP, 32 * wave, sG, PTB, 32 * jj, 32, lane);
        mma_tile<MODE, true, false>(o, sdS, PP, 32 * wave, sT, PTB, 32 * jj, 32, lane);
        const int c = 32 * jj + (lane & 31);
        const float bias = p.da[(long)b * p.C + (cs + c < p.C ? cs + c : p.C - 1)] * p.invN;      // unconditional, clamped (columns >= C are never copied out)
#pragma unroll
        for (int r = 0; r < 16; ++r) ste<MODE>(sY + (32 * wave + mt_row(r, lane)) * PS, c, o[r] + bias);
      }
      copy_out_rows<MODE, CS>(sY, PS, 32 * wave, p.dYp, nullptr, p.C, (long)b * p.N + n0 + 32 * wave, p.N - n0 - 32 * wave, cs, p.C,
                              lane);
    }
  }
}

// D[b][t] = sum_c dtok[b][t][c] * (tok[b][t][c] - T0[t][c])      one wavefront per (b, t) row
__global__ __launch_bounds__(256) void tok_rowdot_k(const float* dtok, const float* tok, const float* T0, int rows, int tk, int C,
                                                    float* D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int t = row % tk;
  const float* a = dtok + (long)row * C;
  const float* q = tok + (long)row * C;
  const float* z = T0 + (long)t * C;
  float s = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    const float4 x = *reinterpret_cast<const float4*>(a + c), y = *reinterpret_cast<const float4*>(q + c),
                 w = *reinterpret_cast<const float4*>(z + c);
    s += x.x * (y.x - w.x) + x.y * (y.y - w.y) + x.z * (y.z - w.z) + x.w * (y.w - w.w);
  }
  s = group_sum(s, 64);
  if (lane == 0) D[row] = s;
}

// ---- host side -------------------------------------------------------------------------------------------------------
static bool attn_shape_ok(const Ctx& ctx, int N, int C, int tk) {
  const int ve = ctx.mode == DT_BF16 ? 8 : 4;
  if (tk < 1 || tk > 32) { set_error("latent-token attention: tk=%d must be in 1..32", tk); return false; }
  if (C % ve != 0) { set_error("latent-token attention: C=%d must be a multiple of %d in this dtype", C, ve); return false; }
  if ((N + NCH_ROWS - 1) / NCH_ROWS > MAX_CHUNKS) { set_error("latent-token attention: N=%d too large", N); return false; }
  return true;
}
long tokattn_scratch_floats(int B, int N, int C) {
  const long nch = (N + NCH_ROWS - 1) / NCH_ROWS;
  return (long)B * nch * (32L * C + 64);
}
// attn2.hip
bool attn2_ok(const Ctx& ctx, int C);
void xattn_fwd2(const Ctx& ctx, const void* X, const void* tokpk, const float* gate_av, int B, int N, int C, int tk, void* X1);
void xattn_bwd2(const Ctx& ctx, const void* X, const void* dX1, const void* tokpk, const float* gate_av, int B, int N, int C, int tk,
                void* dX, const void* R2, float* dtok, float* dgate);
void tokattn_bwd2(const Ctx& ctx, const void* Yp, const void* T0pk, const void* dtokpk, const float* lse, const float* D,
                  const float* da, float invN, int B, int N, int C, int tk, void* dYp, float* dT0b);

bool tokattn_fwd_small_ok(const Ctx& ctx, int N, int C);
bool tokattn_bwd_csplit(int B, int N, int C);      // short frames, wide channels: the C-split kernel (attn2.hip)
void tokattn_fwd_small(const Ctx& ctx, const void* Yp, const float* T0, const void* T0pk, int B, int N, int C, int tk, float* tok,
                       void* tokpk, float* lse, float* a, void* aE);

void tokattn_fwd(const Ctx& ctx, const void* Yp, const float* T0, int B, int N, int C, int tk, float* tok, float* lse, float* a,
                 void* aE, float* scratch, void* tokpk, const void* T0pk) {
  if (!attn_shape_ok(ctx, N, C, tk)) return;
  if (tokpk && T0pk && aE && tokattn_fwd_small_ok(ctx, N, C)) {       // short frames: one workgroup per frame, final results
    tokattn_fwd_small(ctx, Yp, T0, T0pk, B, N, C, tk, tok, tokpk, lse, a, aE);
    return;
  }
  const int nch = (N + NCH_ROWS - 1) / NCH_ROWS;
  TokFwdArgs p{Yp, T0, N, C, tk, nch, scratch, scratch + (long)B * nch * 32 * C, a};
  hipStream_t s = (hipStream_t)ctx.stream;
  if (ctx.mode == DT_BF16) hipLaunchKernelGGL(tokattn_fwd_k<DT_BF16>, dim3(nch, B), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(tokattn_fwd_k<DT_F32>, dim3(nch, B), dim3(256), 0, s, p);
  TokCombArgs q{p.partO, p.partML, T0, C, tk, nch, 1.f / (float)N, tok, lse, a, aE, ctx.mode};
  hipLaunchKernelGGL(tokattn_combine_k, dim3((C + 63) / 64, B, (tk + 7) / 8), dim3(256), 0, s, q);
  if (tokpk && attn2_ok(ctx, C)) tok_pack(ctx, tok, B, tk, C, tokpk);
}
void xattn_fwd(const Ctx& ctx, const void* X, const float* tok, const float* gate_av, int B, int N, int C, int tk, void* X1,
               const void* tokpk) {
  if (!attn_shape_ok(ctx, N, C, tk)) return;
  if (tokpk && attn2_ok(ctx, C)) { xattn_fwd2(ctx, X, tokpk, gate_av, B, N, C, tk, X1); return; }
  XFwdArgs p{X, tok, gate_av, N, C, tk, X1};
  const dim3 grid((N + NCH_ROWS - 1) / NCH_ROWS, B);
  if (ctx.mode == DT_BF16) hipLaunchKernelGGL(xattn_fwd_k<DT_BF16>, grid, dim3(256), 0, (hipStream_t)ctx.stream, p);
  else hipLaunchKernelGGL(xattn_fwd_k<DT_F32>, grid, dim3(256), 0, (hipStream_t)ctx.stream, p);
}
void xattn_bwd(const Ctx& ctx, const void* X, const void* dX1, const float* tok, const float* gate_av, int B, int N, int C, int tk,
               void* dX, const void* R2, float* dtok, float* dgate, const void* tokpk) {
  if (!attn_shape_ok(ctx, N, C, tk)) return;
  if (tokpk && attn2_ok(ctx, C)) { xattn_bwd2(ctx, X, dX1, tokpk, gate_av, B, N, C, tk, dX, R2, dtok, dgate); return; }
  XBwdArgs p{X, dX1, tok, gate_av, N, C, tk, dX, R2, dtok, dgate};
  const dim3 grid((N + NCH_ROWS - 1) / NCH_ROWS, B);
  if (ctx.mode == DT_BF16) hipLaunchKernelGGL(xattn_bwd_k<DT_BF16>, grid, dim3(256), 0, (hipStream_t)ctx.stream, p);
  else hipLaunchKernelGGL(xattn_bwd_k<DT_F32>, grid, dim3(256), 0, (hipStream_t)ctx.stream, p);
}
void tokattn_bwd(const Ctx& ctx, const void* Yp, const float* T0, const float* tok, const float* lse, const float* dtok,
                 const float* da, float invN, int B, int N, int C, int tk, void* dYp, float* dT0b, float* Dscratch,
                 const void* T0pk, void* dtokpk) {
  if (!attn_shape_ok(ctx, N, C, tk)) return;
  if (T0pk && dtokpk && attn2_ok(ctx, C) && (N >= 512 || tokattn_bwd_csplit(B, N, C))) {    // (tools/attn_bench.py: the generic kernel wins on short frames)
    tok_pack(ctx, dtok, B, tk, C, dtokpk, tok, T0, Dscratch);       // packed dtok + D[b][t] = dtok . (tok - T0)
    tokattn_bwd2(ctx, Yp, T0pk, dtokpk, lse, Dscratch, da, invN, B, N, C, tk, dYp, dT0b);
    return;
  }
  hipStream_t s = (hipStream_t)ctx.stream;
  hipLaunchKernelGGL(tok_rowdot_k, dim3((B * tk + 3) / 4), dim3(256), 0, s, dtok, tok, T0, B * tk, tk, C, Dscratch);
  TokBwdArgs p{Yp, T0, lse, Dscratch, dtok, da, invN, N, C, tk, dYp, dT0b};
  const dim3 grid((N + NCH_ROWS - 1) / NCH_ROWS, B);
  if (ctx.mode == DT_BF16) hipLaunchKernelGGL(tokattn_bwd_k<DT_BF16>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(tokattn_bwd_k<DT_F32>, grid, dim3(256), 0, s, p);
}

}  // namespace dgsct
