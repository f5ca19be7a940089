// Fused row passes of the adapter's gate / bottleneck chain for the early stages (C <= 256 channels: every backbone's stages 0-1,
// 58 % of the step), bf16.  Reference math: DG-SCT/AVE/nets/net_trans.py:598-613 (spatial gate, modulation), :626-638
// (ln_before, grouped down-projection, BatchNorm-1) and their autograd.
//
// Why: between X1 (the output of the latent-token attention) and the bottleneck the unfused schedule makes seven passes over
// [rows, C] tensors (scale_cols -> GEMM -> rowdot -> modulation + LayerNorm -> projection -> statistics), each a launch that
// streams 50-130 MB at stage 0.  The work per token row is tiny -- a [C/2 x C] and a [C/8 x C] product -- and every step is
// row-local except two per-frame vectors (ch, aq2) and the BatchNorm sums, so ONE pass over X1 does all of it:
//
//   gatemod_fwd:   s  = relu(X1 (1 + ch_b) Wv2^T + bv2) . (aq2_b * ws) + bs            (spatial logits, net_trans.py:602-605)
//                  X2 = X1 (alpha ch_b + beta sigmoid(s) + gamma tg_b + 1 - alpha)        (:611-612)
//                  X3 = LN(X2)  ->  Zp = X3 (x)_g Wd  ->  BatchNorm-1 sums of Zp          (:627-636)
//
// Structure (cf. attn2.hip): a WAVEFRONT owns a block of 32 consecutive token rows of one frame.  The block arrives as
// coalesced 16-byte loads (the next block's loads are in flight while this one is processed), goes through a wave-private LDS
// image and becomes the B operand (token = lane & 31, 8 channels per k-step) of v_mfma_f32_32x32x16_bf16 -- so every product
// comes out as [output channel][token]: a lane holds 16 output channels of ITS token, reductions over channels (the row dot
// with aq2*ws, the LayerNorm statistics) are in-lane sums plus one exchange with lane ^ 32, and a product's result (rounded to
// bf16) is directly the B operand of the next product.  The per-frame scaling X1 (1 + ch_b) is folded into the WEIGHTS: a
// workgroup works on one frame and builds bf16(Wv2 (1 + ch_b)) (from the fp32 master) in LDS once; the grouped down-projection
// is a block-diagonal [32 x C] image.  No __syncthreads in the token loop.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include "prims.h"
#include "device_util.h"
#include "mma_tile.h"
#include "err.h"

namespace dgsct {

namespace {
typedef mt_bf16x8 bfx8;
typedef mt_f32x16 f32x16;

__device__ __forceinline__ float fg_xor32(float v) { return __shfl_xor(v, 32, 64); }
__device__ __forceinline__ void fg_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  asm volatile("" ::: "memory");      // also keeps the loop-invariant LDS vectors (modulation, LN weights) OUT of registers: hoisted, they cost 3C/2 VGPRs
}
__device__ __forceinline__ bfx8 fg_lds8(const char* p) { return *reinterpret_cast<const bfx8*>(p); }
__device__ __forceinline__ void fg_unpack8(const bfx8& f, float (&x)[8]) {
  const uint4 u = __builtin_bit_cast(uint4, f);
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) { x[2 * e] = __uint_as_float(w[e] << 16); x[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
}
__device__ __forceinline__ bfx8 fg_pack8(const float (&x)[8]) {
  const uint4 u = make_uint4(f2bf2(x[0], x[1]), f2bf2(x[2], x[3]), f2bf2(x[4], x[5]), f2bf2(x[6], x[7]));
  return __builtin_bit_cast(bfx8, u);
}

typedef unsigned fg_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned fg_u32x2 __attribute__((ext_vector_type(2)));   // (HIP's uint4 is a struct: arrays of it are copied with memcpy and stay in scratch)
template <int NV>
__device__ __forceinline__ void fg_gload(fg_u32x4 (&nx)[NV], const unsigned short* base, int lane) {
  const fg_u32x4* src = reinterpret_cast<const fg_u32x4*>(base);
#pragma unroll
  for (int i = 0; i < NV; ++i) nx[i] = src[i * 64 + lane];
}

template <int C_>
struct FG {
  static constexpr int C = C_, DD = C / 2, DDP = (DD + 31) / 32 * 32, KS = C / 16, MT = DDP / 32;
  static constexpr int PW = C * 2 + 16;                 // pitch of a K-major [rows][C] bf16 image: conflict-free ds_read_b128 rows
  static constexpr int NV = C / 16;                     // 16-byte chunks per lane of one 32 x C block
  static constexpr int CPR = C / 8;                     // 16-byte chunks per row
  static constexpr int W2_BYTES = DDP * PW, WD_BYTES = 32 * PW, XIMG = 32 * PW;
  static constexpr int VEC_FLOATS = 2 * DDP + 3 * C + 32;      // bv2 | aq2*ws | modulation | ln weight | ln bias | BN shift
  static constexpr int RED_FLOATS = 4 * 64;
  static constexpr int SMEM = W2_BYTES + WD_BYTES + 4 * XIMG + (VEC_FLOATS + RED_FLOATS) * 4;
  static_assert(SMEM <= 160 * 1024, "LDS budget");
};

struct GF {
  const unsigned short* X1; const float* ch; const unsigned short* aq2; const float* Wv2; const float* bv2; const float* ws;
  const float* bs; const float* tg; float alpha, beta, gamma; const float* lnw; const float* lnb; float eps;
  int N, ds, g, wpf; const float* Wd;
  float* sl; unsigned short* X3; float* mu; float* rstd; unsigned short* Zp; float* stats; unsigned short* vq2;
};

// the per-frame operand images of a workgroup: bf16(Wv2 (1 + ch_b)) [DDP x C] and the block-diagonal down-projection [32 x C]
template <int C>
__device__ __forceinline__ void fg_build_weights(char* w2img, char* wdimg, const float* Wv2, const float* chb, const float* Wd, int ds,
                                                 int g, int tid) {
  using G = FG<C>;
  for (int i = tid; i < G::DDP * G::CPR; i += 256) {
    const int j = i / G::CPR, c = (i - j * G::CPR) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (j < G::DD) {
      float w[8], s[8];
      ldf<8>(Wv2, (long)j * C + c, w);
      ldf<8>(chb, c, s);
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] *= 1.f + s[e];
      v = make_uint4(f2bf2(w[0], w[1]), f2bf2(w[2], w[3]), f2bf2(w[4], w[5]), f2bf2(w[6], w[7]));
    }
    *reinterpret_cast<uint4*>(w2img + j * G::PW + c * 2) = v;
  }
  const int cg = C / g, dg = ds / g;
  for (int i = tid; i < 32 * G::CPR; i += 256) {
    const int jz = i / G::CPR, c = (i - jz * G::CPR) * 8;
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = 0.f;
    if (jz < ds) {
      const int gi = jz / dg;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int cl = c + e - gi * cg;
        const bool in = cl >= 0 && cl < cg;
        const float t = Wd[(long)jz * cg + (in ? cl : 0)];
        w[e] = in ? t : 0.f;
      }
    }
    *reinterpret_cast<uint4*>(wdimg + jz * G::PW + c * 2) = make_uint4(f2bf2(w[0], w[1]), f2bf2(w[2], w[3]), f2bf2(w[4], w[5]), f2bf2(w[6], w[7]));
  }
}

template <int C>
__global__ __launch_bounds__(256) void gatemod_fwd_k(const GF p) {
  using G = FG<C>;
  __shared__ __attribute__((aligned(16))) char smem[G::SMEM];
  char* w2img = smem;
  char* wdimg = w2img + G::W2_BYTES;
  char* ximg0 = wdimg + G::WD_BYTES;
  float* bv2s = reinterpret_cast<float*>(ximg0 + 4 * G::XIMG);
  float* w2s = bv2s + G::DDP;
  float* mcs = w2s + G::DDP;
  float* lnws = mcs + C;
  float* lnbs = lnws + C;
  float* shs = lnbs + C;
  float* red = shs + 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, tl = lane & 31;
  const int b = blockIdx.y;
  const int ds = p.ds;
  const float* chb = p.ch + (long)b * C;

  fg_build_weights<C>(w2img, wdimg, p.Wv2, chb, p.Wd, ds, p.g, tid);
  for (int i = tid; i < G::DDP; i += 256) {
    const int ic = i < G::DD ? i : 0;
    const float bb = p.bv2[ic], w = bf2f(p.aq2[(long)b * G::DD + ic]) * p.ws[ic];
    bv2s[i] = i < G::DD ? bb : 0.f;
    w2s[i] = i < G::DD ? w : 0.f;
  }
  {
    const float tgv = p.tg ? p.gamma * p.tg[b] : 0.f;
    for (int i = tid; i < C; i += 256) {
      mcs[i] = p.alpha * chb[i] + 1.f - p.alpha + tgv;
      lnws[i] = p.lnw ? p.lnw[i] : 1.f;
      lnbs[i] = p.lnw ? p.lnb[i] : 0.f;
    }
  }
  __syncthreads();
  // shift of the one-pass BatchNorm sums: the LayerNorm bias through the (rounded) down-projection -- the expected channel mean of
  // Zp, and the same bits in every workgroup (bn_stats' own choice, row 0 of the tensor, would cost every wave an extra block)
  if (tid < 32) {
    float s = 0.f;
    if (p.lnw)
      for (int c = 0; c < C; ++c) s += lnbs[c] * bf2f(*reinterpret_cast<const unsigned short*>(wdimg + tid * G::PW + c * 2));
    shs[tid] = bf2f(f2bf(s));
  }
  __syncthreads();
  if (p.stats && blockIdx.x == 0 && b == 0 && tid < ds) p.stats[tid] = shs[tid];

  const int nblk = p.N / 32, stride = p.wpf * 4;
  char* img = ximg0 + wave * G::XIMG;
  const long frame0 = (long)b * p.N;
  fg_u32x4 nx[G::NV];
  int blk = blockIdx.x * 4 + wave;
  // (block loads are UNCONDITIONAL from a clamped block index: under a condition hipcc keeps the prefetch registers in scratch
  //  memory and waits for every load at once)
  fg_gload<G::NV>(nx, p.X1 + (frame0 + (long)(blk < nblk ? blk : nblk - 1) * 32) * C, lane);
  f32x16 zs, zq, shv;
#pragma unroll
  for (int r = 0; r < 16; ++r) { zs[r] = 0.f; zq[r] = 0.f; shv[r] = shs[mt_row(r, lane)]; }
  const float bsv = *p.bs;
  const float invC = 1.f / (float)C;
  const char* xrow = img + tl * G::PW + h * 16;

  for (; blk < nblk; blk += stride) {
#pragma unroll
    for (int i = 0; i < G::NV; ++i) {
      const int idx = i * 64 + lane, r = idx / G::CPR, c8 = idx - r * G::CPR;
      *reinterpret_cast<fg_u32x4*>(img + r * G::PW + c8 * 16) = nx[i];
    }
    fg_wave_sync();
    fg_gload<G::NV>(nx, p.X1 + (frame0 + (long)(blk + stride < nblk ? blk + stride : nblk - 1) * 32) * C, lane);
    bfx8 xf[G::KS];
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) xf[kk] = fg_lds8(xrow + kk * 32);
    const long row = frame0 + (long)blk * 32 + tl;

    // ---- spatial logit: s = relu(W2_b x + bv2) . (aq2_b * ws) + bs
    float s = 0.f;
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const char* wrow = w2img + (mt * 32 + tl) * G::PW + h * 16;
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fg_lds8(wrow + kk * 32), xf[kk], acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j0 = mt * 32 + 8 * q + 4 * h;
        const float4 bb = *reinterpret_cast<const float4*>(bv2s + j0);
        const float4 ww = *reinterpret_cast<const float4*>(w2s + j0);
        const float v0 = fmaxf(acc[4 * q] + bb.x, 0.f), v1 = fmaxf(acc[4 * q + 1] + bb.y, 0.f);
        const float v2 = fmaxf(acc[4 * q + 2] + bb.z, 0.f), v3 = fmaxf(acc[4 * q + 3] + bb.w, 0.f);
        s += v0 * ww.x + v1 * ww.y + v2 * ww.z + v3 * ww.w;
        if (p.vq2 && j0 < G::DD) *reinterpret_cast<uint2*>(p.vq2 + row * G::DD + j0) = make_uint2(f2bf2(v0, v1), f2bf2(v2, v3));
      }
    }
    s += fg_xor32(s);
    s += bsv;
    if (lane < 32) p.sl[row] = s;
    const float sgv = p.beta * sigmoidf_(s);

    // ---- modulation + LayerNorm over the channels of the lane's token (its half of the row + lane ^ 32)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) {
      float x[8];
      fg_unpack8(xf[kk], x);
      const int c0 = 16 * kk + 8 * h;
      const float4 m0 = *reinterpret_cast<const float4*>(mcs + c0), m1 = *reinterpret_cast<const float4*>(mcs + c0 + 4);
      const float m[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float v = x[e] * (m[e] + sgv); s1 += v; s2 += v * v; }
    }
    float mean = 0.f, rs = 1.f;
    if (p.lnw) {
      s1 += fg_xor32(s1); s2 += fg_xor32(s2);
      mean = s1 * invC;
      rs = rsqrtf(fmaxf(s2 * invC - mean * mean, 0.f) + p.eps);
      if (lane < 32) { p.mu[row] = mean; p.rstd[row] = rs; }
    }
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) {
      float x[8];
      fg_unpack8(xf[kk], x);
      const int c0 = 16 * kk + 8 * h;
      const float4 m0 = *reinterpret_cast<const float4*>(mcs + c0), m1 = *reinterpret_cast<const float4*>(mcs + c0 + 4);
      const float4 w0 = *reinterpret_cast<const float4*>(lnws + c0), w1 = *reinterpret_cast<const float4*>(lnws + c0 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(lnbs + c0), b1 = *reinterpret_cast<const float4*>(lnbs + c0 + 4);
      const float m[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
      const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (x[e] * (m[e] + sgv) - mean) * rs * w[e] + bb[e];
      xf[kk] = fg_pack8(x);                                 // rounded once: the stored X3 and the projection operand are the same bits
      *reinterpret_cast<bfx8*>(img + tl * G::PW + kk * 32 + h * 16) = xf[kk];
    }

    // ---- grouped down-projection (block-diagonal [32 x C] image) + BatchNorm-1 sums of Zp as stored
    {
      f32x16 az;
#pragma unroll
      for (int r = 0; r < 16; ++r) az[r] = 0.f;
      const char* wrow = wdimg + tl * G::PW + h * 16;
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk) az = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fg_lds8(wrow + kk * 32), xf[kk], az, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int jz0 = 8 * q + 4 * h;
        if (jz0 < ds) {
          const unsigned lo = f2bf2(az[4 * q], az[4 * q + 1]), hi = f2bf2(az[4 * q + 2], az[4 * q + 3]);
          *reinterpret_cast<uint2*>(p.Zp + row * ds + jz0) = make_uint2(lo, hi);
          const float d0 = __uint_as_float(lo << 16) - shv[4 * q], d1 = __uint_as_float(lo & 0xffff0000u) - shv[4 * q + 1];
          const float d2 = __uint_as_float(hi << 16) - shv[4 * q + 2], d3 = __uint_as_float(hi & 0xffff0000u) - shv[4 * q + 3];
          zs[4 * q] += d0; zs[4 * q + 1] += d1; zs[4 * q + 2] += d2; zs[4 * q + 3] += d3;
          zq[4 * q] += d0 * d0; zq[4 * q + 1] += d1 * d1; zq[4 * q + 2] += d2 * d2; zq[4 * q + 3] += d3 * d3;
        }
      }
    }
    fg_wave_sync();
    {   // X3 rows leave as coalesced 16-byte stores
      fg_u32x4* dst = reinterpret_cast<fg_u32x4*>(p.X3 + (frame0 + (long)blk * 32) * C);
#pragma unroll
      for (int i = 0; i < G::NV; ++i) {
        const int idx = i * 64 + lane, r = idx / G::CPR, c8 = idx - r * G::CPR;
        dst[idx] = *reinterpret_cast<const fg_u32x4*>(img + r * G::PW + c8 * 16);
      }
    }
    fg_wave_sync();
  }

  if (p.stats) {                                             // sums over the 32 token lanes, then over the 4 waves, one atomic per channel
#pragma unroll
    for (int r = 0; r < 16; ++r) { zs[r] = group_sum(zs[r], 32); zq[r] = group_sum(zq[r], 32); }
    if (tl == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { red[wave * 64 + mt_row(r, lane)] = zs[r]; red[wave * 64 + 32 + mt_row(r, lane)] = zq[r]; }
    }
    __syncthreads();
    if (tid < 64) {
      const int jz = tid & 31, which = tid >> 5;
      if (jz < ds) {
        const float t = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
        unsafeAtomicAdd(p.stats + (1 + which) * ds + jz, t);
      }
    }
  }
}


// =====================================================================================================================
// Backward of the same chain (net_trans.py:598-638 through autograd), C in {96, 128}: ONE pass over X1 instead of
//   bn_bwd_apply -> gproj_wide (dX3) -> modln_bwd -> spatial_bwd -> colsum (u) -> relu_bwd_scale (dvq2) -> GEMM (dXc) -> xc_bwd.
// Here the products are issued the OTHER way round -- token rows are the M operand, so an accumulator tile is [token][channel]
// with the CHANNEL in the lane: the many per-channel sums over tokens of a backward pass (d ln_before.weight / bias, dch, d bv2,
// u = sum_n dsl * vq2) are in-lane sums over the 16 accumulator registers, and the three per-token sums over channels (the
// LayerNorm backward pair and dsg) are one cross-lane reduction of 16 values each.  vq2 is RECOMPUTED from X1 (same
// per-frame-scaled weight image as the forward: same bits, same ReLU decisions) instead of being stored by the forward; Xc, which
// only the dWv2 GEMM on the aux stream still wants materialised, is written from here.
//   dZp  = BN1 backward of dZ (in place)                                  -> consumed by the dWd GEMM
//   dX3  = dZp (x)_g Wd;  LN / modulation backward -> dX1a, dsg, d lnw, d lnb, dch (first part), dtg
//   dsl  = dsg sg (1 - sg) + map (dMap - sum_n map dMap) (1 - tanh(sl)^2)                      (spatial_bwd)
//   dvq2 = dsl * (aq2_b * ws) * (vq2 > 0);  u += dsl * vq2;  d bv2 += dvq2                     -> dvq2 stored for the dWv2 GEMM
//   dX1  = dX1a + dvq2 W2_b   (= dXc (1 + ch_b));   dch += dXc * X1
// 8 wavefronts per workgroup (one frame per workgroup, like the forward), wave-private LDS images, no __syncthreads in the loop.
template <int C_>
struct BG {
  static constexpr int C = C_, DD = C / 2, DDP = (DD + 31) / 32 * 32, DS = C / 8, KS = C / 16, JT = DDP / 32, NT = C / 32;
  static constexpr int PW = C * 2 + 16, PV = DDP * 2 + 16, PZ = 16 * 2 + 16;     // pitches: [.][C], [.][DDP], [.][16] bf16 images
  static constexpr int NV = C / 16, CPR = C / 8;
  static constexpr int NZ = (8 * DS + 63) / 64;          // 8-byte chunks (4 channels of one token) of a 32 x DS block per lane
  static constexpr int WAVES = 4;
  static constexpr int NTHR = WAVES * 64;
  static constexpr int W2_BYTES = DDP * PW, WDT_BYTES = C * PZ;
  static constexpr int XIMG = 32 * PW, DVIMG = 32 * PV, DZIMG = 32 * PZ, TOKV = 10 * 32 * 4;
  static constexpr int WAVE_BYTES = XIMG + DVIMG + DZIMG + TOKV;
  static constexpr int VEC_FLOATS = 2 * DDP + 3 * C + 5 * 16;     // bv2 | aq2*ws | modulation | ln weight | 1 + ch | BN1: a, sh, k1(=a), k2, k3
  static constexpr int NQ = 4 * C + 2 * DDP;              // per-channel sums of a wave: dlnw | dlnb | dchA | dchB | u | dbv2
  static constexpr int NK = (NQ + 255) / 256;             // reduction slots per thread
  static constexpr int SMEM = W2_BYTES + WDT_BYTES + WAVES * WAVE_BYTES + VEC_FLOATS * 4 + 64;
  static_assert(DS <= 16 && DS % 4 == 0, "bottleneck width");
  static_assert(WAVES * WAVE_BYTES >= WAVES * NQ * 4, "the wave images double as the reduction scratch");
  static_assert(SMEM <= 160 * 1024, "LDS budget");
};

struct GB {
  const unsigned short* X1; const float* ch; const unsigned short* aq2; const float* Wv2; const float* bv2; const float* ws;
  const float* tg; float alpha, beta, gamma; const float* lnw; const float* mu; const float* rstd;
  const float* sl; const float* sg; const float* map; const float* dMap; const float* mapdot;
  int B, N, g, wpf; const float* Wd;
  unsigned short* dZ; const unsigned short* Zp; const float* bn_mean; const float* bn_rstd; const float* bn_sc; const float* bn_sh;
  const float* bn_sums; int has_bn, training; float inv_rows;
  unsigned short* dX1; unsigned short* dvq2; unsigned short* Xc;
  float* dch; float* u; float* dtg; float* part;      // part: [workgroup][2 C + DD + 1] = d lnw | d lnb | d bv2 | d bs
};

__device__ __forceinline__ unsigned short fg_ldsu16(const char* p) { return *reinterpret_cast<const unsigned short*>(p); }

// HAS_LN (ln_before present) is a compile-time flag: as a run-time condition inside the unrolled element loops it compiled to a
// branch per element, each behind its own ds_read + s_waitcnt lgkmcnt(0) -- 48 serialised LDS round trips per pass
template <int C, bool HAS_LN>
__global__ __launch_bounds__(256) void gatemod_bwd_k(const GB p) {
  using G = BG<C>;
  __shared__ __attribute__((aligned(16))) char smem[G::SMEM];
  char* w2img = smem;
  char* wdT = w2img + G::W2_BYTES;
  char* wave0 = wdT + G::WDT_BYTES;
  float* bv2s = reinterpret_cast<float*>(wave0 + G::WAVES * G::WAVE_BYTES);
  float* w2s = bv2s + G::DDP;
  float* mcs = w2s + G::DDP;
  float* lnws = mcs + C;
  float* opcs = lnws + C;
  float* bnv = opcs + C;                                  // [5][16]
  float* misc = bnv + 80;
  const int tid = threadIdx.x, lane_ = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = lane_, h = lane_ >> 5, tl = lane_ & 31;
  // (measured: with has_ln a compile-time constant the straight-line block body needs > 512 registers and spills: 335 / 543 us
  //  against 237 / 257 us with the per-element branches of the run-time flag, which keep the scheduling regions small)
  const bool has_ln = p.lnw != nullptr;
  // Work items = (frame, part of the frame); a workgroup walks items blockIdx.x, + gridDim.x, ...: the grid is ONE workgroup per CU
  // (LDS) and the host picks the parts per frame so that the items fill whole rounds of it (160 frames x 8 parts = 5 x 256).
  const int nitems = p.B * p.wpf;
  float keep[G::NK];
#pragma unroll
  for (int k = 0; k < G::NK; ++k) keep[k] = 0.f;
  //                              // this thread's slots of [d lnw | d lnb | d bv2], summed over its items
  float keep_bs = 0.f;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
  const int b = item / p.wpf, part = item - b * p.wpf;
  const float* chb = p.ch + (long)b * C;
  __syncthreads();                                         // the previous item's reduction has read the images / vectors

  // ---- per-frame operand images and vectors
  for (int i = tid; i < G::DDP * G::CPR; i += G::NTHR) {
    const int j = i / G::CPR, c = (i - j * G::CPR) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (j < G::DD) {
      float w[8], s[8];
      ldf<8>(p.Wv2, (long)j * C + c, w);
      ldf<8>(chb, c, s);
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] *= 1.f + s[e];
      v = make_uint4(f2bf2(w[0], w[1]), f2bf2(w[2], w[3]), f2bf2(w[4], w[5]), f2bf2(w[6], w[7]));
    }
    *reinterpret_cast<uint4*>(w2img + j * G::PW + c * 2) = v;
  }
  {
    const int cg = C / p.g, dg = G::DS / p.g;
    for (int i = tid; i < C * 2; i += G::NTHR) {               // WdT[c][jz] (block diagonal), 16 k-columns = 2 chunks per row
      const int c = i >> 1, j0 = (i & 1) * 8;
      float w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int jz = j0 + e;
        const int jc = jz < G::DS ? jz : 0;
        const int gi = jc / dg, cl = c - gi * cg;
        const bool in = jz < G::DS && cl >= 0 && cl < cg;
        const float t = p.Wd[(long)jc * cg + (in ? cl : 0)];
        w[e] = in ? t : 0.f;
      }
      *reinterpret_cast<uint4*>(wdT + c * G::PZ + j0 * 2) = make_uint4(f2bf2(w[0], w[1]), f2bf2(w[2], w[3]), f2bf2(w[4], w[5]), f2bf2(w[6], w[7]));
    }
  }
  for (int i = tid; i < G::DDP; i += G::NTHR) {
    const int ic = i < G::DD ? i : 0;
    const float bb = p.bv2[ic], w = bf2f(p.aq2[(long)b * G::DD + ic]) * p.ws[ic];
    bv2s[i] = i < G::DD ? bb : 0.f;
    w2s[i] = i < G::DD ? w : 0.f;
  }
  {
    const float tgv = p.tg ? p.gamma * p.tg[b] : 0.f;
    for (int i = tid; i < C; i += G::NTHR) {
      mcs[i] = p.alpha * chb[i] + 1.f - p.alpha + tgv;
      lnws[i] = has_ln ? p.lnw[i] : 1.f;
      opcs[i] = 1.f + chb[i];
    }
  }
  if (tid < 16) {                                          // BatchNorm-1 backward: dx = k1 * dyb - k2 - x * k3, mask = (x * a + sh > 0)
    const int c = tid < G::DS ? tid : 0;
    float a = 1.f, sh = 0.f, k2 = 0.f, k3 = 0.f;
    if (p.has_bn) {
      a = p.bn_sc[c]; sh = p.bn_sh[c];
      if (p.training) { k3 = a * p.bn_rstd[c] * p.bn_sums[G::DS + c] * p.inv_rows; k2 = a * p.bn_sums[c] * p.inv_rows - p.bn_mean[c] * k3; }
    }
    bnv[tid] = a; bnv[16 + tid] = sh; bnv[32 + tid] = a; bnv[48 + tid] = k2; bnv[64 + tid] = k3;
  }
  char* ximg = wave0 + wave * G::WAVE_BYTES;
  char* dvimg = ximg + G::XIMG;
  char* dzimg = dvimg + G::DVIMG;
  float* tokv = reinterpret_cast<float*>(dzimg + G::DZIMG);          // [10][32]: sg | mu | rstd | sl | map | dMap | dsl | S1 | S2
  for (int i = lane; i < G::DZIMG / 16; i += 64) *reinterpret_cast<uint4*>(dzimg + i * 16) = make_uint4(0, 0, 0, 0);   // columns >= DS stay zero
  __syncthreads();

  const int nblk = p.N / 32, stride = p.wpf * G::WAVES;
  const long frame0 = (long)b * p.N;
  const float mapdot = p.dMap ? p.mapdot[b] : 0.f;
  int blk = part * G::WAVES + wave;
  fg_u32x4 nx[G::NV];
  fg_u32x2 ndz[G::NZ], nzp[G::NZ];
  float ntok[6];
  auto prefetch = [&](int bk) {                            // everything of block bk that comes from HBM (unconditional, clamped index)
    const long r0 = frame0 + (long)bk * 32;
    fg_gload<G::NV>(nx, p.X1 + r0 * C, lane);
#pragma unroll
    for (int i = 0; i < G::NZ; ++i) {
      int q = i * 64 + lane; q = q < 8 * G::DS ? q : 8 * G::DS - 1;
      ndz[i] = *reinterpret_cast<const fg_u32x2*>(p.dZ + r0 * G::DS + q * 4);
      nzp[i] = *reinterpret_cast<const fg_u32x2*>(p.Zp + r0 * G::DS + q * 4);
    }
    const long row = r0 + tl;
    ntok[0] = p.sg[row]; ntok[1] = has_ln ? p.mu[row] : 0.f; ntok[2] = has_ln ? p.rstd[row] : 1.f;
    ntok[3] = p.sl[row]; ntok[4] = p.map[row]; ntok[5] = p.dMap ? p.dMap[row] : 0.f;
  };
  prefetch(blk < nblk ? blk : nblk - 1);

  float a_lnw[G::NT], a_lnb[G::NT], a_chA[G::NT], a_chB[G::NT], a_u[G::JT], a_bv[G::JT];
#pragma unroll
  for (int i = 0; i < G::NT; ++i) { a_lnw[i] = 0.f; a_lnb[i] = 0.f; a_chA[i] = 0.f; a_chB[i] = 0.f; }
#pragma unroll
  for (int i = 0; i < G::JT; ++i) { a_u[i] = 0.f; a_bv[i] = 0.f; }
  float tsum = 0.f, bsum = 0.f;
  const float invC = 1.f / (float)C;

  for (; blk < nblk; blk += stride) {
    // The lane id is made OPAQUE once per block: every per-lane address below (a few dozen LDS / global offsets) is loop-invariant,
    // and hipcc hoists them all out of the block loop -- 100+ VGPRs of addresses, i.e. spills.  Recomputing them costs ~150 VALU
    // instructions per block.
    int lane = lane_;
    asm volatile("" : "+v"(lane));
    const int h = lane >> 5, tl = lane & 31;
    // element (token = mt_row(r, lane), channel = 32 tile + tl) of an image = ONE per-lane base + a compile-time offset
    char* const xcol = ximg + 4 * h * G::PW + tl * 2;
    char* const dvcol = dvimg + 4 * h * G::PV + tl * 2;
    const long r0 = frame0 + (long)blk * 32;
    // ---- block -> images; Xc = X1 (1 + ch) and dZp leave from the load layout (coalesced)
#pragma unroll
    for (int i = 0; i < G::NV; ++i) {
      const int idx = i * 64 + lane, r = idx / G::CPR, c8 = idx - r * G::CPR;
      *reinterpret_cast<fg_u32x4*>(ximg + r * G::PW + c8 * 16) = nx[i];
      float x[8];
      fg_unpack8(__builtin_bit_cast(bfx8, nx[i]), x);
      const float4 o0 = *reinterpret_cast<const float4*>(opcs + c8 * 8), o1 = *reinterpret_cast<const float4*>(opcs + c8 * 8 + 4);
      x[0] *= o0.x; x[1] *= o0.y; x[2] *= o0.z; x[3] *= o0.w; x[4] *= o1.x; x[5] *= o1.y; x[6] *= o1.z; x[7] *= o1.w;
      reinterpret_cast<bfx8*>(p.Xc + r0 * C)[idx] = fg_pack8(x);
    }
#pragma unroll
    for (int i = 0; i < G::NZ; ++i) {
      const int q = i * 64 + lane;
      if (q < 8 * G::DS) {
        const int e0 = q * 4, tok = e0 / G::DS, jz0 = e0 - tok * G::DS;
        const float4 va = *reinterpret_cast<const float4*>(bnv + jz0), vs = *reinterpret_cast<const float4*>(bnv + 16 + jz0);
        const float4 v2 = *reinterpret_cast<const float4*>(bnv + 48 + jz0), v3 = *reinterpret_cast<const float4*>(bnv + 64 + jz0);
        const float a[4] = {va.x, va.y, va.z, va.w}, sh[4] = {vs.x, vs.y, vs.z, vs.w};
        const float k2[4] = {v2.x, v2.y, v2.z, v2.w}, k3[4] = {v3.x, v3.y, v3.z, v3.w};
        const float dy[4] = {__uint_as_float(ndz[i].x << 16), __uint_as_float(ndz[i].x & 0xffff0000u), __uint_as_float(ndz[i].y << 16),
                             __uint_as_float(ndz[i].y & 0xffff0000u)};
        const float zx[4] = {__uint_as_float(nzp[i].x << 16), __uint_as_float(nzp[i].x & 0xffff0000u), __uint_as_float(nzp[i].y << 16),
                             __uint_as_float(nzp[i].y & 0xffff0000u)};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gg = (zx[e] * a[e] + sh[e] > 0.f) ? dy[e] : 0.f;
          o[e] = a[e] * gg - k2[e] - zx[e] * k3[e];
        }
        const uint2 w = make_uint2(f2bf2(o[0], o[1]), f2bf2(o[2], o[3]));
        *reinterpret_cast<uint2*>(p.dZ + r0 * G::DS + e0) = w;
        *reinterpret_cast<uint2*>(dzimg + tok * G::PZ + jz0 * 2) = w;
      }
    }
    if (lane < 32) {
#pragma unroll
      for (int i = 0; i < 6; ++i) tokv[i * 32 + lane] = ntok[i];
    }
    fg_wave_sync();

    // per-token scalars of the 16 tokens whose accumulator rows this lane holds (token = mt_row(r, lane)) are re-read from the
    // wave's LDS vectors tile by tile (float4 broadcasts): held in registers they are 48 VGPRs this kernel does not have
    auto tokvec = [&](int which, f32x16& v) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(tokv + which * 32 + 8 * q + 4 * h);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
      }
    };
    const bfx8 zf = fg_lds8(dzimg + tl * G::PZ + h * 16);            // dZp as the A operand (row = token, k = bottleneck channel)

    // ---- pass 1, channel tiles as the M operand: acc[channel][token], the lane's OWN token (tl) -- the per-token sums over
    // channels of the LayerNorm backward (S1 = mean gw, S2 = mean gw xh) and of dsg (T = sum_c dX2 X1, expanded into three more
    // sums) are in-lane sums + one exchange with lane ^ 32; the spatial-gate backward is then ONE evaluation per lane.
    {
      const float sg_t = tokv[tl], mu_t = tokv[32 + tl], rs_t = tokv[64 + tl];
      const float bsg = p.beta * sg_t;
      float S1 = 0.f, S2 = 0.f, A1 = 0.f, A2 = 0.f, A3 = 0.f;
      const char* xr = ximg + tl * G::PW + 8 * h;
#pragma unroll
      for (int nt = 0; nt < G::NT; ++nt) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fg_lds8(wdT + (32 * nt + tl) * G::PZ + h * 16), zf, acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c0 = 32 * nt + 8 * q + 4 * h;
          const uint2 raw = *reinterpret_cast<const uint2*>(xr + (32 * nt + 8 * q) * 2);
          const float4 m4 = *reinterpret_cast<const float4*>(mcs + c0), w4 = *reinterpret_cast<const float4*>(lnws + c0);
          const float x1[4] = {__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u), __uint_as_float(raw.y << 16),
                               __uint_as_float(raw.y & 0xffff0000u)};
          const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = (x1[e] * (mm[e] + bsg) - mu_t) * rs_t;
            const float gw = acc[4 * q + e] * ww[e];
            S1 += gw; S2 += gw * xh; A1 += gw * x1[e]; A2 += x1[e]; A3 += xh * x1[e];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      S1 += fg_xor32(S1); S2 += fg_xor32(S2); A1 += fg_xor32(A1); A2 += fg_xor32(A2); A3 += fg_xor32(A3);
      float T = A1;                                        // no ln_before: dX2 = dX3
      if (has_ln) { S1 *= invC; S2 *= invC; T = rs_t * (A1 - S1 * A2 - S2 * A3); }
      float d = p.beta * T * sg_t * (1.f - sg_t);
      if (p.dMap) { const float t = tanhf(tokv[96 + tl]); d += tokv[128 + tl] * (tokv[160 + tl] - mapdot) * (1.f - t * t); }
      if (lane < 32) { tokv[192 + tl] = d; tokv[224 + tl] = S1; tokv[256 + tl] = S2; bsum += d; tsum += T; }
    }
    fg_wave_sync();
    // ---- pass 2, token rows as the M operand: acc[token][channel], the lane's channel = 32 nt + tl -- the per-channel sums over
    // tokens (d lnw, d lnb, first part of dch) are in-lane sums over the 16 registers; dX1a = dX2 * modulation is kept packed
    unsigned pk[G::NT][8];
#pragma unroll
    for (int nt = 0; nt < G::NT; ++nt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf, fg_lds8(wdT + (32 * nt + tl) * G::PZ + h * 16), acc, 0, 0, 0);
      const int c = 32 * nt + tl;
      const float mc = mcs[c], lw = lnws[c];
      float sw = 0.f, sb = 0.f, sc = 0.f;
      float o[16];
      f32x16 sgv, muv, rsv, s1v, s2v;
      asm volatile("" ::: "memory");
      tokvec(0, sgv); tokvec(1, muv); tokvec(2, rsv); tokvec(7, s1v); tokvec(8, s2v);
      // (the ln_before flag is tested once per TILE, with two straight-line element loops behind it: tested per element it is
      //  16 branches per tile, each serialising its ds_read_u16 behind an s_waitcnt)
      float x1v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) x1v[r] = bf2f(fg_ldsu16(xcol + ((r & 3) + 8 * (r >> 2)) * G::PW + 64 * nt));
      if (has_ln) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float m = mc + p.beta * sgv[r];
          const float xh = (x1v[r] * m - muv[r]) * rsv[r];
          const float dx2 = rsv[r] * (acc[r] * lw - s1v[r] - xh * s2v[r]);
          sw += acc[r] * xh; sb += acc[r];
          sc += dx2 * x1v[r];
          o[r] = dx2 * m;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float m = mc + p.beta * sgv[r];
          sb += acc[r];
          sc += acc[r] * x1v[r];
          o[r] = acc[r] * m;
        }
      }
      a_lnw[nt] += sw; a_lnb[nt] += sb; a_chA[nt] += sc;
#pragma unroll
      for (int r = 0; r < 8; ++r) pk[nt][r] = f2bf2(o[2 * r], o[2 * r + 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // (the next block's loads are issued here, not at the top: the row-sum registers of the two passes above are dead now)
    prefetch(blk + stride < nblk ? blk + stride : nblk - 1);
    // ---- vq2 recomputed (token rows x per-frame-scaled Wv2), dvq2, u, d bv2
#pragma unroll
    for (int jt = 0; jt < G::JT; ++jt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const char* arow = ximg + tl * G::PW + h * 16;
      const char* brow = w2img + (32 * jt + tl) * G::PW + h * 16;
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fg_lds8(arow + kk * 32), fg_lds8(brow + kk * 32), acc, 0, 0, 0);
      const int j = 32 * jt + tl;
      const float bb = bv2s[j], wj = w2s[j];
      float su = 0.f, sv = 0.f;
      f32x16 dsl;
      tokvec(6, dsl);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = fmaxf(acc[r] + bb, 0.f);
        const float dv = v > 0.f ? dsl[r] * wj : 0.f;
        su += dsl[r] * v; sv += dv;
        *reinterpret_cast<unsigned short*>(dvcol + ((r & 3) + 8 * (r >> 2)) * G::PV + 64 * jt) = f2bf(dv);
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      a_u[jt] += su; a_bv[jt] += sv;
      __builtin_amdgcn_sched_barrier(0);
    }
    fg_wave_sync();
    {   // dvq2 rows leave as 16-byte stores
      constexpr int CPRV = G::DD / 8, NCH = 32 * CPRV;
      fg_u32x4* dst = reinterpret_cast<fg_u32x4*>(p.dvq2 + r0 * G::DD);
#pragma unroll
      for (int i = 0; i < (NCH + 63) / 64; ++i) {
        const int idx = i * 64 + lane;
        if (NCH % 64 == 0 || idx < NCH) {
          const int r = idx / CPRV, c8 = idx - r * CPRV;
          dst[idx] = *reinterpret_cast<const fg_u32x4*>(dvimg + r * G::PV + c8 * 16);
        }
      }
    }
    // ---- dX1 = dX1a + dvq2 W2_b  (= dXc (1 + ch)),  dch += dXc * X1; result replaces the X1 image tile by tile
    const char* drow = dvimg + tl * G::PV + h * 16;
#pragma unroll
    for (int nt = 0; nt < G::NT; ++nt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < G::DDP / 16; ++kk)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fg_lds8(drow + kk * 32), mt_frag_mn(w2img, G::PW, 32 * nt, kk, lane), acc, 0, 0, 0);
      const int c = 32 * nt + tl;
      float sc = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        char* px = xcol + ((r & 3) + 8 * (r >> 2)) * G::PW + 64 * nt;
        const float x1 = bf2f(fg_ldsu16(px));
        sc += acc[r] * x1;
        const unsigned w = pk[nt][r >> 1];
        const float a1 = (r & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
        *reinterpret_cast<unsigned short*>(px) = f2bf(a1 + acc[r]);
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      a_chB[nt] += sc;
      __builtin_amdgcn_sched_barrier(0);
    }
    fg_wave_sync();
    {
      fg_u32x4* dst = reinterpret_cast<fg_u32x4*>(p.dX1 + r0 * C);
#pragma unroll
      for (int i = 0; i < G::NV; ++i) {
        const int idx = i * 64 + lane, r = idx / G::CPR, c8 = idx - r * G::CPR;
        dst[idx] = *reinterpret_cast<const fg_u32x4*>(ximg + r * G::PW + c8 * 16);
      }
    }
    fg_wave_sync();
  }

  // ---- per-channel sums: the two token halves of a wave, then the 8 waves through LDS, then out
  __syncthreads();
  float* red = reinterpret_cast<float*>(wave0);             // [WAVES][NQ]
  {
    float* rw = red + wave * G::NQ;
#pragma unroll
    for (int nt = 0; nt < G::NT; ++nt) {
      const float v0 = a_lnw[nt] + fg_xor32(a_lnw[nt]), v1 = a_lnb[nt] + fg_xor32(a_lnb[nt]);
      const float v2 = a_chA[nt] + fg_xor32(a_chA[nt]), v3 = a_chB[nt] + fg_xor32(a_chB[nt]);
      if (lane < 32) { rw[32 * nt + tl] = v0; rw[C + 32 * nt + tl] = v1; rw[2 * C + 32 * nt + tl] = v2; rw[3 * C + 32 * nt + tl] = v3; }
    }
#pragma unroll
    for (int jt = 0; jt < G::JT; ++jt) {
      const float v0 = a_u[jt] + fg_xor32(a_u[jt]), v1 = a_bv[jt] + fg_xor32(a_bv[jt]);
      if (lane < 32) { rw[4 * C + 32 * jt + tl] = v0; rw[4 * C + G::DDP + 32 * jt + tl] = v1; }
    }
    tsum = group_sum(tsum, 64); bsum = group_sum(bsum, 64);
    if (lane == 0) { misc[wave] = tsum; misc[8 + wave] = bsum; }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < G::NK; ++k) {
    const int i = tid + k * G::NTHR;
    if (i < G::NQ) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < G::WAVES; ++w) s += red[w * G::NQ + i];
      if (i < 2 * C) keep[k] += s;                                                   // d lnw | d lnb
      else if (i < 3 * C) unsafeAtomicAdd(p.dch + (long)b * C + (i - 2 * C), p.alpha * s);
      else if (i < 4 * C) unsafeAtomicAdd(p.dch + (long)b * C + (i - 3 * C), s / opcs[i - 3 * C]);
      else if (i < 4 * C + G::DDP) { const int j = i - 4 * C; if (j < G::DD) unsafeAtomicAdd(p.u + (long)b * G::DD + j, s); }
      else keep[k] += s;                                                             // d bv2
    }
  }
  if (tid == 0) {
    float t = 0.f, bs = 0.f;
#pragma unroll
    for (int w = 0; w < G::WAVES; ++w) { t += misc[w]; bs += misc[8 + w]; }
    keep_bs += bs;
    if (p.dtg) unsafeAtomicAdd(p.dtg + b, p.gamma * t);
  }
  }   // items
  constexpr int L = 2 * C + G::DD + 1;
  float* prow = p.part + (long)blockIdx.x * L;
#pragma unroll
  for (int k = 0; k < G::NK; ++k) {
    const int i = tid + k * G::NTHR;
    if (i < 2 * C) prow[i] = keep[k];
    else if (i >= 4 * C + G::DDP && i < G::NQ) { const int j = i - 4 * C - G::DDP; if (j < G::DD) prow[2 * C + j] = keep[k]; }
  }
  if (tid == 0) prow[2 * C + G::DD] = keep_bs;
}

// mapdot[b] = sum_n map[b][n] * dMap[b][n]   (softmax-backward inner product of the returned map; last layer only)
__global__ __launch_bounds__(256) void mapdot_k(const float* map, const float* dMap, int N, float* out) {
  __shared__ float red[4];
  const long o = (long)blockIdx.x * N;
  float pd = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) pd += map[o + n] * dMap[o + n];
  pd = block_sum(pd, red);
  if (threadIdx.x == 0) out[blockIdx.x] = pd;
}
// grads += sum over workgroups of part rows [d lnw | d lnb | d bv2 | d bs]
struct GFin { const float* part; int nwg, L, C, DD; float* dlnw; float* dlnb; float* dbv2; float* dbs; };
__global__ __launch_bounds__(256) void gate_bwd_finish_k(const GFin p) {
  // 64 columns x 4 row slices per workgroup
  __shared__ float red[4][64];
  const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + col;
  float s0 = 0.f, s1 = 0.f;
  if (i < p.L) {
    int k = sl;
    for (; k + 4 < p.nwg; k += 8) { s0 += p.part[(long)k * p.L + i]; s1 += p.part[(long)(k + 4) * p.L + i]; }
    for (; k < p.nwg; k += 4) s0 += p.part[(long)k * p.L + i];
  }
  red[sl][col] = s0 + s1;
  __syncthreads();
  if (sl != 0 || i >= p.L) return;
  const float s = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
  float* dst = i < p.C ? (p.dlnw ? p.dlnw + i : nullptr) : i < 2 * p.C ? (p.dlnb ? p.dlnb + (i - p.C) : nullptr)
             : i < 2 * p.C + p.DD ? (p.dbv2 ? p.dbv2 + (i - 2 * p.C) : nullptr) : p.dbs;
  if (dst) *dst += s;
}

// =====================================================================================================================
// Channel-gate query vq1 = relu(X1 Wv1^T + bv1) without the tensor (net_trans.py:593-594 and their autograd), C in {96, 128}.
// The forward only wants mean_N vq1 per frame and the backward only  dvq1 = (vq1 > 0) * coef_b / N  -- a [rows, C] tensor written,
// re-read by a column sum, re-read and rewritten by the ReLU backward and re-read twice by the two products behind it (126 MB a
// pass at stage 0).  Here:
//   vq1_fwd_k: one pass over X1: the product in token-rows-as-M orientation (the lane holds an output CHANNEL: the sum over tokens
//              is an in-lane sum of the 16 accumulator registers), ReLU, per-(frame, channel) sums.  Nothing else is stored.
//   vq1_bwd_k: one pass over X1 and dX1: the same product again (same operands, same MFMA order: same bits, same ReLU decisions),
//              dvq1 = mask * bf16(coef_b / N), d bv1 += dvq1, dX1 += dvq1 Wv1 in place (the weight image read MN-major), and dvq1
//              leaves once for the dWv1 product on the aux stream.
template <int C_>
struct VG {
  static constexpr int C = C_, KS = C / 16, NT = C / 32, PW = C * 2 + 16, NV = C / 16, CPR = C / 8;
  static constexpr int W_BYTES = C * PW, XIMG = 32 * PW;
  static constexpr int SMEM_F = W_BYTES + 4 * XIMG + (C + 4 * C) * 4;
  static constexpr int SMEM_B = W_BYTES + 8 * XIMG + (2 * C + 4 * C) * 4;
  static_assert(SMEM_B <= 160 * 1024, "LDS budget");
  static_assert(8 * XIMG >= C * C * 4, "the wave images double as the dWv1 reduction scratch");
};
struct VF { const unsigned short* X1; const unsigned short* W; const float* bias; int N, wpf; float invN; float* msum; unsigned short* vq1; };
struct VB {
  const unsigned short* X1; const unsigned short* W; const float* bias; const float* coef; int B, N, wpf; float invN;
  unsigned short* dX1; unsigned short* dvq1; float* part;
  float* wpart;        // DW: [workgroup][C][C] partial dWv1 (dvq1 is then not written at all)
};

template <int C>
__device__ __forceinline__ void vg_build(char* wimg, float* bs, const unsigned short* W, const float* bias, int tid) {
  using G = VG<C>;
  for (int i = tid; i < C * G::CPR; i += 256) {
    const int j = i / G::CPR, c8 = i - j * G::CPR;
    *reinterpret_cast<uint4*>(wimg + j * G::PW + c8 * 16) = *reinterpret_cast<const uint4*>(W + (long)j * C + c8 * 8);
  }
  for (int i = tid; i < C; i += 256) bs[i] = bias[i];
}

template <int C>
__global__ __launch_bounds__(256) void vq1_fwd_k(const VF p) {
  using G = VG<C>;
  __shared__ __attribute__((aligned(16))) char smem[G::SMEM_F];
  char* wimg = smem;
  char* ximg0 = wimg + G::W_BYTES;
  float* bs = reinterpret_cast<float*>(ximg0 + 4 * G::XIMG);
  float* red = bs + C;                                       // [4][C]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, tl = lane & 31;
  const int b = blockIdx.y;
  vg_build<C>(wimg, bs, p.W, p.bias, tid);
  __syncthreads();
  const int nblk = p.N / 32, stride = p.wpf * 4;
  char* img = ximg0 + wave * G::XIMG;
  const long frame0 = (long)b * p.N;
  fg_u32x4 nx[G::NV];
  int blk = blockIdx.x * 4 + wave;
  fg_gload<G::NV>(nx, p.X1 + (frame0 + (long)(blk < nblk ? blk : nblk - 1) * 32) * C, lane);
  float a_s[G::NT], bb[G::NT];
#pragma unroll
  for (int jt = 0; jt < G::NT; ++jt) { a_s[jt] = 0.f; bb[jt] = bs[32 * jt + tl]; }
  for (; blk < nblk; blk += stride) {
#pragma unroll
    for (int i = 0; i < G::NV; ++i) {
      const int idx = i * 64 + lane, r = idx / G::CPR, c8 = idx - r * G::CPR;
      *reinterpret_cast<fg_u32x4*>(img + r * G::PW + c8 * 16) = nx[i];
    }
    fg_wave_sync();
    fg_gload<G::NV>(nx, p.X1 + (frame0 + (long)(blk + stride < nblk ? blk + stride : nblk - 1) * 32) * C, lane);
    const char* arow = img + tl * G::PW + h * 16;
#pragma unroll
    for (int jt = 0; jt < G::NT; ++jt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const char* brow = wimg + (32 * jt + tl) * G::PW + h * 16;
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fg_lds8(arow + kk * 32), fg_lds8(brow + kk * 32), acc, 0, 0, 0);
      float sv = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sv += fmaxf(acc[r] + bb[jt], 0.f);
      a_s[jt] += sv;
      if (p.vq1) {                                           // test mode ("vq1fuse" = 2): the tensor as well, for tests that pin the ReLU masks
        const long row0 = frame0 + (long)blk * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) p.vq1[(row0 + mt_row(r, lane)) * C + 32 * jt + tl] = f2bf(fmaxf(acc[r] + bb[jt], 0.f));
      }
    }
    fg_wave_sync();
  }
#pragma unroll
  for (int jt = 0; jt < G::NT; ++jt) {
    const float v = a_s[jt] + fg_xor32(a_s[jt]);
    if (lane < 32) red[wave * C + 32 * jt + tl] = v;
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256)
    unsafeAtomicAdd(p.msum + (long)b * C + c, p.invN * ((red[c] + red[C + c]) + (red[2 * C + c] + red[3 * C + c])));
}

// DW (experiment, "vq1fuse" = 3, off by default: slower, see plan.cpp B5): dWv1 = dvq1^T X1 accumulated HERE (both operands read
// MN-major from the block's two images, C x C fp32 accumulators per wave: 144 / 256 registers) -- dvq1 never reaches HBM and the
// weight-gradient product of the aux stream disappears.
template <int C, bool DW>
__global__ __launch_bounds__(256) void vq1_bwd_k(const VB p) {
  using G = VG<C>;
  __shared__ __attribute__((aligned(16))) char smem[G::SMEM_B];
  char* wimg = smem;
  char* wave0 = wimg + G::W_BYTES;
  float* bs = reinterpret_cast<float*>(wave0 + 8 * G::XIMG);
  float* cn = bs + C;
  float* red = cn + C;                                       // [4][C]
  const int tid = threadIdx.x, lane_ = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  vg_build<C>(wimg, bs, p.W, p.bias, tid);
  // work items = (frame, part of the frame), one workgroup per CU walking items blockIdx.x, + gridDim.x, ... (as gatemod_bwd_k)
  const int nitems = p.B * p.wpf;
  float keep = 0.f;                                          // thread c < C: d bv1[c] over this workgroup's items
  f32x16 accw[DW ? G::NT : 1][DW ? G::NT : 1];               // dWv1 tile (c_out tile, c_in tile), over all of this wave's blocks
#pragma unroll
  for (int i = 0; i < (DW ? G::NT : 1); ++i)
#pragma unroll
    for (int j = 0; j < (DW ? G::NT : 1); ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) accw[i][j][r] = 0.f;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
  const int b = item / p.wpf, part = item - b * p.wpf;
  __syncthreads();                                           // the previous item's reduction has read cn / red (and wimg is built)
  for (int i = tid; i < C; i += 256) cn[i] = bf2f(f2bf(p.invN * p.coef[(long)b * C + i]));      // dvq1's one non-zero value per (frame, channel), as stored
  __syncthreads();
  const int nblk = p.N / 32, stride = p.wpf * 4;
  char* ximg = wave0 + wave * 2 * G::XIMG;
  char* dvimg = ximg + G::XIMG;
  const long frame0 = (long)b * p.N;
  fg_u32x4 nx[G::NV], ndx[G::NV];
  int blk = part * 4 + wave;
  {
    const long r0 = frame0 + (long)(blk < nblk ? blk : nblk - 1) * 32;
    fg_gload<G::NV>(nx, p.X1 + r0 * C, lane_);
    fg_gload<G::NV>(ndx, p.dX1 + r0 * C, lane_);
  }
  float a_bv[G::NT];
#pragma unroll
  for (int jt = 0; jt < G::NT; ++jt) a_bv[jt] = 0.f;
  for (; blk < nblk; blk += stride) {
    int lane = lane_;
    asm volatile("" : "+v"(lane));                           // (per-lane addresses recomputed per block instead of hoisted into 100 VGPRs: see gatemod_bwd_k)
    const int h = lane >> 5, tl = lane & 31;
    char* const xcol = ximg + 4 * h * G::PW + tl * 2;
    char* const dvcol = dvimg + 4 * h * G::PW + tl * 2;
    const long r0 = frame0 + (long)blk * 32;
#pragma unroll
    for (int i = 0; i < G::NV; ++i) {
      const int idx = i * 64 + lane, r = idx / G::CPR, c8 = idx - r * G::CPR;
      *reinterpret_cast<fg_u32x4*>(ximg + r * G::PW + c8 * 16) = nx[i];
    }
    fg_wave_sync();
    // ---- vq1 again (token rows x Wv1 rows: the forward's product), dvq1 = mask * cn, d bv1
    const char* arow = ximg + tl * G::PW + h * 16;
#pragma unroll
    for (int jt = 0; jt < G::NT; ++jt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const char* brow = wimg + (32 * jt + tl) * G::PW + h * 16;
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fg_lds8(arow + kk * 32), fg_lds8(brow + kk * 32), acc, 0, 0, 0);
      const int c = 32 * jt + tl;
      const float bb = bs[c], cv = cn[c];
      const unsigned short cvb = f2bf(cv);
      float sv = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool on = acc[r] + bb > 0.f;
        sv += on ? cv : 0.f;
        *reinterpret_cast<unsigned short*>(dvcol + ((r & 3) + 8 * (r >> 2)) * G::PW + 64 * jt) = on ? cvb : (unsigned short)0;
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      a_bv[jt] += sv;
      __builtin_amdgcn_sched_barrier(0);
    }
    fg_wave_sync();
    if (DW) {                                                // dWv1 += dvq1^T X1: contraction over the block's 32 tokens (two k-steps)
      // (operand fragments re-read per tile: held for the whole block they are 2 x 8 NT registers next to NT^2 x 16 accumulators --
      //  at C = 128 that spilled)
#pragma unroll
      for (int jt = 0; jt < G::NT; ++jt) {
        const bfx8 a0 = mt_frag_mn(dvimg, G::PW, 32 * jt, 0, lane), a1 = mt_frag_mn(dvimg, G::PW, 32 * jt, 1, lane);
#pragma unroll
        for (int nt = 0; nt < G::NT; ++nt) {
          const bfx8 b0 = mt_frag_mn(ximg, G::PW, 32 * nt, 0, lane), b1 = mt_frag_mn(ximg, G::PW, 32 * nt, 1, lane);
          accw[jt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, accw[jt][nt], 0, 0, 0);
          accw[jt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, accw[jt][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      fg_wave_sync();                                        // every lane is done with X1's image
    }
    // X1's image is dead: the block's dX1 rows take its place; dvq1 rows leave as 16-byte stores (unless dWv1 is made here)
    {
      fg_u32x4* dst = reinterpret_cast<fg_u32x4*>(p.dvq1 + r0 * C);
#pragma unroll
      for (int i = 0; i < G::NV; ++i) {
        const int idx = i * 64 + lane, r = idx / G::CPR, c8 = idx - r * G::CPR;
        if (!DW) dst[idx] = *reinterpret_cast<const fg_u32x4*>(dvimg + r * G::PW + c8 * 16);
        *reinterpret_cast<fg_u32x4*>(ximg + r * G::PW + c8 * 16) = ndx[i];
      }
    }
    fg_wave_sync();
    {
      const long rn = frame0 + (long)(blk + stride < nblk ? blk + stride : nblk - 1) * 32;
      fg_gload<G::NV>(nx, p.X1 + rn * C, lane);
      fg_gload<G::NV>(ndx, p.dX1 + rn * C, lane);
    }
    // ---- dX1 += dvq1 Wv1
    const char* drow = dvimg + tl * G::PW + h * 16;
#pragma unroll
    for (int nt = 0; nt < G::NT; ++nt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fg_lds8(drow + kk * 32), mt_frag_mn(wimg, G::PW, 32 * nt, kk, lane), acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        char* px = xcol + ((r & 3) + 8 * (r >> 2)) * G::PW + 64 * nt;
        *reinterpret_cast<unsigned short*>(px) = f2bf(bf2f(fg_ldsu16(px)) + acc[r]);
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    fg_wave_sync();
    {
      fg_u32x4* dst = reinterpret_cast<fg_u32x4*>(p.dX1 + r0 * C);
#pragma unroll
      for (int i = 0; i < G::NV; ++i) {
        const int idx = i * 64 + lane, r = idx / G::CPR, c8 = idx - r * G::CPR;
        dst[idx] = *reinterpret_cast<const fg_u32x4*>(ximg + r * G::PW + c8 * 16);
      }
    }
    fg_wave_sync();
  }
  {
    const int tl = lane_ & 31;
#pragma unroll
    for (int jt = 0; jt < G::NT; ++jt) {
      const float v = a_bv[jt] + fg_xor32(a_bv[jt]);
      if (lane_ < 32) red[wave * C + 32 * jt + tl] = v;
    }
  }
  __syncthreads();
  if (tid < C) keep += (red[tid] + red[C + tid]) + (red[2 * C + tid] + red[3 * C + tid]);
  }   // items
  if (tid < C) p.part[(long)blockIdx.x * C + tid] = keep;
  if (DW) {   // the four waves' C x C tiles meet in LDS (the wave images are free now), one partial matrix per workgroup leaves
    float* wred = reinterpret_cast<float*>(wave0);
    const int h = lane_ >> 5, tl = lane_ & 31;
    __syncthreads();
    for (int i = tid; i < C * C / 4; i += 256) reinterpret_cast<float4*>(wred)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
#pragma unroll
    for (int jt = 0; jt < G::NT; ++jt)
#pragma unroll
      for (int nt = 0; nt < G::NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r)                         // (LDS float add: the four waves in any order)
          atomicAdd(wred + (32 * jt + (r & 3) + 8 * (r >> 2) + 4 * h) * C + 32 * nt + tl, accw[jt][nt][r]);
        __builtin_amdgcn_sched_barrier(0);
      }
    __syncthreads();
    float4* dst = reinterpret_cast<float4*>(p.wpart + (long)blockIdx.x * C * C);
    for (int i = tid; i < C * C / 4; i += 256) dst[i] = reinterpret_cast<const float4*>(wred)[i];
  }
}

std::atomic<int> g_vq1fuse{-1};
std::atomic<int> g_gatefuse{-1};
}  // namespace

int vq1fuse_mode(int set) {
  if (g_vq1fuse.load(std::memory_order_relaxed) < 0) g_vq1fuse.store(getenv("DGSCT_NO_VQ1FUSE") ? 0 : 1, std::memory_order_relaxed);
  const int old = g_vq1fuse.load(std::memory_order_relaxed);
  if (set >= 0) g_vq1fuse.store(set > 3 ? 1 : set, std::memory_order_relaxed);      // 2: fused, and the forward also stores vq1 (tests); 3: dWv1 accumulated inside vq1_bwd_k (experiment, slower)
  return old;
}
bool vq1_fused_shape(int mode, int N, int C) { return mode == DT_BF16 && (C == 96 || C == 128) && N >= 32 && N % 32 == 0; }
bool vq1_fused_supported(int mode, int N, int C) { return vq1fuse_mode(-1) && vq1_fused_shape(mode, N, C); }
static int vq1_wpf(const void* kern, int B, int N) {
  int cap = wg_capacity(kern, 0);
  int wpf = cap / B;
  const int maxw = (N / 32 + 7) / 8;                         // >= 2 blocks per wave
  if (wpf > maxw) wpf = maxw;
  return wpf < 1 ? 1 : wpf;
}
void vq1sum_fwd(const Ctx& ctx, const void* X1, const void* Wv1, const float* bv1, int B, int N, int C, float invN, float* msum, void* vq1) {
  VF a{(const unsigned short*)X1, (const unsigned short*)Wv1, bv1, N, 1, invN, msum, (unsigned short*)vq1};
  auto launch = [&](auto kern) {
    a.wpf = vq1_wpf(reinterpret_cast<const void*>(kern), B, N);
    hipLaunchKernelGGL(kern, dim3(a.wpf, B), dim3(256), 0, (hipStream_t)ctx.stream, a);
  };
  if (C == 96) launch(vq1_fwd_k<96>); else if (C == 128) launch(vq1_fwd_k<128>); else set_error("vq1sum_fwd: unsupported width %d", C);
}
long vq1_wpart_floats(int C) { return (long)512 * C * C; }
void vq1_bwd(const Ctx& ctx, const void* X1, const void* Wv1, const float* bv1, const float* coef, int B, int N, int C, float invN,
             void* dX1, void* dvq1, float* dbv1, float* part, long part_floats, float* dWv1, float* wpart) {
  VB a{(const unsigned short*)X1, (const unsigned short*)Wv1, bv1, coef, B, N, 1, invN, (unsigned short*)dX1, (unsigned short*)dvq1, part, wpart};
  const bool dw = dWv1 != nullptr;
  if (!dw && !dvq1) { set_error("vq1_bwd: neither dvq1 nor dWv1 requested"); return; }
  auto launch = [&](auto kern) {
    // parts per frame: >= 2 blocks per wave of an item, the items filling whole rounds of the resident workgroups (gatemod_bwd's rule)
    const int cap = wg_capacity(reinterpret_cast<const void*>(kern), 0);
    int maxw = (N / 32) / 8; if (maxw < 1) maxw = 1;
    int best = 1; double be = -1;
    for (int w = 1; w <= maxw; ++w) {
      const long items = (long)B * w;
      const double e = (double)items / (double)(((items + cap - 1) / cap) * cap);
      if (e > be + 1e-9) { be = e; best = w; }
    }
    a.wpf = best;
    const long items = (long)B * best;
    const int nwg = (int)(items < cap ? items : cap);
    if ((long)nwg * C > part_floats || (dw && nwg > 512)) { set_error("vq1_bwd: partial-sum scratch too small"); return; }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, (hipStream_t)ctx.stream, a);
    PartJob j; j.part = part; j.n = 1; j.t.NQ = 1; j.t.C = C;
    j.t.d[0] = PartDesc{0, 1, nwg, 1, dbv1, 0, 1.f};
    if (dw) {                                                // second job: the C x C partial matrices, as one row of C * C "channels"
      j.part2 = wpart; j.n2 = 1; j.t2.NQ = 1; j.t2.C = C * C;
      j.t2.d[0] = PartDesc{0, 1, nwg, 1, dWv1, 0, 1.f};
    }
    if (ctx.late) *ctx.late = j;                             // parameter gradients: their second stages may run on the aux stream
    else part_reduce_run(ctx.stream, j);
  };
  if (C == 96) { if (dw) launch(vq1_bwd_k<96, true>); else launch(vq1_bwd_k<96, false>); }
  else if (C == 128) { if (dw) launch(vq1_bwd_k<128, true>); else launch(vq1_bwd_k<128, false>); }
  else set_error("vq1_bwd: unsupported width %d", C);
}


int gatefuse_mode(int set) {
  if (g_gatefuse.load(std::memory_order_relaxed) < 0) g_gatefuse.store(getenv("DGSCT_NO_GATEFUSE") ? 0 : 1, std::memory_order_relaxed);
  const int old = g_gatefuse.load(std::memory_order_relaxed);
  if (set >= 0) g_gatefuse.store(set > 2 ? 1 : set, std::memory_order_relaxed);
  return old;
}

static bool gate_shape_ok(int mode, int N, int C, int ds, int g) {
  if (mode != DT_BF16) return false;
  if (!(C == 96 || C == 128 || C == 192 || C == 256)) return false;
  if (N % 32 || N < 32 || g < 1 || ds < 4 || ds > 32 || ds % 4 || ds % g || C % g) return false;
  return true;
}
bool gate_fused_supported(int mode, int N, int C, int ds, int g) { return gatefuse_mode(-1) && gate_shape_ok(mode, N, C, ds, g); }
bool gate_bwd_fused_shape(int mode, int N, int C, int ds, int g) {
  return gate_shape_ok(mode, N, C, ds, g) && (C == 96 || C == 128) && ds == C / 8;
}

void gatemod_fwd(const Ctx& ctx, const void* X1, const float* ch, const void* aq2, const float* Wv2, const float* bv2, const float* ws,
                 const float* bs, const float* tg, float alpha, float beta, float gamma, const float* lnw, const float* lnb, float eps,
                 int B, int N, int C, int ds, int g, const float* Wd, float* sl, void* X3, float* mu, float* rstd, void* Zp,
                 float* stats, void* vq2) {
  GF a{(const unsigned short*)X1, ch, (const unsigned short*)aq2, Wv2, bv2, ws, bs, tg, alpha, beta, gamma, lnw, lnb, eps,
       N, ds, g, 1, Wd, sl, (unsigned short*)X3, mu, rstd, (unsigned short*)Zp, stats, (unsigned short*)vq2};
  const int nblk = N / 32;
  auto launch = [&](auto kern) {
    int cap = wg_capacity(reinterpret_cast<const void*>(kern), 0);
    int wpf = cap / B;
    const int maxw = (nblk + 3) / 4;
    if (wpf > maxw) wpf = maxw;
    if (wpf < 1) wpf = 1;
    a.wpf = wpf;
    hipLaunchKernelGGL(kern, dim3(wpf, B), dim3(256), 0, (hipStream_t)ctx.stream, a);
  };
  switch (C) {
    case 96: launch(gatemod_fwd_k<96>); break;
    case 128: launch(gatemod_fwd_k<128>); break;
    case 192: launch(gatemod_fwd_k<192>); break;
    case 256: launch(gatemod_fwd_k<256>); break;
    default: set_error("gatemod_fwd: unsupported width %d", C);
  }
}


bool gate_bwd_fused_supported(int mode, int N, int C, int ds, int g) {
  return gatefuse_mode(-1) && gate_bwd_fused_shape(mode, N, C, ds, g);
}

long gate_bwd_part_floats(int B, int C) { return (long)(1024 + B) * (2 * C + C / 2 + 1); }

void gatemod_bwd(const Ctx& ctx, const void* X1, const float* ch, const void* aq2, const float* Wv2, const float* bv2, const float* ws,
                 const float* tg, float alpha, float beta, float gamma, const float* lnw, const float* mu, const float* rstd,
                 const float* sl, const float* sg, const float* map, const float* dMap, int B, int N, int C, int ds, int g, const float* Wd,
                 void* dZ, const void* Zp, const float* bn_mean, const float* bn_rstd, const float* bn_sc, const float* bn_sh,
                 const float* bn_sums, int has_bn, int training, void* dX1, void* dvq2, void* Xc, float* dch, float* u, float* dtg,
                 float* dlnw, float* dlnb, float* dbv2, float* dbs, float* mapdot_scratch, float* part, long part_floats) {
  hipStream_t st = (hipStream_t)ctx.stream;
  if (dMap) hipLaunchKernelGGL(mapdot_k, dim3(B), dim3(256), 0, st, map, dMap, N, mapdot_scratch);
  GB a{(const unsigned short*)X1, ch, (const unsigned short*)aq2, Wv2, bv2, ws, tg, alpha, beta, gamma, lnw, mu, rstd, sl, sg, map, dMap,
       mapdot_scratch, B, N, g, 1, Wd, (unsigned short*)dZ, (const unsigned short*)Zp, bn_mean, bn_rstd, bn_sc, bn_sh, bn_sums, has_bn,
       training, 1.f / ((float)B * (float)N), (unsigned short*)dX1, (unsigned short*)dvq2, (unsigned short*)Xc, dch, u, tg ? dtg : nullptr, part};
  const int nblk = N / 32;
  const int L = 2 * C + C / 2 + 1;
  int nwg = 0;
  auto launch = [&](auto kern) {
    // parts per frame: every wave of an item should get >= 2 blocks, and the items should fill whole rounds of the grid
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int maxw = nblk / 8; if (maxw < 1) maxw = 1;
    int best = 1; double be = -1;
    for (int w = 1; w <= maxw; ++w) {
      const long items = (long)B * w;
      const double e = (double)items / (double)(((items + cus - 1) / cus) * cus);
      if (e > be + 1e-9) { be = e; best = w; }
    }
    a.wpf = best;
    const long items = (long)B * best;
    nwg = (int)(items < cus ? items : cus);
    if ((long)nwg * L > part_floats) { set_error("gatemod_bwd: partial-sum scratch too small for %d workgroups", nwg); nwg = 0; return; }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, st, a);
  };
  if ((long)1024 * L > part_floats) { set_error("gatemod_bwd: partial-sum scratch too small"); return; }
  switch (C) {
    case 96: if (lnw) launch(gatemod_bwd_k<96, true>); else launch(gatemod_bwd_k<96, false>); break;
    case 128: if (lnw) launch(gatemod_bwd_k<128, true>); else launch(gatemod_bwd_k<128, false>); break;
    default: set_error("gatemod_bwd: unsupported width %d", C); return;
  }
  if (nwg <= 0) return;
  GFin f{part, nwg, L, C, C / 2, dlnw, dlnb, dbv2, dbs};
  hipLaunchKernelGGL(gate_bwd_finish_k, dim3((L + 63) / 64), dim3(256), 0, st, f);
}

}  // namespace dgsct
