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

typedef unsigned fg_u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 is a struct: arrays of it are copied with memcpy and stay in scratch)
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

std::atomic<int> g_gatefuse{-1};
}  // namespace

int gatefuse_mode(int set) {
  if (g_gatefuse.load(std::memory_order_relaxed) < 0) g_gatefuse.store(getenv("DGSCT_NO_GATEFUSE") ? 0 : 1, std::memory_order_relaxed);
  const int old = g_gatefuse.load(std::memory_order_relaxed);
  if (set >= 0) g_gatefuse.store(set ? 1 : 0, std::memory_order_relaxed);
  return old;
}

bool gate_fused_supported(int mode, int N, int C, int ds, int g) {
  if (!gatefuse_mode(-1) || mode != DT_BF16) return false;
  if (!(C == 96 || C == 128 || C == 192 || C == 256)) return false;
  if (N % 32 || N < 32 || g < 1 || ds < 4 || ds > 32 || ds % 4 || ds % g || C % g) return false;
  return true;
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

}  // namespace dgsct
