// Wave-centric bf16 fast path of the fused latent-token attention (see attn.hip for the math and for the generic
// kernels, which stay the fp32 parity path and the fallback for shapes this file does not take).
//
// Here a WAVEFRONT, not a workgroup, owns a block of 32 token rows: it stages its own rows through a private LDS
// buffer, takes the latent-token operands straight from global memory (they are tiny, L2-resident, and PRE-PACKED as
// bf16 hi / lo / transposed images by their producer), keeps the probabilities in registers and feeds them to the
// second product as MFMA operands without a round trip (the contraction index of an MFMA is free to be permuted as
// long as both operands agree -- the packed transposed image is stored in exactly that order).  There is no
// __syncthreads in the token loops: 16 independent wavefronts per CU overlap each other's loads and MFMAs.
//
// Packed latent tokens ("tokpk", per frame or per parameter): [hi 32 x C | lo 32 x C | T C x 32] bf16, rows >= tk zero,
//   T[c][perm(t)] = hi[t][c] with perm(16k + 4h + 8q + e) = 16k + 8h + 4q + e   (k, h, q in {0,1}, e < 4).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "prims.h"
#include "device_util.h"
#include "mma_tile.h"
#include "err.h"

namespace dgsct {

namespace {
typedef mt_bf16x8 bfx8;
typedef mt_f32x16 f32x16;
constexpr int CS2 = 128;                       // channels per slab
constexpr int PX2 = CS2 * 2 + 16;              // pitch of a private [32][CS2] bf16 image (272 B: conflict-free b128 rows)
constexpr int IMG2 = 32 * PX2;                 // 8704 B per image

__host__ __device__ inline int perm_pos(int t) { return (t & 16) | (((t >> 2) & 1) << 3) | (((t >> 3) & 1) << 2) | (t & 3); }
__device__ __forceinline__ float xor32b(float v) { return __shfl_xor(v, 32, 64); }
// LDS traffic between lanes of ONE wavefront: the hardware executes a wave's LDS instructions in order; this only stops
// the compiler from moving accesses across the hand-off
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ bfx8 ldg8(const unsigned short* p) { return __builtin_bit_cast(bfx8, *reinterpret_cast<const uint4*>(p)); }
__device__ __forceinline__ bfx8 lds8(const char* p) { return *reinterpret_cast<const bfx8*>(p); }
__device__ __forceinline__ unsigned pack2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }
// accumulator registers [8k, 8k + 8) of a [latent token x token] tile as the bf16 B-operand fragment of k-step k
__device__ __forceinline__ bfx8 regs_frag(const float* v) {
  uint4 u = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
  return __builtin_bit_cast(bfx8, u);
}

// 32 rows x CS2 channels (bf16) global -> registers (8 x 16 B per lane, coalesced: 16 lanes per 256-byte row)
struct Slab { uint4 v[8]; };
__device__ __forceinline__ void slab_load(Slab& s, const unsigned short* g, long ld, int rows_valid, int c0, int C, int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = i * 64 + lane, r = idx >> 4, c = (idx & 15) * 8;
    const int rr = r < rows_valid ? r : rows_valid - 1;
    const int cc = c0 + c < C ? c0 + c : C - 8;
    s.v[i] = *reinterpret_cast<const uint4*>(g + (long)rr * ld + cc);
  }
}
__device__ __forceinline__ void slab_store_lds(const Slab& s, char* img, int rows_valid, int c0, int C, int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = i * 64 + lane, r = idx >> 4, c = (idx & 15) * 8;
    uint4 v = s.v[i];
    if (r >= rows_valid || c0 + c >= C) v = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(img + r * PX2 + c * 2) = v;
  }
}
// private image rows -> token-major global rows (16-byte stores); `add` (same layout as dst) optional
__device__ __forceinline__ void slab_copy_out(const char* img, unsigned short* dst, const unsigned short* add, long ld, int rows_valid,
                                              int c0, int C, int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = i * 64 + lane, r = idx >> 4, c = (idx & 15) * 8;
    if (r >= rows_valid || c0 + c >= C) continue;
    uint4 v = *reinterpret_cast<const uint4*>(img + r * PX2 + c * 2);
    const long o = (long)r * ld + c0 + c;
    if (add) {
      float x[8], y[8];
      unpack<DT_BF16, 8>(v, x);
      unpack<DT_BF16, 8>(*reinterpret_cast<const uint4*>(add + o), y);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] += y[e];
      stv<DT_BF16, 8>(dst, o, x);
    } else {
      *reinterpret_cast<uint4*>(dst + o) = v;
    }
  }
}
// acc[t][n] += sum_c tok[t][c] * X[n][c] over one slab: A fragments (latent tokens, hi and optionally lo) from global,
// B fragments (the wave's token rows) from its private image
template <bool LO>
__device__ __forceinline__ void logits_slab(f32x16& acc, const unsigned short* th, const unsigned short* tl, long C, int c0, int kc,
                                            const char* img, int lane) {
  const unsigned short* ah = th + (long)(lane & 31) * C + c0 + (lane >> 5) * 8;
  const unsigned short* al = tl + (long)(lane & 31) * C + c0 + (lane >> 5) * 8;
  const char* bp = img + (lane & 31) * PX2 + (lane >> 5) * 16;
  for (int kk = 0; kk < kc; ++kk) {
    const bfx8 bf = lds8(bp + kk * 32);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ldg8(ah + kk * 16), bf, acc, 0, 0, 0);
    if (LO) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ldg8(al + kk * 16), bf, acc, 0, 0, 0);
  }
}
// softmax over the latent tokens of a [t][n] accumulator tile (t along the registers + lane ^ 32); tk valid tokens
__device__ __forceinline__ void softmax_regs(f32x16& a, int tk, int lane) {
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (mt_row(r, lane) < tk) mx = fmaxf(mx, a[r]);
  mx = fmaxf(mx, xor32b(mx));
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) { a[r] = mt_row(r, lane) < tk ? __expf(a[r] - mx) : 0.f; sum += a[r]; }
  sum += xor32b(sum);
  const float inv = 1.f / sum;
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] *= inv;
}
}  // namespace

// ---- packing --------------------------------------------------------------------------------------------------------
// src fp32 [nb][tk][C] -> pk [nb][hi 32 x C | lo 32 x C | T C x 32] bf16; optionally D[b][t] = sum_c src * (other - base)
struct PackArgs { const float* src; unsigned short* pk; int tk, C; const float* other; const float* base; float* D; };
__global__ __launch_bounds__(256) void tok_pack_k(const PackArgs p) {
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long C = p.C;
  unsigned short* hi = p.pk + (long)b * 96 * C;
  unsigned short* lo = hi + 32 * C;
  unsigned short* T = lo + 32 * C;
  const float* s = p.src + (long)b * p.tk * C;
  // blockIdx.x: 8 latent-token rows per workgroup, 2 per wave
  for (int i = 0; i < 2; ++i) {
    const int t = blockIdx.x * 8 + wave * 2 + i;
    if (t >= 32) break;
    float d = 0.f;
    const int pp = perm_pos(t);
    for (int c = lane; c < p.C; c += 64) {
      const float x = t < p.tk ? s[(long)t * C + c] : 0.f;
      const unsigned short h = f2bf(x);
      hi[(long)t * C + c] = h;
      lo[(long)t * C + c] = f2bf(x - bf2f(h));
      T[(long)c * 32 + pp] = h;
      if (p.D && t < p.tk) d += x * (p.other[((long)b * p.tk + t) * C + c] - p.base[(long)t * C + c]);
    }
    if (p.D && t < p.tk) {
      d = group_sum(d, 64);
      if (lane == 0) p.D[(long)b * p.tk + t] = d;
    }
  }
}
void tok_pack(const Ctx& ctx, const float* src, int nb, int tk, int C, void* pk, const float* other, const float* base, float* D) {
  PackArgs a{src, (unsigned short*)pk, tk, C, other, base, D};
  hipLaunchKernelGGL(tok_pack_k, dim3(4, nb), dim3(256), 0, (hipStream_t)ctx.stream, a);
}
long tok_pack_elems(int nb, int C) { return (long)nb * 96 * C; }

// ---- xattn_fwd ------------------------------------------------------------------------------------------------------
struct XF2Args { const unsigned short* X; const unsigned short* pk; const float* gate_av; int N, C, tk, nrb, total; unsigned short* X1; };
__global__ __launch_bounds__(256) void xattn_fwd2_k(const XF2Args p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * IMG2];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * 4 + wave;
  if (gw >= p.total) return;
  const int b = gw / p.nrb, rb = gw - b * p.nrb, n0 = rb * 32;
  const int rows = p.N - n0 < 32 ? p.N - n0 : 32;
  const long C = p.C;
  char* img = smem + wave * IMG2;
  const unsigned short* Xg = p.X + ((long)b * p.N + n0) * C;
  unsigned short* Og = p.X1 + ((long)b * p.N + n0) * C;
  const unsigned short* th = p.pk + (long)b * 96 * C;
  const unsigned short* tl = th + 32 * C;
  const unsigned short* tT = tl + 32 * C;
  const float g = *p.gate_av;
  const int nsl = (p.C + CS2 - 1) / CS2;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  Slab s;
  slab_load(s, Xg, C, rows, 0, p.C, lane);
  for (int si = 0; si < nsl; ++si) {
    const int c0 = si * CS2, kc = (p.C - c0 < CS2 ? p.C - c0 : CS2) / 16;
    wave_sync();
    slab_store_lds(s, img, rows, c0, p.C, lane);
    wave_sync();
    if (si + 1 < nsl) slab_load(s, Xg, C, rows, c0 + CS2, p.C, lane);
    else if (nsl > 1) slab_load(s, Xg, C, rows, 0, p.C, lane);           // first slab of the second pass
    logits_slab<true>(acc, th, tl, C, c0, kc, img, lane);
  }
  softmax_regs(acc, p.tk, lane);
  float pr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) pr[r] = acc[r];
  const bfx8 pb0 = regs_frag(pr), pb1 = regs_frag(pr + 8);
  // X1^T[c][n] = X^T + g * sum_t tok[t][c] P[t][n]: per 32-channel tile two MFMAs; the lane gets 4 x 4 consecutive channels of
  // ITS token row -> 8-byte read-modify-write of the row in the private image, then whole rows leave as 16-byte stores
  for (int si = 0; si < nsl; ++si) {
    const int c0 = si * CS2, nt = (p.C - c0 < CS2 ? p.C - c0 : CS2) / 32;
    if (nsl > 1) {
      wave_sync();
      slab_store_lds(s, img, rows, c0, p.C, lane);
      wave_sync();
      if (si + 1 < nsl) slab_load(s, Xg, C, rows, c0 + CS2, p.C, lane);
    }
    for (int j = 0; j < nt; ++j) {
      const unsigned short* ap = tT + ((long)(c0 + 32 * j + (lane & 31))) * 32 + (lane >> 5) * 8;
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ldg8(ap), pb0, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ldg8(ap + 16), pb1, o, 0, 0, 0);
      char* xr = img + (lane & 31) * PX2 + (32 * j + 4 * (lane >> 5)) * 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint2 w = *reinterpret_cast<uint2*>(xr + q * 16);
        const float x0 = __uint_as_float(w.x << 16) + g * o[4 * q], x1 = __uint_as_float(w.x & 0xffff0000u) + g * o[4 * q + 1];
        const float x2 = __uint_as_float(w.y << 16) + g * o[4 * q + 2], x3 = __uint_as_float(w.y & 0xffff0000u) + g * o[4 * q + 3];
        *reinterpret_cast<uint2*>(xr + q * 16) = make_uint2(pack2(x0, x1), pack2(x2, x3));
      }
    }
    wave_sync();
    slab_copy_out(img, Og, nullptr, C, rows, c0, p.C, lane);
  }
}
bool attn2_ok(const Ctx& ctx, int C) { return ctx.mode == DT_BF16 && C % 32 == 0 && !getenv("DGSCT_ATTN_V1"); }
void xattn_fwd2(const Ctx& ctx, const void* X, const void* tokpk, const float* gate_av, int B, int N, int C, int tk, void* X1) {
  const int nrb = (N + 31) / 32;
  XF2Args a{(const unsigned short*)X, (const unsigned short*)tokpk, gate_av, N, C, tk, nrb, B * nrb, (unsigned short*)X1};
  hipLaunchKernelGGL(xattn_fwd2_k, dim3((a.total + 3) / 4), dim3(256), 0, (hipStream_t)ctx.stream, a);
}

}  // namespace dgsct
