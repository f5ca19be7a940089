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
// Packed latent tokens ("tokpk", per frame or per parameter): three bf16 images of 32 x C elements each, rows >= tk zero,
// stored in MFMA-FRAGMENT ORDER so that one operand load of a wave is 64 lanes x 16 B = 1 KiB contiguous (a row-major
// image costs 32 cache lines per load -- one per latent token -- and the L1 transaction rate, not HBM, bounds the kernel):
//   hiF / loF [C/16][2][32][8] : element ((kk*2 + h)*32 + t)*8 + e            = hi / lo of tok[t][16 kk + 8 h + e]
//   TF [C/32][2][2][32][8]     : element (((j*2 + kk)*2 + h)*32 + c)*8 + e    = hi of tok[16 kk + 4 h + (e&3) + 8 (e>>2)][32 j + c]
// (TF's latent-token order is the order in which a lane holds the probabilities of its token row after the first product).
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
constexpr int CS2 = 128;                       // channels per slab
constexpr int PX2 = CS2 * 2 + 16;              // pitch of a private [32][CS2] bf16 image (272 B: conflict-free b128 rows)
constexpr int IMG2 = 32 * PX2;                 // 8704 B per image

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
__device__ __forceinline__ unsigned pack2(float a, float b) { return f2bf2(a, b); }
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
  // the optional addend is fetched up front, unconditionally, from clamped addresses: loaded inside the per-chunk
  // `if (valid) { if (add) ... }` it was one exposed round trip per chunk (8 per slab)
  uint4 av[8];
  if (add) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = i * 64 + lane, r = idx >> 4, c = (idx & 15) * 8;
      const int rr = r < rows_valid ? r : rows_valid - 1;
      const int cc = c0 + c < C ? c0 + c : C - 8;
      av[i] = *reinterpret_cast<const uint4*>(add + (long)rr * ld + cc);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = i * 64 + lane, r = idx >> 4, c = (idx & 15) * 8;
    if (r >= rows_valid || c0 + c >= C) continue;
    uint4 v = *reinterpret_cast<const uint4*>(img + r * PX2 + c * 2);
    const long o = (long)r * ld + c0 + c;
    if (add) {
      float x[8], y[8];
      unpack<DT_BF16, 8>(v, x);
      unpack<DT_BF16, 8>(av[i], y);
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
__device__ __forceinline__ void logits_slab(f32x16& acc, const unsigned short* hiF, const unsigned short* loF, int c0, int kc,
                                            const char* img, int lane) {
  const unsigned short* ah = hiF + ((long)(c0 >> 4) * 64 + lane) * 8;      // fragment (kk, lane): 16 B, a wave reads 1 KiB contiguous
  const unsigned short* al = loF + ((long)(c0 >> 4) * 64 + lane) * 8;
  const char* bp = img + (lane & 31) * PX2 + (lane >> 5) * 16;
  // k-steps in groups of 4: the 8 operand loads of a group are in flight together (one exposed L2 round trip per group
  // instead of one per load); slabs are whole multiples of 32 channels, so a group is 2 or 4 steps
#pragma unroll
  for (int g = 0; g < CS2 / 64; ++g) {
    if (g * 4 >= kc) break;
    bfx8 fh[4], fl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kk = g * 4 + i < kc ? g * 4 + i : 0;
      fh[i] = ldg8(ah + kk * 512);
      if (LO) fl[i] = ldg8(al + kk * 512);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (g * 4 + i < kc) {
        const bfx8 bf = lds8(bp + (g * 4 + i) * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[i], bf, acc, 0, 0, 0);
        if (LO) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[i], bf, acc, 0, 0, 0);
      }
    }
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
  const int b = blockIdx.y, tid = threadIdx.x;
  const long C = p.C;
  unsigned short* hiF = p.pk + (long)b * 96 * C;
  unsigned short* loF = hiF + 32 * C;
  unsigned short* TF = loF + 32 * C;
  const float* s = p.src + (long)b * p.tk * C;
  // one thread per 8-element fragment piece (16 B out), 4 C of them per image
  for (long f = (long)blockIdx.x * 256 + tid; f < 4 * C; f += (long)gridDim.x * 256) {
    // loads are unconditional from clamped rows and masked afterwards (`t < tk ? s[..] : 0` per element compiles to a
    // branch + load + s_waitcnt vmcnt(0) each: 16 serial round trips per piece)
    {   // hiF / loF piece f = (kk*2 + h)*32 + t
      const int t = (int)(f & 31), h = (int)((f >> 5) & 1), kk = (int)(f >> 6);
      const float* row = s + (long)(t < p.tk ? t : 0) * C + 16 * kk + 8 * h;
      const float4 q0 = *reinterpret_cast<const float4*>(row), q1 = *reinterpret_cast<const float4*>(row + 4);
      const float m = t < p.tk ? 1.f : 0.f;
      const float x[8] = {q0.x * m, q0.y * m, q0.z * m, q0.w * m, q1.x * m, q1.y * m, q1.z * m, q1.w * m};
      unsigned hw[4], lw[4];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const unsigned short h0 = f2bf(x[e]), h1 = f2bf(x[e + 1]);
        hw[e >> 1] = (unsigned)h0 | ((unsigned)h1 << 16);
        lw[e >> 1] = f2bf2(x[e] - bf2f(h0), x[e + 1] - bf2f(h1));
      }
      *reinterpret_cast<uint4*>(hiF + f * 8) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
      *reinterpret_cast<uint4*>(loF + f * 8) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
    {   // TF piece f = ((j*2 + kk)*2 + h)*32 + c
      const int c = (int)(f & 31), h = (int)((f >> 5) & 1), kk = (int)((f >> 6) & 1), j = (int)(f >> 7);
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int t = 16 * kk + 4 * h + (e & 3) + 8 * (e >> 2);
        x[e] = s[(long)(t < p.tk ? t : 0) * C + 32 * j + c];
      }
      unsigned w[4];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const int t0 = 16 * kk + 4 * h + (e & 3) + 8 * (e >> 2);
        w[e >> 1] = f2bf2(t0 < p.tk ? x[e] : 0.f, t0 + 1 < p.tk ? x[e + 1] : 0.f);
      }
      *reinterpret_cast<uint4*>(TF + f * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
  if (p.D) {          // D[b][t] = sum_c src * (other - base): the waves of all the frame's workgroups share the latent tokens
    const int lane = tid & 63, gw = blockIdx.x * 4 + (tid >> 6), nw = gridDim.x * 4;
    for (int t = gw; t < p.tk; t += nw) {
      const float* o = p.other + ((long)b * p.tk + t) * C;
      const float* bs = p.base + (long)t * C;
      float d0 = 0.f, d1 = 0.f;
      int c = lane;
      for (; c + 64 < p.C; c += 128) { d0 += s[(long)t * C + c] * (o[c] - bs[c]); d1 += s[(long)t * C + c + 64] * (o[c + 64] - bs[c + 64]); }
      if (c < p.C) d0 += s[(long)t * C + c] * (o[c] - bs[c]);
      const float d = group_sum(d0 + d1, 64);
      if (lane == 0) p.D[(long)b * p.tk + t] = d;
    }
  }
}
void tok_pack(const Ctx& ctx, const float* src, int nb, int tk, int C, void* pk, const float* other, const float* base, float* D) {
  PackArgs a{src, (unsigned short*)pk, tk, C, other, base, D};
  const int gx = (4 * C + 255) / 256;
  hipLaunchKernelGGL(tok_pack_k, dim3(gx < 4 ? gx : 4, nb), dim3(256), 0, (hipStream_t)ctx.stream, a);
}
long tok_pack_elems(int nb, int C) { return (long)nb * 96 * C; }

// ---- xattn_fwd ------------------------------------------------------------------------------------------------------
struct XF2Args { const unsigned short* X; const unsigned short* pk; const float* gate_av; int N, C, tk, nrb, total; unsigned short* X1; };
static bool xattn_fwd3_launch(const Ctx& ctx, const XF2Args& a, int B);      // C-split variant for short frames (below)
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
    logits_slab<true>(acc, th, tl, c0, kc, img, lane);
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
    const unsigned short* ap = tT + ((long)(c0 >> 5) * 128 + lane) * 8;      // TF fragment (tile j, kk, lane) at + (j*128 + kk*64)*8
    bfx8 ta[CS2 / 32][2];
#pragma unroll
    for (int j = 0; j < CS2 / 32; ++j) {
      const int jj = j < nt ? j : 0;
      ta[j][0] = ldg8(ap + jj * 1024);
      ta[j][1] = ldg8(ap + jj * 1024 + 512);
    }
#pragma unroll
    for (int j = 0; j < CS2 / 32; ++j) {
      if (j >= nt) break;
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[j][0], pb0, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[j][1], pb1, o, 0, 0, 0);
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
bool attn2_ok(const Ctx& ctx, int C) {
  static const bool v1 = getenv("DGSCT_ATTN_V1") != nullptr;
  return ctx.mode == DT_BF16 && C % 32 == 0 && !v1;
}
void xattn_fwd2(const Ctx& ctx, const void* X, const void* tokpk, const float* gate_av, int B, int N, int C, int tk, void* X1) {
  const int nrb = (N + 31) / 32;
  XF2Args a{(const unsigned short*)X, (const unsigned short*)tokpk, gate_av, N, C, tk, nrb, B * nrb, (unsigned short*)X1};
  if (xattn_fwd3_launch(ctx, a, B)) return;
  hipLaunchKernelGGL(xattn_fwd2_k, dim3((a.total + 3) / 4), dim3(256), 0, (hipStream_t)ctx.stream, a);
}


// ====================================================================================================================
// Backward kernels.  A workgroup = 4 wavefronts = 4 consecutive 32-row blocks of ONE frame: the row-local part (logits,
// softmax backward, the token-major output rows) is wave-private as above; the [32 x C] latent-token gradients, which
// sum over all rows of the frame, are reduced over the 4 waves in LDS (ds_add_f32) and leave as one fp32 atomic per
// element per workgroup and slab.
// ====================================================================================================================
namespace {
constexpr int PP2 = 64;                        // pitch of a private [32 rows][32 latent tokens] bf16 image (transpose-read)
// the lane's 16 values of a [t][n] register tile -> image[n][t] (bf16): 4 x 8-byte stores
__device__ __forceinline__ void regs_to_img(const float* v, char* img, int lane) {
  char* row = img + (lane & 31) * PP2 + (lane >> 5) * 8;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<uint2*>(row + q * 16) = make_uint2(pack2(v[4 * q], v[4 * q + 1]), pack2(v[4 * q + 2], v[4 * q + 3]));
}
// o[t][c] += sum over the 128 rows of the workgroup of A[n][t] * S[n][c] for ONE 32-channel tile (column tile `j` of the slab):
// the wave walks the P / dS images and the slab images of all 4 waves (visible after a __syncthreads), so every wave
// produces a complete tile of the workgroup's contribution -- no cross-wave reduction
__device__ __forceinline__ void tokgrad_tile(f32x16& o, const char* aimg0, int astride, const char* simg0, int j, int lane) {
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const char* aimg = aimg0 + w * astride;
    const char* simg = simg0 + w * IMG2;
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mt_frag_mn(aimg, PP2, 0, 0, lane), mt_frag_mn(simg, PX2, 32 * j, 0, lane), o, 0, 0, 0);
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mt_frag_mn(aimg, PP2, 0, 1, lane), mt_frag_mn(simg, PX2, 32 * j, 1, lane), o, 0, 0, 0);
  }
}
__device__ __forceinline__ void tile_atomic_out(const f32x16& o, float* dst, long C, int c, int tk, int lane) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int t = mt_row(r, lane);
    if (t < tk) unsafeAtomicAdd(dst + (long)t * C + c, o[r]);
  }
}
// o[c][n] tile = sum_t TF[c][t] * frag(t, n): two MFMAs with the packed transposed latent tokens as A (fragment order)
struct TFrag { bfx8 k0, k1; };
__device__ __forceinline__ TFrag tokT_load(const unsigned short* TFslab, int j, int lane) {
  const unsigned short* ap = TFslab + ((long)j * 128 + lane) * 8;
  TFrag f; f.k0 = ldg8(ap); f.k1 = ldg8(ap + 512);
  return f;
}
__device__ __forceinline__ void tokT_mma(f32x16& o, const TFrag& f, const bfx8& b0, const bfx8& b1) {
  o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.k0, b0, o, 0, 0, 0);
  o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.k1, b1, o, 0, 0, 0);
}
}  // namespace

// ---- xattn_bwd ------------------------------------------------------------------------------------------------------
struct XB2Args {
  const unsigned short* X; const unsigned short* G; const unsigned short* pk; const float* gate_av; int N, C, tk, wpf;
  unsigned short* dX; const unsigned short* R2; float* dtok; float* dgate;
};
static bool xattn_bwd3_launch(const Ctx& ctx, XB2Args a, int B);
// KEEPX (round 5; C <= 128, one slab per row block: the stage-0 shapes): the X slab and the dX1 slab are requested TOGETHER up front and
// the X slab stays in registers for the second half -- one exposed global round trip per row block instead of three (X, then dX1, then
// the L2-hot re-read of X).
template <bool KEEPX>
__global__ __launch_bounds__(256) void xattn_bwd2_k(const XB2Args p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * (IMG2 + 2 * 32 * PP2)];
  __shared__ float dgs[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / p.wpf, n0 = ((blockIdx.x - b * p.wpf) * 4 + wave) * 32;
  const bool active = n0 < p.N;
  const int rows = active ? (p.N - n0 < 32 ? p.N - n0 : 32) : 1;
  const int nb = active ? n0 : 0;
  const long C = p.C;
  char* img = smem + wave * IMG2;
  char* imgP = smem + 4 * IMG2 + wave * 2 * 32 * PP2;
  char* imgS = imgP + 32 * PP2;
  const unsigned short* Xg = p.X + ((long)b * p.N + nb) * C;
  const unsigned short* Gg = p.G + ((long)b * p.N + nb) * C;
  const unsigned short* hiF = p.pk + (long)b * 96 * C;
  const unsigned short* loF = hiF + 32 * C;
  const unsigned short* TF = loF + 32 * C;
  const float g = *p.gate_av;
  const int nsl = (p.C + CS2 - 1) / CS2;
  f32x16 aS, aU;
#pragma unroll
  for (int r = 0; r < 16; ++r) { aS[r] = 0.f; aU[r] = 0.f; }
  Slab s, sx;
  if constexpr (KEEPX) {
    const int kc = (p.C < CS2 ? p.C : CS2) / 16;
    slab_load(sx, Xg, C, rows, 0, p.C, lane);
    slab_load(s, Gg, C, rows, 0, p.C, lane);
    wave_sync();
    slab_store_lds(sx, img, rows, 0, p.C, lane);
    wave_sync();
    logits_slab<true>(aS, hiF, loF, 0, kc, img, lane);
    wave_sync();
    slab_store_lds(s, img, rows, 0, p.C, lane);
    wave_sync();
    logits_slab<true>(aU, hiF, loF, 0, kc, img, lane);
  } else {
  slab_load(s, Xg, C, rows, 0, p.C, lane);
  for (int si = 0; si < 2 * nsl; ++si) {                         // X slabs (logits), then dX1 slabs (U = dX1 . tok^T)
    const bool second = si >= nsl;
    const int c0 = (second ? si - nsl : si) * CS2, kc = (p.C - c0 < CS2 ? p.C - c0 : CS2) / 16;
    wave_sync();
    slab_store_lds(s, img, rows, c0, p.C, lane);
    wave_sync();
    if (si + 1 < 2 * nsl) slab_load(s, si + 1 < nsl ? Xg : Gg, C, rows, (si + 1 < nsl ? si + 1 : si + 1 - nsl) * CS2, p.C, lane);
    else if (nsl > 1) slab_load(s, Gg, C, rows, 0, p.C, lane);
    if (!second) logits_slab<true>(aS, hiF, loF, c0, kc, img, lane);
    else logits_slab<true>(aU, hiF, loF, c0, kc, img, lane);
  }
  }
  softmax_regs(aS, p.tk, lane);
  float dot = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) dot += aS[r] * aU[r];
  dot += xor32b(dot);
  const bool nvalid = active && (lane & 31) < rows;
  float pv[16], dv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { pv[r] = nvalid ? aS[r] : 0.f; dv[r] = nvalid ? g * aS[r] * (aU[r] - dot) : 0.f; }
  const bfx8 db0 = regs_frag(dv), db1 = regs_frag(dv + 8);
  regs_to_img(pv, imgP, lane);
  regs_to_img(dv, imgS, lane);
  {
    float part = (nvalid && lane < 32) ? dot : 0.f;
    part = group_sum(part, 64);
    if (lane == 0) dgs[wave] = part;
  }
  // second pass over the slabs: dX1 (dX rows + P^T dX1), then X (dS^T X)
  const char* img0 = smem;
  const char* imgP0 = smem + 4 * IMG2;
  for (int si = 0; si < nsl; ++si) {
    const int c0 = si * CS2, nt = (p.C - c0 < CS2 ? p.C - c0 : CS2) / 32;
    if (nsl > 1) { wave_sync(); slab_store_lds(s, img, rows, c0, p.C, lane); }
    if constexpr (!KEEPX) slab_load(s, Xg, C, rows, c0, p.C, lane);   // X slab for the second half of this iteration (L2-hot re-read)
    __syncthreads();                                              // dX1 slab images + P / dS images of all 4 waves visible
    f32x16 a1, a2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a1[r] = 0.f; a2[r] = 0.f; }
    if (wave < nt) tokgrad_tile(a1, imgP0, 2 * 32 * PP2, img0, wave, lane);              // P^T . dX1, column tile `wave`
    const unsigned short* TFs = TF + (long)(c0 >> 5) * 1024;
    __syncthreads();                                              // operand reads done before the rows are modified
#pragma unroll
    for (int j = 0; j < CS2 / 32; ++j) {                          // dX rows = dX1 + dS . tok
      if (j >= nt) break;
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
      tokT_mma(o, tokT_load(TFs, j, lane), db0, db1);
      char* xr = img + (lane & 31) * PX2 + (32 * j + 4 * (lane >> 5)) * 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint2 w = *reinterpret_cast<uint2*>(xr + q * 16);
        const float x0 = __uint_as_float(w.x << 16) + o[4 * q], x1 = __uint_as_float(w.x & 0xffff0000u) + o[4 * q + 1];
        const float x2 = __uint_as_float(w.y << 16) + o[4 * q + 2], x3 = __uint_as_float(w.y & 0xffff0000u) + o[4 * q + 3];
        *reinterpret_cast<uint2*>(xr + q * 16) = make_uint2(pack2(x0, x1), pack2(x2, x3));
      }
    }
    wave_sync();
    if (active) slab_copy_out(img, p.dX + ((long)b * p.N + n0) * C, p.R2 ? p.R2 + ((long)b * p.N + n0) * C : nullptr, C, rows, c0, p.C, lane);
    wave_sync();
    if constexpr (KEEPX) slab_store_lds(sx, img, active ? rows : 0, c0, p.C, lane);     // X slab (still in registers)
    else slab_store_lds(s, img, active ? rows : 0, c0, p.C, lane);
    if (si + 1 < nsl) slab_load(s, Gg, C, rows, c0 + CS2, p.C, lane);
    __syncthreads();
    if (wave < nt) {
      tokgrad_tile(a2, imgP0 + 32 * PP2, 2 * 32 * PP2, img0, wave, lane);                 // dS^T . X
#pragma unroll
      for (int r = 0; r < 16; ++r) a2[r] += g * a1[r];
      tile_atomic_out(a2, p.dtok + (long)b * p.tk * C, C, c0 + 32 * wave + (lane & 31), p.tk, lane);
    }
    __syncthreads();
  }
  if (p.dgate && tid == 0) unsafeAtomicAdd(p.dgate, dgs[0] + dgs[1] + dgs[2] + dgs[3]);
}
void xattn_bwd2(const Ctx& ctx, const void* X, const void* dX1, const void* tokpk, const float* gate_av, int B, int N, int C, int tk,
                void* dX, const void* R2, float* dtok, float* dgate) {
  const int wpf = ((N + 31) / 32 + 3) / 4;
  XB2Args a{(const unsigned short*)X, (const unsigned short*)dX1, (const unsigned short*)tokpk, gate_av, N, C, tk, wpf,
            (unsigned short*)dX, (const unsigned short*)R2, dtok, dgate};
  if (xattn_bwd3_launch(ctx, a, B)) return;
  static const bool nokeep = getenv("DGSCT_ATTN_NOKEEPX") != nullptr;
  if (C <= CS2 && !nokeep) hipLaunchKernelGGL(xattn_bwd2_k<true>, dim3(B * wpf), dim3(256), 0, (hipStream_t)ctx.stream, a);
  else hipLaunchKernelGGL(xattn_bwd2_k<false>, dim3(B * wpf), dim3(256), 0, (hipStream_t)ctx.stream, a);
}

// ---- tokattn_bwd ----------------------------------------------------------------------------------------------------
// T0pk: packed my_tokens (one image set); dpk: packed dtok (per frame)
struct TB2Args {
  const unsigned short* Yp; const unsigned short* T0pk; const unsigned short* dpk; const float* lse; const float* D; const float* da;
  float invN; int N, C, tk, wpf; unsigned short* dYp; float* dT0b;
};
static bool tokattn_bwd3_launch(const Ctx& ctx, TB2Args a, int B);
__global__ __launch_bounds__(256) void tokattn_bwd2_k(const TB2Args p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * (IMG2 + 32 * PP2)];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / p.wpf, n0 = ((blockIdx.x - b * p.wpf) * 4 + wave) * 32;
  const bool active = n0 < p.N;
  const int rows = active ? (p.N - n0 < 32 ? p.N - n0 : 32) : 1;
  const int nb = active ? n0 : 0;
  const long C = p.C;
  char* img = smem + wave * IMG2;
  char* imgS = smem + 4 * IMG2 + wave * 32 * PP2;
  const unsigned short* Yg = p.Yp + ((long)b * p.N + nb) * C;
  const unsigned short* thF = p.T0pk;
  const unsigned short* tlF = thF + 32 * C;
  const unsigned short* tTF = tlF + 32 * C;
  const unsigned short* dhF = p.dpk + (long)b * 96 * C;
  const unsigned short* dTF = dhF + 64 * C;
  const int nsl = (p.C + CS2 - 1) / CS2;
  f32x16 aS, aD;
#pragma unroll
  for (int r = 0; r < 16; ++r) { aS[r] = 0.f; aD[r] = 0.f; }
  Slab s;
  slab_load(s, Yg, C, rows, 0, p.C, lane);
  for (int si = 0; si < nsl; ++si) {
    const int c0 = si * CS2, kc = (p.C - c0 < CS2 ? p.C - c0 : CS2) / 16;
    wave_sync();
    slab_store_lds(s, img, rows, c0, p.C, lane);
    wave_sync();
    if (si + 1 < nsl) slab_load(s, Yg, C, rows, c0 + CS2, p.C, lane);
    else if (nsl > 1) slab_load(s, Yg, C, rows, 0, p.C, lane);
    logits_slab<true>(aS, thF, tlF, c0, kc, img, lane);
    logits_slab<false>(aD, dhF, dhF, c0, kc, img, lane);
  }
  const bool nvalid = active && (lane & 31) < rows;
  float pv[16], dv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {          // statistics first, unconditionally (a load inside `ok ? .. : 0` is a serialised round trip each)
    const int t = mt_row(r, lane), tc = t < p.tk ? t : 0;
    pv[r] = p.lse[(long)b * p.tk + tc];
    dv[r] = p.D[(long)b * p.tk + tc];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const bool ok = nvalid && mt_row(r, lane) < p.tk;
    const float pr = ok ? __expf(aS[r] - pv[r]) : 0.f;
    dv[r] = ok ? pr * (aD[r] - dv[r]) : 0.f;
    pv[r] = pr;
  }
  const bfx8 pb0 = regs_frag(pv), pb1 = regs_frag(pv + 8), db0 = regs_frag(dv), db1 = regs_frag(dv + 8);
  regs_to_img(dv, imgS, lane);
  const char* img0 = smem;
  const char* imgS0 = smem + 4 * IMG2;
  for (int si = 0; si < nsl; ++si) {
    const int c0 = si * CS2, nt = (p.C - c0 < CS2 ? p.C - c0 : CS2) / 32;
    if (nsl > 1) {
      wave_sync();
      slab_store_lds(s, img, active ? rows : 0, c0, p.C, lane);
      if (si + 1 < nsl) slab_load(s, Yg, C, rows, c0 + CS2, p.C, lane);
    } else if (!active) {
      slab_store_lds(s, img, 0, c0, p.C, lane);
    }
    __syncthreads();
    if (wave < nt) {                                              // dT0b += dS1^T . Yp, column tile `wave`
      f32x16 a2;
#pragma unroll
      for (int r = 0; r < 16; ++r) a2[r] = 0.f;
      tokgrad_tile(a2, imgS0, 32 * PP2, img0, wave, lane);
      tile_atomic_out(a2, p.dT0b + (long)b * p.tk * C, C, c0 + 32 * wave + (lane & 31), p.tk, lane);
    }
    const unsigned short* dTs = dTF + (long)(c0 >> 5) * 1024;
    const unsigned short* tTs = tTF + (long)(c0 >> 5) * 1024;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CS2 / 32; ++j) {                          // dYp rows = P1 . dtok + dS1 . T0 + da / N  (over the Yp image)
      if (j >= nt) break;
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
      const TFrag fd = tokT_load(dTs, j, lane), ft = tokT_load(tTs, j, lane);
      tokT_mma(o, fd, pb0, pb1);
      tokT_mma(o, ft, db0, db1);
      char* xr = img + (lane & 31) * PX2 + (32 * j + 4 * (lane >> 5)) * 2;
      const float* dap = p.da + (long)b * C + c0 + 32 * j + 4 * (lane >> 5);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 dq = *reinterpret_cast<const float4*>(dap + 8 * q);
        *reinterpret_cast<uint2*>(xr + q * 16) = make_uint2(pack2(o[4 * q] + dq.x * p.invN, o[4 * q + 1] + dq.y * p.invN),
                                                            pack2(o[4 * q + 2] + dq.z * p.invN, o[4 * q + 3] + dq.w * p.invN));
      }
    }
    wave_sync();
    if (active) slab_copy_out(img, p.dYp + ((long)b * p.N + n0) * C, nullptr, C, rows, c0, p.C, lane);
    __syncthreads();
  }
}
void tokattn_bwd2(const Ctx& ctx, const void* Yp, const void* T0pk, const void* dtokpk, const float* lse, const float* D,
                  const float* da, float invN, int B, int N, int C, int tk, void* dYp, float* dT0b) {
  const int wpf = ((N + 31) / 32 + 3) / 4;
  TB2Args a{(const unsigned short*)Yp, (const unsigned short*)T0pk, (const unsigned short*)dtokpk, lse, D, da, invN, N, C, tk, wpf,
            (unsigned short*)dYp, dT0b};
  if (tokattn_bwd3_launch(ctx, a, B)) return;
  hipLaunchKernelGGL(tokattn_bwd2_k, dim3(B * wpf), dim3(256), 0, (hipStream_t)ctx.stream, a);
}


// ====================================================================================================================
// C-split kernels for SHORT frames (B * ceil(N / 32) <= 2048 row blocks: the stage-2/3 shapes of the AVE stack, 32 of its
// 48 adapter calls).  With one wavefront per 32-row block those launches put < 2 waves on a SIMD, and each wave walks
// C / 128 slabs twice as one dependent chain (29.9 us for 47 MB at N = 144, C = 512; 47.9 us for 12 MB at N = 36,
// C = 1024).  Here a WORKGROUP owns the row block and its 4 waves split the CHANNELS: wave w takes slabs w, w + 4, ...
// (SPW <= 3 of them, all loaded up front), the partial logit tiles are summed through LDS (two barriers), and from then
// on everything is wave-private again -- every wave holds the full probabilities and finishes its own channels, whose rows
// are still in its LDS images: one pass over the inputs, no second read.
// ====================================================================================================================
namespace {
// The exchange buffer is 4 KB -- [wave][4 registers][lane] fp32, the 16 registers of a tile go through it in four trips -- and the
// images of the probabilities that follow the reduction alias it as ONE copy for the workgroup (every wave holds the same reduced values
// and writes the same bytes).  With the 16 KB buffer / per-wave images of the first version the SPW = 1 kernels needed 50 KB of LDS:
// 3 workgroups per CU = 768 slots for the 800 row blocks of N = 144 (160 frames x 5) -- 1.04 rounds, i.e. two; the SPW = 2 kernels 84 KB:
// ONE workgroup per CU, 1.25 rounds for the 320 row blocks of N = 36.  Now 38 KB (4 per CU) and 72 KB (2 per CU): one round each.
constexpr int RED_REGS = 4;
constexpr int RED_BYTES = 4 * RED_REGS * 64 * 4;
static_assert(RED_BYTES >= 2 * 32 * 64, "the shared [32][32] bf16 images of P and dS alias the exchange buffer");
__device__ __forceinline__ void wg_reduce_acc(f32x16& a, float* red, int wave, int lane) {
#pragma unroll
  for (int h = 0; h < 16 / RED_REGS; ++h) {
#pragma unroll
    for (int r = 0; r < RED_REGS; ++r) red[(wave * RED_REGS + r) * 64 + lane] = a[h * RED_REGS + r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RED_REGS; ++r)
      a[h * RED_REGS + r] = (red[r * 64 + lane] + red[(RED_REGS + r) * 64 + lane]) + (red[(2 * RED_REGS + r) * 64 + lane] + red[(3 * RED_REGS + r) * 64 + lane]);
    __syncthreads();                                  // the buffer may be reused
  }
}
// rows of a private image += (or =) the [c][n] tiles  sum_t TFa[c][t] fa(t, n) (+ sum_t TFb[c][t] fb(t, n))
template <bool TWO, bool RMW>
__device__ __forceinline__ void rows_update(char* img, const unsigned short* TFa, const unsigned short* TFb, int nt, const bfx8& a0,
                                            const bfx8& a1, const bfx8& b0, const bfx8& b1, float scale, const float* addv, float addscale,
                                            int lane) {
  TFrag fa[CS2 / 32], fb[CS2 / 32];
#pragma unroll
  for (int j = 0; j < CS2 / 32; ++j) {
    const int jj = j < nt ? j : 0;
    fa[j] = tokT_load(TFa, jj, lane);
    if (TWO) fb[j] = tokT_load(TFb, jj, lane);
  }
#pragma unroll
  for (int j = 0; j < CS2 / 32; ++j) {
    if (j >= nt) break;
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    tokT_mma(o, fa[j], a0, a1);
    if (TWO) tokT_mma(o, fb[j], b0, b1);
    char* xr = img + (lane & 31) * PX2 + (32 * j + 4 * (lane >> 5)) * 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float x0 = scale * o[4 * q], x1 = scale * o[4 * q + 1], x2 = scale * o[4 * q + 2], x3 = scale * o[4 * q + 3];
      if (RMW) {
        const uint2 w = *reinterpret_cast<uint2*>(xr + q * 16);
        x0 += __uint_as_float(w.x << 16); x1 += __uint_as_float(w.x & 0xffff0000u);
        x2 += __uint_as_float(w.y << 16); x3 += __uint_as_float(w.y & 0xffff0000u);
      }
      if (addv) {
        const float4 dq = *reinterpret_cast<const float4*>(addv + 32 * j + 4 * (lane >> 5) + 8 * q);
        x0 += dq.x * addscale; x1 += dq.y * addscale; x2 += dq.z * addscale; x3 += dq.w * addscale;
      }
      *reinterpret_cast<uint2*>(xr + q * 16) = make_uint2(pack2(x0, x1), pack2(x2, x3));
    }
  }
}
// o[t][c] tile j = sum over the 32 rows of A[n][t] * S[n][c]  (A: [n][t] image, S: the slab image)
__device__ __forceinline__ void tokgrad_tile1(f32x16& o, const char* aimg, const char* simg, int j, int lane) {
  o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mt_frag_mn(aimg, PP2, 0, 0, lane), mt_frag_mn(simg, PX2, 32 * j, 0, lane), o, 0, 0, 0);
  o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mt_frag_mn(aimg, PP2, 0, 1, lane), mt_frag_mn(simg, PX2, 32 * j, 1, lane), o, 0, 0, 0);
}
}  // namespace

template <int SPW>
__global__ __launch_bounds__(256) void xattn_fwd3_k(const XF2Args p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * SPW * IMG2 + RED_BYTES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / p.nrb, rb = blockIdx.x - b * p.nrb, n0 = rb * 32;
  const int rows = p.N - n0 < 32 ? p.N - n0 : 32;
  const long C = p.C;
  char* img = smem + wave * SPW * IMG2;
  float* red = reinterpret_cast<float*>(smem + 4 * SPW * IMG2);
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
  Slab s[SPW];
#pragma unroll
  for (int i = 0; i < SPW; ++i)
    if (wave + 4 * i < nsl) slab_load(s[i], Xg, C, rows, (wave + 4 * i) * CS2, p.C, lane);
#pragma unroll
  for (int i = 0; i < SPW; ++i)
    if (wave + 4 * i < nsl) slab_store_lds(s[i], img + i * IMG2, rows, (wave + 4 * i) * CS2, p.C, lane);
  wave_sync();
#pragma unroll
  for (int i = 0; i < SPW; ++i) {
    const int c0 = (wave + 4 * i) * CS2;
    if (wave + 4 * i < nsl) logits_slab<true>(acc, th, tl, c0, (p.C - c0 < CS2 ? p.C - c0 : CS2) / 16, img + i * IMG2, lane);
  }
  wg_reduce_acc(acc, red, wave, lane);
  softmax_regs(acc, p.tk, lane);
  float pr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) pr[r] = acc[r];
  const bfx8 pb0 = regs_frag(pr), pb1 = regs_frag(pr + 8);
#pragma unroll
  for (int i = 0; i < SPW; ++i) {
    const int c0 = (wave + 4 * i) * CS2;
    if (wave + 4 * i >= nsl) break;
    const int nt = (p.C - c0 < CS2 ? p.C - c0 : CS2) / 32;
    rows_update<false, true>(img + i * IMG2, tT + (long)(c0 >> 5) * 1024, nullptr, nt, pb0, pb1, pb0, pb1, g, nullptr, 0.f, lane);
    wave_sync();
    slab_copy_out(img + i * IMG2, Og, nullptr, C, rows, c0, p.C, lane);
  }
}

template <int SPW>
__global__ __launch_bounds__(256) void xattn_bwd3_k(const XB2Args p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * SPW * IMG2 + RED_BYTES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / p.wpf, rb = blockIdx.x - b * p.wpf, n0 = rb * 32;          // wpf = row blocks per frame here
  const int rows = p.N - n0 < 32 ? p.N - n0 : 32;
  const long C = p.C;
  char* img = smem + wave * SPW * IMG2;
  float* red = reinterpret_cast<float*>(smem + 4 * SPW * IMG2);
  char* imgP = smem + 4 * SPW * IMG2;                            // aliases `red` (free after the reductions); one copy: see RED_BYTES
  char* imgS = imgP + 32 * PP2;
  const unsigned short* Xg = p.X + ((long)b * p.N + n0) * C;
  const unsigned short* Gg = p.G + ((long)b * p.N + n0) * C;
  const unsigned short* hiF = p.pk + (long)b * 96 * C;
  const unsigned short* loF = hiF + 32 * C;
  const unsigned short* TF = loF + 32 * C;
  const float g = *p.gate_av;
  const int nsl = (p.C + CS2 - 1) / CS2;
  f32x16 aS, aU;
#pragma unroll
  for (int r = 0; r < 16; ++r) { aS[r] = 0.f; aU[r] = 0.f; }
  Slab sx[SPW], sg[SPW];
#pragma unroll
  for (int i = 0; i < SPW; ++i)
    if (wave + 4 * i < nsl) {
      slab_load(sx[i], Xg, C, rows, (wave + 4 * i) * CS2, p.C, lane);
      slab_load(sg[i], Gg, C, rows, (wave + 4 * i) * CS2, p.C, lane);
    }
#pragma unroll
  for (int i = 0; i < SPW; ++i)
    if (wave + 4 * i < nsl) slab_store_lds(sx[i], img + i * IMG2, rows, (wave + 4 * i) * CS2, p.C, lane);
  wave_sync();
#pragma unroll
  for (int i = 0; i < SPW; ++i) {
    const int c0 = (wave + 4 * i) * CS2;
    if (wave + 4 * i < nsl) logits_slab<true>(aS, hiF, loF, c0, (p.C - c0 < CS2 ? p.C - c0 : CS2) / 16, img + i * IMG2, lane);
  }
  wave_sync();
#pragma unroll
  for (int i = 0; i < SPW; ++i)
    if (wave + 4 * i < nsl) slab_store_lds(sg[i], img + i * IMG2, rows, (wave + 4 * i) * CS2, p.C, lane);      // dX1 rows stay in the image
  wave_sync();
#pragma unroll
  for (int i = 0; i < SPW; ++i) {
    const int c0 = (wave + 4 * i) * CS2;
    if (wave + 4 * i < nsl) logits_slab<true>(aU, hiF, loF, c0, (p.C - c0 < CS2 ? p.C - c0 : CS2) / 16, img + i * IMG2, lane);
  }
  wg_reduce_acc(aS, red, wave, lane);
  wg_reduce_acc(aU, red, wave, lane);
  softmax_regs(aS, p.tk, lane);
  float dot = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) dot += aS[r] * aU[r];
  dot += xor32b(dot);
  const bool nvalid = (lane & 31) < rows;
  float pv[16], dv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { pv[r] = nvalid ? aS[r] : 0.f; dv[r] = nvalid ? g * aS[r] * (aU[r] - dot) : 0.f; }
  const bfx8 db0 = regs_frag(dv), db1 = regs_frag(dv + 8);
  regs_to_img(pv, imgP, lane);
  regs_to_img(dv, imgS, lane);
  if (p.dgate && wave == 0) {
    float part = (nvalid && lane < 32) ? dot : 0.f;
    part = group_sum(part, 64);
    if (lane == 0) unsafeAtomicAdd(p.dgate, part);
  }
  wave_sync();
#pragma unroll
  for (int i = 0; i < SPW; ++i) {
    const int c0 = (wave + 4 * i) * CS2;
    if (wave + 4 * i >= nsl) break;
    const int nt = (p.C - c0 < CS2 ? p.C - c0 : CS2) / 32;
    char* im = img + i * IMG2;
    f32x16 a1[CS2 / 32];
#pragma unroll
    for (int j = 0; j < CS2 / 32; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[j][r] = 0.f;
      if (j < nt) tokgrad_tile1(a1[j], imgP, im, j, lane);                                  // P^T . dX1
    }
    wave_sync();                                                                            // operand reads before the rows change
    rows_update<false, true>(im, TF + (long)(c0 >> 5) * 1024, nullptr, nt, db0, db1, db0, db1, 1.f, nullptr, 0.f, lane);   // dX = dX1 + dS . tok
    wave_sync();
    slab_copy_out(im, p.dX + ((long)b * p.N + n0) * C, p.R2 ? p.R2 + ((long)b * p.N + n0) * C : nullptr, C, rows, c0, p.C, lane);
    wave_sync();
    slab_store_lds(sx[i], im, rows, c0, p.C, lane);                                         // X rows back (from registers)
    wave_sync();
#pragma unroll
    for (int j = 0; j < CS2 / 32; ++j) {
      if (j >= nt) break;
      f32x16 a2;
#pragma unroll
      for (int r = 0; r < 16; ++r) a2[r] = g * a1[j][r];
      tokgrad_tile1(a2, imgS, im, j, lane);                                                 // + dS^T . X
      tile_atomic_out(a2, p.dtok + (long)b * p.tk * C, C, c0 + 32 * j + (lane & 31), p.tk, lane);
    }
  }
}

template <int SPW>
__global__ __launch_bounds__(256) void tokattn_bwd3_k(const TB2Args p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * SPW * IMG2 + RED_BYTES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / p.wpf, rb = blockIdx.x - b * p.wpf, n0 = rb * 32;          // wpf = row blocks per frame here
  const int rows = p.N - n0 < 32 ? p.N - n0 : 32;
  const long C = p.C;
  char* img = smem + wave * SPW * IMG2;
  float* red = reinterpret_cast<float*>(smem + 4 * SPW * IMG2);
  char* imgS = smem + 4 * SPW * IMG2;                             // aliases `red`; one copy for the workgroup (RED_BYTES)
  const unsigned short* Yg = p.Yp + ((long)b * p.N + n0) * C;
  const unsigned short* thF = p.T0pk;
  const unsigned short* tlF = thF + 32 * C;
  const unsigned short* tTF = tlF + 32 * C;
  const unsigned short* dhF = p.dpk + (long)b * 96 * C;
  const unsigned short* dTF = dhF + 64 * C;
  const int nsl = (p.C + CS2 - 1) / CS2;
  f32x16 aS, aD;
#pragma unroll
  for (int r = 0; r < 16; ++r) { aS[r] = 0.f; aD[r] = 0.f; }
  {
    Slab s[SPW];
#pragma unroll
    for (int i = 0; i < SPW; ++i)
      if (wave + 4 * i < nsl) slab_load(s[i], Yg, C, rows, (wave + 4 * i) * CS2, p.C, lane);
#pragma unroll
    for (int i = 0; i < SPW; ++i)
      if (wave + 4 * i < nsl) slab_store_lds(s[i], img + i * IMG2, rows, (wave + 4 * i) * CS2, p.C, lane);
  }
  wave_sync();
#pragma unroll
  for (int i = 0; i < SPW; ++i) {
    const int c0 = (wave + 4 * i) * CS2, kc = (p.C - c0 < CS2 ? p.C - c0 : CS2) / 16;
    if (wave + 4 * i < nsl) {
      logits_slab<true>(aS, thF, tlF, c0, kc, img + i * IMG2, lane);
      logits_slab<false>(aD, dhF, dhF, c0, kc, img + i * IMG2, lane);
    }
  }
  wg_reduce_acc(aS, red, wave, lane);
  wg_reduce_acc(aD, red, wave, lane);
  const bool nvalid = (lane & 31) < rows;
  float pv[16], dv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {          // statistics first, unconditionally (a load inside `ok ? .. : 0` is a serialised round trip each)
    const int t = mt_row(r, lane), tc = t < p.tk ? t : 0;
    pv[r] = p.lse[(long)b * p.tk + tc];
    dv[r] = p.D[(long)b * p.tk + tc];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const bool ok = nvalid && mt_row(r, lane) < p.tk;
    const float pr = ok ? __expf(aS[r] - pv[r]) : 0.f;
    dv[r] = ok ? pr * (aD[r] - dv[r]) : 0.f;
    pv[r] = pr;
  }
  const bfx8 pb0 = regs_frag(pv), pb1 = regs_frag(pv + 8), db0 = regs_frag(dv), db1 = regs_frag(dv + 8);
  regs_to_img(dv, imgS, lane);
  wave_sync();
#pragma unroll
  for (int i = 0; i < SPW; ++i) {
    const int c0 = (wave + 4 * i) * CS2;
    if (wave + 4 * i >= nsl) break;
    const int nt = (p.C - c0 < CS2 ? p.C - c0 : CS2) / 32;
    char* im = img + i * IMG2;
#pragma unroll
    for (int j = 0; j < CS2 / 32; ++j) {
      if (j >= nt) break;
      f32x16 a2;
#pragma unroll
      for (int r = 0; r < 16; ++r) a2[r] = 0.f;
      tokgrad_tile1(a2, imgS, im, j, lane);                                                 // d my_tokens += dS1^T . Yp
      tile_atomic_out(a2, p.dT0b + (long)b * p.tk * C, C, c0 + 32 * j + (lane & 31), p.tk, lane);
    }
    wave_sync();
    // dYp rows = P1 . dtok + dS1 . T0 + da / N, written over the Yp image
    rows_update<true, false>(im, dTF + (long)(c0 >> 5) * 1024, tTF + (long)(c0 >> 5) * 1024, nt, pb0, pb1, db0, db1, 1.f,
                             p.da + (long)b * C + c0, p.invN, lane);
    wave_sync();
    slab_copy_out(im, p.dYp + ((long)b * p.N + n0) * C, nullptr, C, rows, c0, p.C, lane);
  }
}

namespace {
// one wave per row block leaves the SIMDs under-filled below ~2 waves each: let a workgroup split the channels instead
bool csplit_ok(int B, int N, int C, int max_spw) {
  static const bool off = getenv("DGSCT_ATTN_NOCSPLIT") != nullptr;
  static const int min_c = getenv("DGSCT_ATTN_CSPLIT_MINC") ? atoi(getenv("DGSCT_ATTN_CSPLIT_MINC")) : 512;
  const long blocks = (long)B * ((N + 31) / 32);
  const int spw = ((C + CS2 - 1) / CS2 + 3) / 4;
  // measured (tools/attn_bench.py, B = 160): wins where the per-wave slab chain is long (C >= 768: xattn_fwd 47.9 -> 33.0 us,
  // xattn_bwd 117 -> 79 us at N = 36, C = 1024), even at C = 512, loses at C = 384 (3 slabs: one wave of four idle)
  return !off && blocks <= 1024 && C >= min_c && spw <= max_spw;
}
}  // namespace

#define DGSCT_SPW_LAUNCH(K, SPW, GRID, ARGS)                                                                                  \
  switch (SPW) {                                                                                                             \
    case 1: hipLaunchKernelGGL(K<1>, dim3(GRID), dim3(256), 0, (hipStream_t)ctx.stream, ARGS); break;                        \
    case 2: hipLaunchKernelGGL(K<2>, dim3(GRID), dim3(256), 0, (hipStream_t)ctx.stream, ARGS); break;                        \
    default: hipLaunchKernelGGL(K<3>, dim3(GRID), dim3(256), 0, (hipStream_t)ctx.stream, ARGS); break;                       \
  }
static bool xattn_fwd3_launch(const Ctx& ctx, const XF2Args& a, int B) {
  if (!csplit_ok(B, a.N, a.C, 3)) return false;
  const int spw = ((a.C + CS2 - 1) / CS2 + 3) / 4;
  DGSCT_SPW_LAUNCH(xattn_fwd3_k, spw, a.total, a);
  return true;
}
static bool xattn_bwd3_launch(const Ctx& ctx, XB2Args a, int B) {
  // round 6 (tools/attn_bench.py, DGSCT_ATTN_CSPLIT_MINC sweep): at C = 512 the wave-per-block kernel is the faster BACKWARD (N = 144:
  // 58.3 vs 70.3 us) while the C-split FORWARD still wins there (20.5 vs 24.9 us): the backward splits from C >= 768 only
  static const int min_c_bwd = getenv("DGSCT_XBWD_CSPLIT_MINC") ? atoi(getenv("DGSCT_XBWD_CSPLIT_MINC")) : 768;
  if (a.C < min_c_bwd) return false;
  if (!csplit_ok(B, a.N, a.C, 2)) return false;          // X and dX1 slabs in registers: 64 VGPRs per slab pair
  const int spw = ((a.C + CS2 - 1) / CS2 + 3) / 4;
  a.wpf = (a.N + 31) / 32;
  if (spw == 1) hipLaunchKernelGGL(xattn_bwd3_k<1>, dim3(B * a.wpf), dim3(256), 0, (hipStream_t)ctx.stream, a);
  else hipLaunchKernelGGL(xattn_bwd3_k<2>, dim3(B * a.wpf), dim3(256), 0, (hipStream_t)ctx.stream, a);
  return true;
}
static bool tokattn_bwd3_launch(const Ctx& ctx, TB2Args a, int B) {
  if (!csplit_ok(B, a.N, a.C, 3)) return false;
  const int spw = ((a.C + CS2 - 1) / CS2 + 3) / 4;
  a.wpf = (a.N + 31) / 32;
  DGSCT_SPW_LAUNCH(tokattn_bwd3_k, spw, B * a.wpf, a);
  return true;
}
#undef DGSCT_SPW_LAUNCH
// tokattn_bwd: against the generic kernel (one workgroup per 128 rows) the C-split one pays 2-5x the fp32 atomics of
// d my_tokens (one contribution per 32-row block: 25 of its 89 us at N = 144, C = 512, measured by switching them off) -- it
// wins only where the slab chain is long: 126 -> 105 us at N = 36, C = 1024; 99 -> 94 at N = 64, C = 768
bool tokattn_bwd_csplit(int B, int N, int C) {
  static const int min_c = getenv("DGSCT_TOKBWD_CSPLIT_MINC") ? atoi(getenv("DGSCT_TOKBWD_CSPLIT_MINC")) : 768;
  return C >= min_c && csplit_ok(B, N, C, 3);
}

// ====================================================================================================================
// tokattn_fwd for short frames (N <= 256: the stage-2/3 shapes, 32 of the 48 adapter calls of the AVE stack): ONE workgroup
// per frame produces the final tok (fp32 + packed bf16 images), lse, a and aE -- no partial results, no combine launch, no
// pack launch.  Phase A: the 4 waves split the frame's 32-row tiles and compute logits^T[n][t] = Yp[n] . T0[t] (Yp rows
// from the private image as the A operand, packed my_tokens from global as B), so the softmax axis n runs along the
// accumulator registers; the probabilities go to ONE shared LDS image [n][t].  Phase B: the waves split the 128-channel
// slabs instead and walk ALL row tiles: O[t][c] = sum_n P[n][t] Yp[n][c] with both operands transpose-read.
// ====================================================================================================================
struct TFS2Args {
  const unsigned short* Yp; const float* T0; const unsigned short* T0pk; int N, C, tk; float invN;
  float* tok; unsigned short* tokpk; float* lse; float* a; unsigned short* aE;
};
__global__ __launch_bounds__(256) void tokattn_fwd_small_k(const TFS2Args p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * IMG2 + 256 * PP2];
  __shared__ float red[2][4][32];
  __shared__ float sl[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const long C = p.C;
  char* img = smem + wave * IMG2;
  char* sP = smem + 4 * IMG2;
  const unsigned short* Yb = p.Yp + (long)b * p.N * C;
  const unsigned short* thF = p.T0pk;
  const unsigned short* tlF = thF + 32 * C;
  const int nrt = (p.N + 31) / 32, nsl = (p.C + CS2 - 1) / CS2;
  const int t = lane & 31;
  // ---- phase A
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rt = wave + 4 * i;
    if (rt >= nrt) break;
    const int rows = p.N - rt * 32 < 32 ? p.N - rt * 32 : 32;
    const unsigned short* Yg = Yb + (long)rt * 32 * C;
    Slab s;
    slab_load(s, Yg, C, rows, 0, p.C, lane);
    for (int si = 0; si < nsl; ++si) {
      const int c0 = si * CS2, kc = (p.C - c0 < CS2 ? p.C - c0 : CS2) / 16;
      wave_sync();
      slab_store_lds(s, img, rows, c0, p.C, lane);
      wave_sync();
      if (si + 1 < nsl) slab_load(s, Yg, C, rows, c0 + CS2, p.C, lane);
      const unsigned short* bh = thF + ((long)(c0 >> 4) * 64 + lane) * 8;
      const unsigned short* bl = tlF + ((long)(c0 >> 4) * 64 + lane) * 8;
      const char* ap = img + (lane & 31) * PX2 + (lane >> 5) * 16;
#pragma unroll
      for (int g = 0; g < CS2 / 64; ++g) {
        if (g * 4 >= kc) break;
        bfx8 fh[4], fl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int kk = g * 4 + q < kc ? g * 4 + q : 0; fh[q] = ldg8(bh + kk * 512); fl[q] = ldg8(bl + kk * 512); }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (g * 4 + q < kc) {
            const bfx8 af = lds8(ap + (g * 4 + q) * 32);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, fh[q], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, fl[q], acc[i], 0, 0, 0);
          }
      }
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if ((wave + 4 * i) * 32 + mt_row(r, lane) < p.N) mx = fmaxf(mx, acc[i][r]);
  mx = fmaxf(mx, xor32b(mx));
  if (lane < 32) red[0][wave][t] = mx;
  __syncthreads();
  const float m = fmaxf(fmaxf(red[0][0][t], red[0][1][t]), fmaxf(red[0][2][t], red[0][3][t]));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rt = wave + 4 * i;
    if (rt >= nrt) break;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = rt * 32 + mt_row(r, lane);
      const float pv = n < p.N ? __expf(acc[i][r] - m) : 0.f;
      sum += pv;
      *reinterpret_cast<unsigned short*>(sP + n * PP2 + t * 2) = f2bf(pv);
    }
  }
  sum += xor32b(sum);
  if (lane < 32) red[1][wave][t] = sum;
  __syncthreads();
  if (wave == 0 && lane < 32) {
    const float l = red[1][0][t] + red[1][1][t] + red[1][2][t] + red[1][3][t];
    sl[t] = 1.f / l;
    if (t < p.tk) p.lse[(long)b * p.tk + t] = m + __logf(l);
  }
  __syncthreads();
  // ---- phase B: wave -> slabs wave, wave + 4, ...
  unsigned short* hiF = p.tokpk + (long)b * 96 * C;
  unsigned short* loF = hiF + 32 * C;
  unsigned short* TFo = loF + 32 * C;
  for (int si = wave; si < nsl; si += 4) {
    const int c0 = si * CS2, nt = (p.C - c0 < CS2 ? p.C - c0 : CS2) / 32;
    f32x16 o[CS2 / 32];
#pragma unroll
    for (int j = 0; j < CS2 / 32; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
    float cs0 = 0.f, cs1 = 0.f;                                   // column sums of channels c0 + 2 lane, + 1
    Slab s;
    slab_load(s, Yb, C, p.N < 32 ? p.N : 32, c0, p.C, lane);
    // my_tokens of this slab for the epilogue, requested NOW: they do not depend on O, and 16 L2 round trips per tile in front of each
    // tile's stores were the largest part of the kernel on wide frames (23 of 47 us at N = 36, C = 1024: eight tiles per wave)
    float t0s[CS2 / 32][16];
#pragma unroll
    for (int j = 0; j < CS2 / 32; ++j) {
      const int c = c0 + 32 * j + (lane & 31), cj = c < p.C ? c : 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) { const int tt = mt_row(r, lane); t0s[j][r] = p.T0[(long)(tt < p.tk ? tt : 0) * C + cj]; }
    }
    for (int rt = 0; rt < nrt; ++rt) {
      const int rows = p.N - rt * 32 < 32 ? p.N - rt * 32 : 32;
      wave_sync();
      slab_store_lds(s, img, rows, c0, p.C, lane);
      wave_sync();
      if (rt + 1 < nrt) slab_load(s, Yb + (long)(rt + 1) * 32 * C, C, p.N - (rt + 1) * 32 < 32 ? p.N - (rt + 1) * 32 : 32, c0, p.C, lane);
      const mt_bf16x8 a0 = mt_frag_mn(sP + rt * 32 * PP2, PP2, 0, 0, lane), a1 = mt_frag_mn(sP + rt * 32 * PP2, PP2, 0, 1, lane);
#pragma unroll
      for (int j = 0; j < CS2 / 32; ++j) {
        if (j >= nt) break;
        o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, mt_frag_mn(img, PX2, 32 * j, 0, lane), o[j], 0, 0, 0);
        o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, mt_frag_mn(img, PX2, 32 * j, 1, lane), o[j], 0, 0, 0);
      }
#pragma unroll 8
      for (int r = 0; r < 32; ++r) {
        const unsigned w = *reinterpret_cast<const unsigned*>(img + r * PX2 + lane * 4);
        cs0 += __uint_as_float(w << 16); cs1 += __uint_as_float(w & 0xffff0000u);
      }
    }
    if (c0 + 2 * lane < p.C) {
      const float a0v = cs0 * p.invN, a1v = cs1 * p.invN;
      *reinterpret_cast<float2*>(p.a + (long)b * C + c0 + 2 * lane) = make_float2(a0v, a1v);
      *reinterpret_cast<unsigned*>(p.aE + (long)b * C + c0 + 2 * lane) = pack2(a0v, a1v);
    }
    // epilogue: tok = T0 + O / l; fp32 rows + the three packed bf16 images
#pragma unroll
    for (int j = 0; j < CS2 / 32; ++j) {
      if (j >= nt) break;
      const int c = c0 + 32 * j + (lane & 31);
      float v[16];
      // (T0 was loaded unconditionally from clamped rows above: `tt < tk ? T0[..] : 0` compiles to branch + load + s_waitcnt vmcnt(0)
      //  PER ELEMENT -- the hipcc pitfall of gemm.hip's guarded staging)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tt = mt_row(r, lane);
        v[r] = tt < p.tk ? t0s[j][r] + o[j][r] * sl[tt] : 0.f;
        if (tt < p.tk) p.tok[((long)b * p.tk + tt) * C + c] = v[r];
        const unsigned short h = f2bf(v[r]);
        // hi / lo of tok[tt][c] -> [tt][32 channels] images in the wave's (now free) slab image; whole 16-byte fragment
        // pieces leave below (2-byte scattered global stores, 32 per lane and tile, were ~10 us of this kernel)
        *reinterpret_cast<unsigned short*>(img + tt * 80 + (lane & 31) * 2) = h;
        *reinterpret_cast<unsigned short*>(img + 2560 + tt * 80 + (lane & 31) * 2) = f2bf(v[r] - bf2f(h));
      }
      wave_sync();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = lane + 64 * i, tt = q & 31, h8 = q >> 5;             // piece: token tt, channels 8 h8 .. + 7 of this tile
        const int cc = c0 + 32 * j + 8 * h8;
        const long fo = (((long)(cc >> 4) * 2 + ((cc >> 3) & 1)) * 32 + tt) * 8;
        *reinterpret_cast<uint4*>(hiF + fo) = *reinterpret_cast<const uint4*>(img + tt * 80 + h8 * 16);
        *reinterpret_cast<uint4*>(loF + fo) = *reinterpret_cast<const uint4*>(img + 2560 + tt * 80 + h8 * 16);
      }
      wave_sync();
      // TF pieces (tile j of this slab, kk = 0 / 1, lane): the lane's registers [8 kk, 8 kk + 8) ARE the piece
      unsigned short* tf = TFo + (((long)((c0 >> 5) + j) * 2) * 64 + lane) * 8;
      *reinterpret_cast<uint4*>(tf) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
      *reinterpret_cast<uint4*>(tf + 512) = make_uint4(pack2(v[8], v[9]), pack2(v[10], v[11]), pack2(v[12], v[13]), pack2(v[14], v[15]));
    }
  }
}

// ---- round 6: the same kernel on EIGHT wavefronts (512 threads), same arithmetic and rounding points ------------------------------
// The 4-wave kernel above is a latency chain at one wave per SIMD (160 workgroups on 256 CUs): phase A gives a wave up to two row
// tiles x every 128-channel slab in turn (8 load -> LDS -> MFMA steps), phase B one slab x every row tile.  Here
//   phase A: one row tile per wave (N <= 256 = 8 tiles); when the frame has <= 4 row tiles the spare waves split the SLABS of a tile
//            (partial logit tiles summed through LDS: fp32 summation order differs from the 4-wave kernel, nothing else);
//   phase B: items of 64 channels (C / 64 = 6 ... 16 items over 8 waves) instead of 128: two MFMA column tiles per item.
// N = 144, C = 512 alone: see profiles/r06_attn_bench.txt.
template <int HS>
struct HSlab { uint4 v[HS / 16]; };
template <int HS>
__device__ __forceinline__ void hslab_load(HSlab<HS>& s, const unsigned short* g, long ld, int rows_valid, int c0, int C, int lane) {
  constexpr int CPR = HS / 8;                       // 16-byte chunks per row
#pragma unroll
  for (int i = 0; i < HS / 16; ++i) {
    const int idx = i * 64 + lane, r = idx / CPR, c = (idx % CPR) * 8;
    const int rr = r < rows_valid ? r : rows_valid - 1;
    const int cc = c0 + c < C ? c0 + c : C - 8;
    s.v[i] = *reinterpret_cast<const uint4*>(g + (long)rr * ld + cc);
  }
}
template <int HS>
__device__ __forceinline__ void hslab_store_lds(const HSlab<HS>& s, char* img, int rows_valid, int c0, int C, int lane) {
  constexpr int CPR = HS / 8;
#pragma unroll
  for (int i = 0; i < HS / 16; ++i) {
    const int idx = i * 64 + lane, r = idx / CPR, c = (idx % CPR) * 8;
    uint4 v = s.v[i];
    if (r >= rows_valid || c0 + c >= C) v = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(img + r * PX2 + c * 2) = v;
  }
}
constexpr int TF8_NW = 8;
constexpr int TF8_HS = 64;
__global__ __launch_bounds__(TF8_NW * 64) void tokattn_fwd_small8_k(const TFS2Args p) {
  constexpr int NW = TF8_NW, HS = TF8_HS;
  // ONE LDS object: [NW private slab images | shared probabilities [256 n][32 t] | partial logit tiles of the slab-split phase A]
  __shared__ __attribute__((aligned(16))) char smem[NW * IMG2 + 256 * PP2 + NW * 4096];
  __shared__ float red[2][NW][32];
  __shared__ float sl[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const long C = p.C;
  char* img = smem + wave * IMG2;
  char* sP = smem + NW * IMG2;
  float* part = reinterpret_cast<float*>(smem + NW * IMG2 + 256 * PP2);
  const unsigned short* Yb = p.Yp + (long)b * p.N * C;
  const unsigned short* thF = p.T0pk;
  const unsigned short* tlF = thF + 32 * C;
  const int nrt = (p.N + 31) / 32, nsl = (p.C + CS2 - 1) / CS2;
  const int t = lane & 31;
  // ---- phase A: wave -> (row tile rt, slab group sg of nsg): nsg = 1 for 5 ... 8 row tiles, 2 for 3 ... 4, 4 for 1 ... 2
  const int nsg = nrt > 4 ? 1 : (nrt > 2 ? 2 : 4);
  const int rt = wave / nsg, sg = wave - rt * nsg;
  const bool activeA = rt < nrt;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (activeA) {
    const int rows = p.N - rt * 32 < 32 ? p.N - rt * 32 : 32;
    const unsigned short* Yg = Yb + (long)rt * 32 * C;
    const int spg = (nsl + nsg - 1) / nsg, s0 = sg * spg, s1 = s0 + spg < nsl ? s0 + spg : nsl;      // this wave's slabs [s0, s1)
    if (s0 < s1) {
      Slab s;
      slab_load(s, Yg, C, rows, s0 * CS2, p.C, lane);
      for (int si = s0; si < s1; ++si) {
        const int c0 = si * CS2, kc = (p.C - c0 < CS2 ? p.C - c0 : CS2) / 16;
        wave_sync();
        slab_store_lds(s, img, rows, c0, p.C, lane);
        wave_sync();
        if (si + 1 < s1) slab_load(s, Yg, C, rows, c0 + CS2, p.C, lane);
        const unsigned short* bh = thF + ((long)(c0 >> 4) * 64 + lane) * 8;
        const unsigned short* bl = tlF + ((long)(c0 >> 4) * 64 + lane) * 8;
        const char* ap = img + (lane & 31) * PX2 + (lane >> 5) * 16;
#pragma unroll
        for (int g = 0; g < CS2 / 64; ++g) {
          if (g * 4 >= kc) break;
          bfx8 fh[4], fl[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { const int kk = g * 4 + q < kc ? g * 4 + q : 0; fh[q] = ldg8(bh + kk * 512); fl[q] = ldg8(bl + kk * 512); }
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (g * 4 + q < kc) {
              const bfx8 af = lds8(ap + (g * 4 + q) * 32);
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, fh[q], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, fl[q], acc, 0, 0, 0);
            }
        }
      }
    }
  }
  if (nsg > 1) {                                                  // sum the slab groups' partial tiles (group 0 of each row tile keeps the sum)
    if (activeA && sg > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) part[wave * 1024 + r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (activeA && sg == 0) {
      for (int k = 1; k < nsg; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += part[(wave + k) * 1024 + r * 64 + lane];
    }
  }
  const bool ownerA = activeA && sg == 0;
  float mx = -INFINITY;
  if (ownerA) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (rt * 32 + mt_row(r, lane) < p.N) mx = fmaxf(mx, acc[r]);
  }
  mx = fmaxf(mx, xor32b(mx));
  if (lane < 32) red[0][wave][t] = mx;
  __syncthreads();
  float m = red[0][0][t];
#pragma unroll
  for (int w = 1; w < NW; ++w) m = fmaxf(m, red[0][w][t]);
  float sum = 0.f;
  if (ownerA) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = rt * 32 + mt_row(r, lane);
      const float pv = n < p.N ? __expf(acc[r] - m) : 0.f;
      sum += pv;
      *reinterpret_cast<unsigned short*>(sP + n * PP2 + t * 2) = f2bf(pv);
    }
  }
  sum += xor32b(sum);
  if (lane < 32) red[1][wave][t] = sum;
  __syncthreads();
  if (wave == 0 && lane < 32) {
    float l = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) l += red[1][w][t];
    sl[t] = 1.f / l;
    if (t < p.tk) p.lse[(long)b * p.tk + t] = m + __logf(l);
  }
  __syncthreads();
  // ---- phase B: wave -> 64-channel items wave, wave + 8, ...
  unsigned short* hiF = p.tokpk + (long)b * 96 * C;
  unsigned short* loF = hiF + 32 * C;
  unsigned short* TFo = loF + 32 * C;
  const int nh = (p.C + HS - 1) / HS;
  for (int hi = wave; hi < nh; hi += NW) {
    const int c0 = hi * HS, nt = (p.C - c0 < HS ? p.C - c0 : HS) / 32;
    f32x16 o[HS / 32];
#pragma unroll
    for (int j = 0; j < HS / 32; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
    float cs0 = 0.f, cs1 = 0.f;                                   // column sums of channels c0 + 2 lane, + 1 (lanes 0 .. 31)
    HSlab<HS> s;
    hslab_load<HS>(s, Yb, C, p.N < 32 ? p.N : 32, c0, p.C, lane);
    float t0s[HS / 32][16];
#pragma unroll
    for (int j = 0; j < HS / 32; ++j) {
      const int c = c0 + 32 * j + (lane & 31), cj = c < p.C ? c : 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) { const int tt = mt_row(r, lane); t0s[j][r] = p.T0[(long)(tt < p.tk ? tt : 0) * C + cj]; }
    }
    for (int r2 = 0; r2 < nrt; ++r2) {
      const int rows = p.N - r2 * 32 < 32 ? p.N - r2 * 32 : 32;
      wave_sync();
      hslab_store_lds<HS>(s, img, rows, c0, p.C, lane);
      wave_sync();
      if (r2 + 1 < nrt) hslab_load<HS>(s, Yb + (long)(r2 + 1) * 32 * C, C, p.N - (r2 + 1) * 32 < 32 ? p.N - (r2 + 1) * 32 : 32, c0, p.C, lane);
      const mt_bf16x8 a0 = mt_frag_mn(sP + r2 * 32 * PP2, PP2, 0, 0, lane), a1 = mt_frag_mn(sP + r2 * 32 * PP2, PP2, 0, 1, lane);
#pragma unroll
      for (int j = 0; j < HS / 32; ++j) {
        if (j >= nt) break;
        o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, mt_frag_mn(img, PX2, 32 * j, 0, lane), o[j], 0, 0, 0);
        o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, mt_frag_mn(img, PX2, 32 * j, 1, lane), o[j], 0, 0, 0);
      }
      if (lane < HS / 2) {
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
          const unsigned w = *reinterpret_cast<const unsigned*>(img + r * PX2 + lane * 4);
          cs0 += __uint_as_float(w << 16); cs1 += __uint_as_float(w & 0xffff0000u);
        }
      }
    }
    if (lane < HS / 2 && c0 + 2 * lane < p.C) {
      const float a0v = cs0 * p.invN, a1v = cs1 * p.invN;
      *reinterpret_cast<float2*>(p.a + (long)b * C + c0 + 2 * lane) = make_float2(a0v, a1v);
      *reinterpret_cast<unsigned*>(p.aE + (long)b * C + c0 + 2 * lane) = pack2(a0v, a1v);
    }
    // epilogue: tok = T0 + O / l; fp32 rows + the three packed bf16 images (as in the 4-wave kernel)
#pragma unroll
    for (int j = 0; j < HS / 32; ++j) {
      if (j >= nt) break;
      const int c = c0 + 32 * j + (lane & 31);
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tt = mt_row(r, lane);
        v[r] = tt < p.tk ? t0s[j][r] + o[j][r] * sl[tt] : 0.f;
        if (tt < p.tk) p.tok[((long)b * p.tk + tt) * C + c] = v[r];
        const unsigned short h = f2bf(v[r]);
        *reinterpret_cast<unsigned short*>(img + tt * 80 + (lane & 31) * 2) = h;
        *reinterpret_cast<unsigned short*>(img + 2560 + tt * 80 + (lane & 31) * 2) = f2bf(v[r] - bf2f(h));
      }
      wave_sync();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = lane + 64 * i, tt = q & 31, h8 = q >> 5;             // piece: token tt, channels 8 h8 .. + 7 of this tile
        const int cc = c0 + 32 * j + 8 * h8;
        const long fo = (((long)(cc >> 4) * 2 + ((cc >> 3) & 1)) * 32 + tt) * 8;
        *reinterpret_cast<uint4*>(hiF + fo) = *reinterpret_cast<const uint4*>(img + tt * 80 + h8 * 16);
        *reinterpret_cast<uint4*>(loF + fo) = *reinterpret_cast<const uint4*>(img + 2560 + tt * 80 + h8 * 16);
      }
      wave_sync();
      unsigned short* tf = TFo + (((long)((c0 >> 5) + j) * 2) * 64 + lane) * 8;
      *reinterpret_cast<uint4*>(tf) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
      *reinterpret_cast<uint4*>(tf + 512) = make_uint4(pack2(v[8], v[9]), pack2(v[10], v[11]), pack2(v[12], v[13]), pack2(v[14], v[15]));
    }
  }
}
static std::atomic<int> g_tfs8{1};
int tokattn_small8_mode(int set) { const int old = g_tfs8.load(); if (set >= 0) g_tfs8.store(set); return old; }     // dgsct_test_tune "tfs8": 0 = the 4-wave kernel
bool tokattn_fwd_small_ok(const Ctx& ctx, int N, int C) { return attn2_ok(ctx, C) && N <= 256; }
void tokattn_fwd_small(const Ctx& ctx, const void* Yp, const float* T0, const void* T0pk, int B, int N, int C, int tk, float* tok,
                       void* tokpk, float* lse, float* a, void* aE) {
  TFS2Args q{(const unsigned short*)Yp, T0, (const unsigned short*)T0pk, N, C, tk, 1.f / (float)N, tok, (unsigned short*)tokpk, lse, a,
             (unsigned short*)aE};
  if (g_tfs8.load(std::memory_order_relaxed)) hipLaunchKernelGGL(tokattn_fwd_small8_k, dim3(B), dim3(TF8_NW * 64), 0, (hipStream_t)ctx.stream, q);
  else hipLaunchKernelGGL(tokattn_fwd_small_k, dim3(B), dim3(256), 0, (hipStream_t)ctx.stream, q);
}

}  // namespace dgsct
