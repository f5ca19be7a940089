// Fused window attention of the FROZEN backbone blocks either side of the adapter calls:
//   HTS-AT  WindowAttention.forward   /root/reference DG-SCT/AVE/nets/htsat.py:50-132   (called from the block, :135-251)
//   Swin-V2 window attention of the timm block the AVE loop calls at net_trans.py:894   (cosine form: q, k arrive normalised)
//
//   O[i] = sum_j softmax_j( scale_h * q_i . k_j + bm[w][h][i][j] ) v_j        per (frame, window, head); n = ws^2 <= 144 tokens, hd <= 32
//
// What the ATen formulation costs (profiles/r05_blocks_kernel_stats.txt): roll + window partition copies of the map, a [windows, heads, n, n]
// logits tensor (424 MB per stage-0 block) written, biased, masked, soft-maxed and re-read, per-head batched GEMMs with K = 24 ... 32, the
// inverse partition / roll copies -- and the same again backward.  Here:
//   * the window partition and the cyclic shift are ADDRESS ARITHMETIC: q, k, v rows are gathered straight from the [B, H*W, 3C] output of
//     the qkv projection applied to the un-partitioned map (a Linear commutes with the token permutation) and O rows land at their map
//     positions -- no roll, no partition tensors;
//   * relative-position bias and shift mask are ONE frozen fp32 table bm[window type][head][n][n], added to the logits in registers;
//   * logits^T[j][i] = k_j . q_i is formed TRANSPOSED (v_mfma_f32_32x32x16_bf16, A = K rows, B = Q rows), so the softmax axis j runs along a
//     lane's accumulator registers (+ lane ^ 32): no cross-lane softmax, and the probability tile in registers IS the B operand of
//     O^T = V^T P^T (the contraction index of an MFMA may be permuted as long as both operands agree: V^T is transpose-read from LDS in the
//     order a lane holds its probabilities) -- the probabilities never leave the registers;
//   * backward = two recompute passes without any cross-wave reduction: an i-owner pass (same orientation) finishes dQ rows, a j-owner pass
//     (logits[i][j], A = Q rows, B = K rows) finishes dK and dV rows; D_i = dO_i . O_i and the forward's log-sum-exp replace the softmax
//     backward's row reductions (flash-attention form).
// One workgroup per (frame, window, head); its wavefronts split the 32-row tiles.  bf16 operands, fp32 accumulation / softmax.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "device_util.h"
#include "mma_tile.h"
#include "err.h"
#include "prims.h"

namespace dgsct {
namespace {
typedef mt_bf16x8 wbf8;
typedef mt_f32x16 wf16;
constexpr int WP = 80;                          // LDS pitch of a [tokens][32] bf16 image: 64 B of data + 16 (conflict-free ds_read_b128 rows)
constexpr int WMAXN = 160;                      // tokens of a window, padded to whole 32-row tiles (ws <= 12: n <= 144)
// The kernels are instantiated for windows of <= 64 tokens (HTS-AT's 8 x 8 windows, the 6 x 6 one-window maps: 3 x 64 x 80 B = 15 KB of
// LDS forward, 21 KB backward -> 8+ workgroups per CU) and <= 160 (Swin-V2's 12 x 12: 39 / 53 KB)

struct WArgs {
  const unsigned short* qkv;                    // [B][L][3][heads][hd] bf16 (the qkv projection of the un-partitioned map)
  const float* bm;                              // [nwm][heads][n][n] fp32: relative-position bias (+ shift mask of window w % nwm)
  const float* scale;                           // [heads] logit scale
  unsigned short* out;                          // forward: O [B][L][heads][hd] bf16
  float* lse;                                   // [B][nW][heads][n] fp32: log-sum-exp of every row (forward writes, backward reads)
  const unsigned short* o_in;                   // backward: forward's O
  const unsigned short* dout;                   // backward: dO, laid out like O
  unsigned short* dqkv;                         // backward: laid out like qkv (every element written exactly once)
  int H, W, ws, shift, heads, hd, n, nW, nwm, nwx;
  int items, chunk;                             // workgroups with work; grid = 8 XCDs x chunk (= window instances per XCD x heads)
  int cosine;                                   // q, k rows are L2-normalised in LDS (Swin-V2 cosine attention); backward returns d(raw q, k)
};

// map row (token index y * W + x) of local token t of window w: window partition + cyclic shift as address arithmetic
__device__ __forceinline__ int tok_row(const WArgs& p, int wy, int wx, int t) {
  const int ly = t / p.ws, lx = t - ly * p.ws;
  int y = wy * p.ws + ly + p.shift, x = wx * p.ws + lx + p.shift;
  y = y >= p.H ? y - p.H : y;
  x = x >= p.W ? x - p.W : x;
  return y * p.W + x;
}
// gather rows t = 0 .. n-1 of NI tensors (q / k / v of a head: `rowstride` elements between tokens, first column col0[i]; or O / dO) into NI
// consecutive [npad][32] bf16 images (WIMG bytes apart); rows >= n and columns >= hd are zero.  4 chunks of 16 B per row; per trip a thread
// has its NI loads of TWO chunks in flight before the first LDS store (one image at a time, one chunk per trip, every load was a
// serialised round trip: the kernels spent their time here, not in the products).
struct GSrc { const unsigned short* base; long rowstride; int col0; };
template <int NI>
__device__ __forceinline__ void gather_imgs(char* img0, int wimg, const GSrc (&src)[NI], const int* rows, int n, int npad, int hd, int tid, int nthr) {
  const int total = npad * 4;
  for (int idx = tid; idx < total; idx += 2 * nthr) {
    uint4 v[2][NI];
    int off[2];
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int id = idx + u * nthr, idc = id < total ? id : total - 1;
      const int t = idc >> 2, c = (idc & 3) * 8;
      const int tt = t < n ? t : n - 1, cc = c < hd ? c : 0;
      const long ro = (long)rows[tt];
#pragma unroll
      for (int i = 0; i < NI; ++i) v[u][i] = *reinterpret_cast<const uint4*>(src[i].base + ro * src[i].rowstride + src[i].col0 + cc);
      if (t >= n || c >= hd) {
#pragma unroll
        for (int i = 0; i < NI; ++i) v[u][i] = make_uint4(0, 0, 0, 0);
      }
      off[u] = t * WP + c * 2;
      live[u] = id < total;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (live[u]) {
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<uint4*>(img0 + i * wimg + off[u]) = v[u][i];
      }
  }
}
// K-major fragment of rows row0 .. +31, k-step kk (16 deep): lane l holds image[row0 + (l & 31)][16 kk + 8 (l >> 5) .. + 7]
__device__ __forceinline__ wbf8 frag_km(const char* img, int row0, int kk, int lane) {
  return *reinterpret_cast<const wbf8*>(img + (row0 + (lane & 31)) * WP + (kk * 16 + (lane >> 5) * 8) * 2);
}
// Transposed fragment in ACCUMULATOR order: lane l (m = column l & 31 of the image, half h = l >> 5), element e of k-step kk is
// image[row0 + 16 kk + 4 h + (e & 3) + 8 (e >> 2)][m] -- the rows whose values a lane holds in registers [8 kk, 8 kk + 8) of a 32 x 32
// accumulator tile (mt_row): with it, bf16(registers) are the other operand as they are.
__device__ __forceinline__ wbf8 frag_tr_acc(const char* img, int row0, int kk, int lane) {
  const int kb = row0 + kk * 16 + (lane >> 5) * 4 + ((lane & 15) >> 2);
  const int c = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
  const char* p0 = img + kb * WP + c * 2;
  mt_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) mt_s16x4*)(p0));
  mt_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) mt_s16x4*)(p0 + 8 * WP));
  mt_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(wbf8, v);
}
__device__ __forceinline__ wbf8 regs8(const float* v) {
  uint4 u = make_uint4(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7]));
  return __builtin_bit_cast(wbf8, u);
}
__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }
__device__ __forceinline__ void zero16(wf16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
// rows (row0 + l & 31) of a transposed [d][token] accumulator tile -> bf16 rows at `dst` (token-major, hd columns): 8-byte stores
__device__ __forceinline__ void store_rows_T(const wf16& o, float mul, unsigned short* base, long rowstride, int col0, const int* rows, int row0,
                                             int n, int hd, int lane) {
  const int t = row0 + (lane & 31);
  if (t >= n) return;
  unsigned short* d = base + (long)rows[t] * rowstride + col0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 8 * q + 4 * (lane >> 5);
    if (c < hd)
      *reinterpret_cast<uint2*>(d + c) = make_uint2(f2bf2(o[4 * q] * mul, o[4 * q + 1] * mul), f2bf2(o[4 * q + 2] * mul, o[4 * q + 3] * mul));
  }
}

// cosine attention (Swin-V2: F.normalize(q), F.normalize(k) in front of the logits): rows of the Q and K images (consecutive, WIMG apart)
// scaled to unit L2 norm in place, x / max(|x|, 1e-12) in fp32, stored bf16; inv (optional, [2][npad]) keeps 1 / max(|x|, eps) for backward
__device__ __forceinline__ void normalise_qk(char* img0, int wimg, int npad, float* inv, int tid, int nthr) {
  for (int idx = tid; idx < 2 * npad; idx += nthr) {
    const int which = idx >= npad, t = idx - which * npad;
    char* row = img0 + which * wimg + t * WP;
    float x[4][8], ss = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unpack<DT_BF16, 8>(*reinterpret_cast<const uint4*>(row + q * 16), x[q]);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += x[q][e] * x[q][e];
    }
    const float r = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<uint4*>(row + q * 16) = make_uint4(f2bf2(x[q][0] * r, x[q][1] * r), f2bf2(x[q][2] * r, x[q][3] * r),
                                                           f2bf2(x[q][4] * r, x[q][5] * r), f2bf2(x[q][6] * r, x[q][7] * r));
    if (inv) inv[idx] = r;
  }
}
// store_rows_T through the backward of the normalisation: o holds d(normalised row)^T; the raw row's gradient is
// inv * (g - xn (xn . g)), xn = the normalised row (from its LDS image).  Every lane takes part in the row dot (lane ^ 32 holds the other half).
__device__ __forceinline__ void store_rows_T_cos(const wf16& o, float mul, const char* img, const float* inv, unsigned short* base, long rowstride,
                                                 int col0, const int* rows, int row0, int n, int hd, int lane) {
  const int t = row0 + (lane & 31);
  float xn[16], dot = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint2 u = *reinterpret_cast<const uint2*>(img + t * WP + (8 * q + 4 * (lane >> 5)) * 2);
    xn[4 * q] = __uint_as_float(u.x << 16); xn[4 * q + 1] = __uint_as_float(u.x & 0xffff0000u);
    xn[4 * q + 2] = __uint_as_float(u.y << 16); xn[4 * q + 3] = __uint_as_float(u.y & 0xffff0000u);
#pragma unroll
    for (int e = 0; e < 4; ++e) dot += o[4 * q + e] * xn[4 * q + e];
  }
  dot += xor32(dot);
  if (t >= n) return;
  const float f = mul * inv[t];
  unsigned short* d = base + (long)rows[t] * rowstride + col0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 8 * q + 4 * (lane >> 5);
    if (c < hd)
      *reinterpret_cast<uint2*>(d + c) = make_uint2(f2bf2(f * (o[4 * q] - xn[4 * q] * dot), f * (o[4 * q + 1] - xn[4 * q + 1] * dot)),
                                                    f2bf2(f * (o[4 * q + 2] - xn[4 * q + 2] * dot), f * (o[4 * q + 3] - xn[4 * q + 3] * dot)));
  }
}


// ---- forward --------------------------------------------------------------------------------------------------------
template <int NPAD>
__global__ __launch_bounds__(NPAD <= 64 ? 128 : 320, 4) void wattn_fwd_k(const WArgs p) {
  constexpr int WIMG = NPAD * WP, WMAXT = NPAD / 32;
  __shared__ __attribute__((aligned(16))) char smem[3 * WIMG];
  __shared__ int rows[NPAD];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = blockDim.x >> 6;
  // XCD-aware order: workgroup g runs on XCD g % 8 (each with its own L2).  Window instances k = b * nW + w go round-robin over the XCDs and
  // the heads of one instance stay together on its XCD: (a) two heads share every 128-byte line of the qkv rows (64-byte pieces at head width
  // 32) -- the second finds it in L2; (b) an XCD only ever sees the window types w = k % 8 (mod nW), i.e. an eighth of a shifted block's
  // bias / mask table (5.3 MB at 16 window types x 4 heads of 144 x 144: more than one L2 holds; with every window type on every XCD the
  // backward of the stage-0 Swin block took 940 us instead of 690).  Measured against the plain order: -3 ... -45 % per launch.
  const int kl = (int)(blockIdx.x >> 3) / p.heads, k = kl * 8 + (int)(blockIdx.x & 7);
  const int item = k * p.heads + (int)(blockIdx.x >> 3) - kl * p.heads;
  if (k >= p.items / p.heads) return;
  const int head = item % p.heads, w = (item / p.heads) % p.nW, b = item / (p.heads * p.nW);
  const int wy = w / p.nwx, wx = w - wy * p.nwx;
  const int n = p.n, nt = (n + 31) / 32, npad = nt * 32, L = p.H * p.W, C = p.heads * p.hd;
  for (int t = tid; t < npad; t += blockDim.x) rows[t] = tok_row(p, wy, wx, t < n ? t : n - 1);
  __syncthreads();
  char* sQ = smem; char* sK = smem + WIMG; char* sV = smem + 2 * WIMG;
  const unsigned short* qb = p.qkv + (long)b * L * 3 * C + head * p.hd;
  {
    const GSrc src[3] = {{qb, 3L * C, 0}, {qb, 3L * C, C}, {qb, 3L * C, 2 * C}};
    gather_imgs<3>(smem, WIMG, src, rows, n, npad, p.hd, tid, blockDim.x);
  }
  __syncthreads();
  if (p.cosine) { normalise_qk(smem, WIMG, npad, nullptr, tid, blockDim.x); __syncthreads(); }
  const float sc = p.scale[head];
  const float* bmh = p.bm + ((long)(w % p.nwm) * p.heads + head) * n * n;
  for (int it = wave; it < nt; it += nwv) {
    const int i = it * 32 + (lane & 31), ic = i < n ? i : n - 1;
    const wbf8 q0 = frag_km(sQ, it * 32, 0, lane), q1 = frag_km(sQ, it * 32, 1, lane);
    // online softmax over the j tiles (running maximum, the accumulator rescaled when it moves): one logits tile live at a time.  With all
    // five tiles of a 12 x 12 window held for a two-pass softmax the kernel needed 196 VGPRs -- two wavefronts per SIMD.
    float mx = -INFINITY, sum = 0.f;
    wf16 o; zero16(o);
#pragma unroll 1
    for (int jt = 0; jt < nt; ++jt) {
      wf16 s; zero16(s);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_km(sK, jt * 32, 0, lane), q0, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_km(sK, jt * 32, 1, lane), q1, s, 0, 0, 0);
      float mt = -INFINITY;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j0 = jt * 32 + 8 * q + 4 * (lane >> 5);          // n % 4 == 0: the four columns are all inside or all outside
        const float4 bv = *reinterpret_cast<const float4*>(bmh + (long)ic * n + (j0 < n ? j0 : n - 4));
        const float be[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = j0 < n ? s[4 * q + e] * sc + be[e] : -INFINITY;
          s[4 * q + e] = v;
          mt = fmaxf(mt, v);
        }
      }
      mt = fmaxf(mt, xor32(mt));                                   // (every tile jt < nt has a column inside: finite)
      const float mn = fmaxf(mx, mt), alpha = __expf(mx - mn);     // first tile: exp(-inf) = 0
      float part = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float e = __expf(s[r] - mn); s[r] = e; part += e; o[r] *= alpha; }
      sum = sum * alpha + part;
      mx = mn;
      float* sv = reinterpret_cast<float*>(&s);
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_acc(sV, jt * 32, 0, lane), regs8(sv), o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_acc(sV, jt * 32, 1, lane), regs8(sv + 8), o, 0, 0, 0);
    }
    sum += xor32(sum);
    store_rows_T(o, 1.f / sum, p.out + (long)b * L * C, C, head * p.hd, rows, it * 32, n, p.hd, lane);
    if (lane < 32 && i < n) p.lse[(((long)b * p.nW + w) * p.heads + head) * n + i] = mx + __logf(sum);
  }
}

// ---- backward -------------------------------------------------------------------------------------------------------
template <int NPAD>
__global__ __launch_bounds__(NPAD <= 64 ? 128 : 320, 4) void wattn_bwd_k(const WArgs p) {
  constexpr int WIMG = NPAD * WP, WMAXT = NPAD / 32;
  __shared__ __attribute__((aligned(16))) char smem[4 * WIMG];
  __shared__ int rows[NPAD];
  __shared__ float sL[NPAD], sD[NPAD], sInv[2 * NPAD];       // sInv: 1 / |q_i|, 1 / |k_j| (cosine mode)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = blockDim.x >> 6;
  // XCD-aware order: workgroup g runs on XCD g % 8 (each with its own L2).  Window instances k = b * nW + w go round-robin over the XCDs and
  // the heads of one instance stay together on its XCD: (a) two heads share every 128-byte line of the qkv rows (64-byte pieces at head width
  // 32) -- the second finds it in L2; (b) an XCD only ever sees the window types w = k % 8 (mod nW), i.e. an eighth of a shifted block's
  // bias / mask table (5.3 MB at 16 window types x 4 heads of 144 x 144: more than one L2 holds; with every window type on every XCD the
  // backward of the stage-0 Swin block took 940 us instead of 690).  Measured against the plain order: -3 ... -45 % per launch.
  const int kl = (int)(blockIdx.x >> 3) / p.heads, k = kl * 8 + (int)(blockIdx.x & 7);
  const int item = k * p.heads + (int)(blockIdx.x >> 3) - kl * p.heads;
  if (k >= p.items / p.heads) return;
  const int head = item % p.heads, w = (item / p.heads) % p.nW, b = item / (p.heads * p.nW);
  const int wy = w / p.nwx, wx = w - wy * p.nwx;
  const int n = p.n, nt = (n + 31) / 32, npad = nt * 32, L = p.H * p.W, C = p.heads * p.hd;
  for (int t = tid; t < npad; t += blockDim.x) rows[t] = tok_row(p, wy, wx, t < n ? t : n - 1);
  __syncthreads();
  char* sQ = smem; char* sK = smem + WIMG; char* sV = smem + 2 * WIMG; char* sG = smem + 3 * WIMG;
  const unsigned short* qb = p.qkv + (long)b * L * 3 * C + head * p.hd;
  {
    const GSrc src[4] = {{qb, 3L * C, 0}, {qb, 3L * C, C}, {qb, 3L * C, 2 * C}, {p.dout + (long)b * L * C + head * p.hd, C, 0}};
    gather_imgs<4>(smem, WIMG, src, rows, n, npad, p.hd, tid, blockDim.x);
  }
  for (int t = tid; t < npad; t += blockDim.x) {             // D_i = dO_i . O_i (fp32), lse_i: all eight 16-byte loads of a row in flight
    float d = 0.f, l = 0.f;
    if (t < n) {
      const unsigned short* og = p.o_in + ((long)b * L + rows[t]) * C + head * p.hd;
      const unsigned short* gg = p.dout + ((long)b * L + rows[t]) * C + head * p.hd;
      uint4 xo[4], xg[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 8 * q < p.hd ? 8 * q : 0;
        xo[q] = *reinterpret_cast<const uint4*>(og + c);
        xg[q] = *reinterpret_cast<const uint4*>(gg + c);
      }
      l = p.lse[(((long)b * p.nW + w) * p.heads + head) * n + t];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float x[8], y[8];
        unpack<DT_BF16, 8>(xo[q], x);
        unpack<DT_BF16, 8>(xg[q], y);
        float dd = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) dd += x[e] * y[e];
        if (8 * q < p.hd) d += dd;
      }
    }
    sD[t] = d; sL[t] = l;
  }
  __syncthreads();
  if (p.cosine) { normalise_qk(smem, WIMG, npad, sInv, tid, blockDim.x); __syncthreads(); }
  const float sc = p.scale[head];
  const float* bmh = p.bm + ((long)(w % p.nwm) * p.heads + head) * n * n;
  unsigned short* dq = p.dqkv + (long)b * L * 3 * C + head * p.hd;
  // pass 1 (i-owner; logits^T[j][i]): dQ rows
  for (int it = wave; it < nt; it += nwv) {
    const int i = it * 32 + (lane & 31), ic = i < n ? i : n - 1;
    const wbf8 q0 = frag_km(sQ, it * 32, 0, lane), q1 = frag_km(sQ, it * 32, 1, lane);
    const wbf8 g0 = frag_km(sG, it * 32, 0, lane), g1 = frag_km(sG, it * 32, 1, lane);
    const float li = sL[ic], di = sD[ic];
    wf16 o; zero16(o);
#pragma unroll 1                                             // (unrolled over 5 tiles the compiler kept every tile's fragments and temporaries
    for (int jt = 0; jt < nt; ++jt) {                        //  live: 482 VGPRs, one wavefront per SIMD)
      wf16 s, dp; zero16(s); zero16(dp);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_km(sK, jt * 32, 0, lane), q0, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_km(sK, jt * 32, 1, lane), q1, s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_km(sV, jt * 32, 0, lane), g0, dp, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_km(sV, jt * 32, 1, lane), g1, dp, 0, 0, 0);
      float ds[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j0 = jt * 32 + 8 * q + 4 * (lane >> 5);
        const float4 bv = *reinterpret_cast<const float4*>(bmh + (long)ic * n + (j0 < n ? j0 : n - 4));
        const float be[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q + e;
          const float pr = (j0 < n && i < n) ? __expf(s[r] * sc + be[e] - li) : 0.f;
          ds[r] = pr * (dp[r] - di);
        }
      }
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_acc(sK, jt * 32, 0, lane), regs8(ds), o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_acc(sK, jt * 32, 1, lane), regs8(ds + 8), o, 0, 0, 0);
    }
    if (p.cosine) store_rows_T_cos(o, sc, sQ, sInv, dq, 3L * C, 0, rows, it * 32, n, p.hd, lane);
    else store_rows_T(o, sc, dq, 3L * C, 0, rows, it * 32, n, p.hd, lane);
  }
  // pass 2 (j-owner; logits[i][j]): dK and dV rows
  for (int jt = wave; jt < nt; jt += nwv) {
    const int j = jt * 32 + (lane & 31), jc = j < n ? j : n - 1;
    const wbf8 k0 = frag_km(sK, jt * 32, 0, lane), k1 = frag_km(sK, jt * 32, 1, lane);
    const wbf8 v0 = frag_km(sV, jt * 32, 0, lane), v1 = frag_km(sV, jt * 32, 1, lane);
    wf16 ok, ov; zero16(ok); zero16(ov);
#pragma unroll 1
    for (int it = 0; it < nt; ++it) {
      wf16 s, dp; zero16(s); zero16(dp);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_km(sQ, it * 32, 0, lane), k0, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_km(sQ, it * 32, 1, lane), k1, s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_km(sG, it * 32, 0, lane), v0, dp, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_km(sG, it * 32, 1, lane), v1, dp, 0, 0, 0);
      float pr[16], ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = it * 32 + mt_row(r, lane), ic = i < n ? i : n - 1;
        const float e = (i < n && j < n) ? __expf(s[r] * sc + bmh[(long)ic * n + jc] - sL[ic]) : 0.f;
        pr[r] = e;
        ds[r] = e * (dp[r] - sD[ic]);
      }
      ov = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_acc(sG, it * 32, 0, lane), regs8(pr), ov, 0, 0, 0);
      ov = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_acc(sG, it * 32, 1, lane), regs8(pr + 8), ov, 0, 0, 0);
      ok = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_acc(sQ, it * 32, 0, lane), regs8(ds), ok, 0, 0, 0);
      ok = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_acc(sQ, it * 32, 1, lane), regs8(ds + 8), ok, 0, 0, 0);
    }
    if (p.cosine) store_rows_T_cos(ok, sc, sK, sInv + npad, dq, 3L * C, C, rows, jt * 32, n, p.hd, lane);
    else store_rows_T(ok, sc, dq, 3L * C, C, rows, jt * 32, n, p.hd, lane);
    store_rows_T(ov, 1.f, dq, 3L * C, 2 * C, rows, jt * 32, n, p.hd, lane);
  }
}
}  // namespace

static bool wattn_check(int B, int H, int W, int ws, int shift, int heads, int hd, int nwm) {
  if (B < 1 || ws < 1 || H % ws || W % ws) { set_error("window attention: the map %d x %d is not a whole number of %d x %d windows", H, W, ws, ws); return false; }
  if (ws * ws > 144) { set_error("window attention: windows of up to 12 x 12 tokens (got %d x %d)", ws, ws); return false; }
  if ((ws * ws) % 4) { set_error("window attention: ws * ws must be a multiple of 4"); return false; }
  if (hd % 8 || hd > 32 || hd < 8) { set_error("window attention: head width %d must be 8, 16, 24 or 32", hd); return false; }
  if (shift < 0 || shift >= ws || heads < 1) { set_error("window attention: bad shift / heads"); return false; }
  const int nW = (H / ws) * (W / ws);
  if (nwm != 1 && nwm != nW) { set_error("window attention: the bias/mask table must have 1 or %d window types (got %d)", nW, nwm); return false; }
  return true;
}
static WArgs wattn_args(int B, int H, int W, int ws, int shift, int heads, int hd, int nwm) {
  WArgs a{};
  a.H = H; a.W = W; a.ws = ws; a.shift = shift; a.heads = heads; a.hd = hd; a.n = ws * ws; a.nwx = W / ws; a.nW = (H / ws) * (W / ws); a.nwm = nwm;
  return a;
}
int window_attn_forward(void* stream, int B, int H, int W, int ws, int shift, int heads, int hd, int nwm, const void* qkv, const float* bm,
                        const float* scale, void* out, float* lse, int cosine) {
  if (!wattn_check(B, H, W, ws, shift, heads, hd, nwm)) return 2;
  WArgs a = wattn_args(B, H, W, ws, shift, heads, hd, nwm);
  a.cosine = cosine;
  a.qkv = (const unsigned short*)qkv; a.bm = bm; a.scale = scale; a.out = (unsigned short*)out; a.lse = lse;
  const int nt = (a.n + 31) / 32, nw = nt;        // one wavefront per 32-row tile (n <= 144: <= 5; four waves left one with two of the five tiles)
  a.items = B * a.nW * heads; a.chunk = (B * a.nW + 7) / 8 * heads;
  const dim3 grid((unsigned)(8 * a.chunk));
  if (a.n <= 64) hipLaunchKernelGGL(wattn_fwd_k<64>, grid, dim3(64 * nw), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(wattn_fwd_k<WMAXN>, grid, dim3(64 * nw), 0, (hipStream_t)stream, a);
  return 0;
}
int window_attn_backward(void* stream, int B, int H, int W, int ws, int shift, int heads, int hd, int nwm, const void* qkv, const float* bm,
                         const float* scale, const void* out, const float* lse, const void* dout, void* dqkv, int cosine) {
  if (!wattn_check(B, H, W, ws, shift, heads, hd, nwm)) return 2;
  WArgs a = wattn_args(B, H, W, ws, shift, heads, hd, nwm);
  a.cosine = cosine;
  a.qkv = (const unsigned short*)qkv; a.bm = bm; a.scale = scale; a.o_in = (const unsigned short*)out; a.lse = const_cast<float*>(lse);
  a.dout = (const unsigned short*)dout; a.dqkv = (unsigned short*)dqkv;
  const int nt = (a.n + 31) / 32, nw = nt;        // one wavefront per 32-row tile (n <= 144: <= 5; four waves left one with two of the five tiles)
  a.items = B * a.nW * heads; a.chunk = (B * a.nW + 7) / 8 * heads;
  const dim3 grid((unsigned)(8 * a.chunk));
  if (a.n <= 64) hipLaunchKernelGGL(wattn_bwd_k<64>, grid, dim3(64 * nw), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(wattn_bwd_k<WMAXN>, grid, dim3(64 * nw), 0, (hipStream_t)stream, a);
  return 0;
}

}  // namespace dgsct
