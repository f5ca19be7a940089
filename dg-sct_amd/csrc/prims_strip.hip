// "Column-strip" kernels of the DG-SCT adapter path for gfx950: per-channel affine maps and reductions
// over tokens on [rows][C] token-major tensors (BatchNorm statistics / backward, channel-gate sums,
// bias gradients, ReLU-masked cotangents).
//
// Mapping: a thread owns VE consecutive channels (16 B per lane when C allows it) and walks rows; the
// per-channel parameters it needs live in registers for the whole walk.  Rows are processed UNR at a time
// with all loads issued before the first use, so every lane keeps UNR x 16 B (x number of inputs) in
// flight -- these kernels are pure HBM streams and latency hiding is the whole game.  Reductions are
// combined across the row-slots of the workgroup in LDS (ds_add_f32) and leave as one fp32 atomic per
// channel per workgroup.  The UNR loads of a trip are UNCONDITIONAL (tail rows read a clamped, valid row and are
// masked afterwards): a per-row "load or zero" makes hipcc branch around each load and emit s_waitcnt vmcnt(0)
// after every one of them, i.e. four serial HBM round trips instead of four loads in flight.
#include <atomic>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include "prims.h"
#include "device_util.h"
#include "err.h"

namespace dgsct {

#define STREAM(ctx) ((hipStream_t)(ctx).stream)
static inline long cdiv(long a, long b) { return (a + b - 1) / b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

// rows per trip, by the number of big tensors a kernel streams (more loads in flight per lane until the register
// file pushes occupancy to 1: measured 76.9 / 75.6 / 75.0 / 77.5 ms per step for a uniform 4 / 8 / 16 / 32)
#ifndef DGSCT_UNR1
#define DGSCT_UNR1 16
#endif
#ifndef DGSCT_UNR2
#define DGSCT_UNR2 8
#endif
#ifndef DGSCT_UNR3
#define DGSCT_UNR3 8
#endif
constexpr int UNR1 = DGSCT_UNR1, UNR2 = DGSCT_UNR2, UNR3 = DGSCT_UNR3;

struct ColGeom { int nvr, tpr, rpp, rpc, chunks; };
// target_wgs: ~4096 for pure streams; ~768 (3 per CU) for reductions, whose per-workgroup LDS combine + one global
// atomic per channel must be amortised over many rows (3840 workgroups x 128 channels of atomics on 128 addresses
// cost more than the 94 MB stream itself).
static ColGeom col_geom(int UNR, int C, int VE, long rows, int B, long target_wgs = 4096, bool floor_to_cap = false) {
  ColGeom g;
  g.nvr = C / VE;
  g.tpr = imin(g.nvr, 256);
  g.rpp = 256 / g.tpr;
  long want = floor_to_cap ? target_wgs / B : cdiv(target_wgs, B);
  long maxc = cdiv(rows, (long)g.rpp * UNR);
  long chunks = want < 1 ? 1 : (want > maxc ? maxc : want);
  if (chunks < 1) chunks = 1;
  g.rpc = (int)(cdiv(cdiv(rows, chunks), g.rpp) * g.rpp);
  g.chunks = (int)cdiv(rows, g.rpc);
  return g;
}
static inline int col_ve(const Ctx& ctx, int C) {
  if (ctx.mode == DT_BF16) return C % 8 == 0 ? 8 : (C % 4 == 0 ? 4 : 1);      // ds = 12 (C = 96): 8-byte vectors, not 2-byte scalars
  return C % 4 == 0 ? 4 : 1;
}
// resident capacity (occupancy x CUs) of the instantiation COL_DISPATCH would launch; reductions use one full round
#define COL_CAPACITY(OUT, ctx, VE_, KERNEL, SHMEM)                                                         \
  do {                                                                                                      \
    const void* fn_;                                                                                        \
    if ((ctx).mode == DT_BF16) fn_ = (VE_) == 8 ? reinterpret_cast<const void*>(&KERNEL<DT_BF16, 8>)         \
                                   : ((VE_) == 4 ? reinterpret_cast<const void*>(&KERNEL<DT_BF16, 4>)        \
                                                 : reinterpret_cast<const void*>(&KERNEL<DT_BF16, 1>));      \
    else fn_ = (VE_) == 4 ? reinterpret_cast<const void*>(&KERNEL<DT_F32, 4>)                                \
                          : reinterpret_cast<const void*>(&KERNEL<DT_F32, 1>);                               \
    OUT = wg_capacity(fn_, SHMEM);                                                                          \
    if (const char* e_ = getenv("DGSCT_COL_CAP")) { if (atoi(e_) > 0) OUT = atoi(e_); }                       \
  } while (0)
#define COL_DISPATCH(ctx, VE_, KERNEL, GRID, SHMEM, ...)                                                   \
  do {                                                                                                      \
    if ((ctx).mode == DT_BF16) {                                                                            \
      if ((VE_) == 8) hipLaunchKernelGGL((KERNEL<DT_BF16, 8>), GRID, dim3(256), SHMEM, STREAM(ctx), __VA_ARGS__); \
      else if ((VE_) == 4) hipLaunchKernelGGL((KERNEL<DT_BF16, 4>), GRID, dim3(256), SHMEM, STREAM(ctx), __VA_ARGS__); \
      else hipLaunchKernelGGL((KERNEL<DT_BF16, 1>), GRID, dim3(256), SHMEM, STREAM(ctx), __VA_ARGS__);      \
    } else {                                                                                                \
      if ((VE_) == 4) hipLaunchKernelGGL((KERNEL<DT_F32, 4>), GRID, dim3(256), SHMEM, STREAM(ctx), __VA_ARGS__); \
      else hipLaunchKernelGGL((KERNEL<DT_F32, 1>), GRID, dim3(256), SHMEM, STREAM(ctx), __VA_ARGS__);       \
    }                                                                                                       \
  } while (0)

// combine NQ per-thread channel accumulators across the row-slots of the workgroup; one global atomic per channel.
// Per quantity: every row-slot stores its vector to lds[tr][col..], barrier, 256 threads sum the rpp rows of a column
// (LDS float atomics here cost more than the streaming loop of the late-stage shapes).  lds: rpp * C floats
// (strip_lds).  [c_lo, c_hi): the columns of the current column super-block.
template <int NQ, int VE>
__device__ __forceinline__ void flush_strip(float (&acc)[NQ][VE], float* lds, int C, int col, bool active,
                                            float* const (&dst)[NQ], int tr, int rpp, int c_lo, int c_hi,
                                            float* part = nullptr) {
  float* pw = part ? part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (NQ * C) : nullptr;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    __syncthreads();
    if (active) {
      if constexpr (VE % 4 == 0) {
#pragma unroll
        for (int e = 0; e < VE; e += 4)
          *reinterpret_cast<float4*>(&lds[tr * C + col + e]) = make_float4(acc[q][e], acc[q][e + 1], acc[q][e + 2], acc[q][e + 3]);
      } else {
#pragma unroll
        for (int e = 0; e < VE; ++e) lds[tr * C + col + e] = acc[q][e];
      }
    }
    __syncthreads();
    if (dst[q]) {
      for (int i = c_lo + threadIdx.x; i < c_hi; i += 256) {
        float s = 0.f;
        for (int r = 0; r < rpp; ++r) s += lds[r * C + i];
        if (pw) pw[q * C + i] = s;          // partial sums, finished by part_reduce_k (device_util.h)
        else unsafeAtomicAdd(dst[q] + i, s);
      }
    }
  }
  __syncthreads();
}
// un-batched pure streams too are launched as ONE round of resident workgroups (bn_bwd_apply at 368 640 x 128: 2880
// workgroups on 768 slots, 102 -> 70 us; per-frame grids (scale_cols) gain nothing: the frame count quantises them).  DGSCT_COL_STREAM_CAP=0 restores the fixed ~4096-workgroup target.
static inline bool stream_cap() {
  static const bool on = !(getenv("DGSCT_COL_STREAM_CAP") && atoi(getenv("DGSCT_COL_STREAM_CAP")) == 0);
  return on;
}
static inline size_t strip_lds(int C, int ve) {
  const int nvr = C / ve, tpr = nvr < 256 ? nvr : 256;
  return (size_t)(256 / tpr) * C * sizeof(float);
}

// Strip iteration skeleton used by every kernel below:
//   for vc0 (column super-blocks, > 1 iteration only when C/VE > 256)
//     thread (tc, tr): columns [vc*VE, vc*VE+VE), rows r0+tr, r0+tr+rpp, ... < r_end, UNR rows per trip.
#define STRIP_PROLOGUE(ROWS_END_EXPR, ROW0_EXPR)                                 \
  const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;                      \
  const int nvr = C / VE;                                                        \
  const long r_end = (ROWS_END_EXPR);                                            \
  const long r_begin = (ROW0_EXPR);

// ---- colsum_batched --------------------------------------------------------------------------------
template <int DT, int VE>
__global__ __launch_bounds__(256) void colsum_k(const void* x, long ld, long bs, int N, int C, const float* roww,
                                                long roww_bs, float scale, int tpr, int rpp, int rpc, float* out,
                                                long out_bs) {
  constexpr int UNR = UNR1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.y;
  STRIP_PROLOGUE(lmin_d(N, (long)(blockIdx.x + 1) * rpc), (long)blockIdx.x * rpc)
  for (int vc0 = 0; vc0 < nvr; vc0 += tpr) {
    const int vc = vc0 + tc;
    const bool active = tr < rpp && vc < nvr;
    float acc[1][VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[0][e] = 0.f;
    if (active) {
      for (long n = r_begin + tr; n < r_end; n += (long)rpp * UNR) {
        float t[UNR][VE], rw[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const long nn = n + (long)u * rpp;
          const long nc = nn < r_end ? nn : r_end - 1;            // unconditional, clamped (see the header comment)
          ldv<DT, VE>(x, (long)b * bs + nc * ld + vc * VE, t[u]);
          rw[u] = roww ? roww[(long)b * roww_bs + nc] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) rw[u] = n + (long)u * rpp < r_end ? rw[u] : 0.f;
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
          for (int e = 0; e < VE; ++e) acc[0][e] += rw[u] * t[u][e];
      }
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[0][e] *= scale;
    }
    float* const dst[1] = {out + (long)b * out_bs};
    flush_strip<1, VE>(acc, lds, C, vc * VE, active, dst, tr, rpp, vc0 * VE, imin_d(C, (vc0 + tpr) * VE));
  }
}

void colsum_batched(const Ctx& ctx, const void* x, long ld, long bs, int B, int N, int C, const float* roww, long roww_bs,
                    float scale, float* out, long out_bs) {
  int ve = col_ve(ctx, C);
  if (ld % ve != 0 || bs % ve != 0) ve = 1;
  int cap = 768;
  COL_CAPACITY(cap, ctx, ve, colsum_k, strip_lds(C, ve));
  if (cap > 800) cap = 800;            // light kernel (5 resident/CU): beyond ~3 per CU the extra atomics cost more than they hide
  ColGeom g = col_geom(UNR1, C, ve, N, B, cap, true);
  COL_DISPATCH(ctx, ve, colsum_k, dim3(g.chunks, B), strip_lds(C, ve), x, ld, bs, N, C, roww, roww_bs, scale, g.tpr,
               g.rpp, g.rpc, out, out_bs);
}

// colsum_batched + a second sum over the POSITIVE entries: out_pos[b][c] += sum_n roww * (x > 0)     (prims.h: colsum_batched_pos)
template <int DT, int VE>
__global__ __launch_bounds__(256) void colsum_pos_k(const void* x, long ld, long bs, int N, int C, const float* roww,
                                                    long roww_bs, float scale, int tpr, int rpp, int rpc, float* out,
                                                    long out_bs, float* out_pos, long out_pos_bs) {
  constexpr int UNR = UNR1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.y;
  STRIP_PROLOGUE(lmin_d(N, (long)(blockIdx.x + 1) * rpc), (long)blockIdx.x * rpc)
  for (int vc0 = 0; vc0 < nvr; vc0 += tpr) {
    const int vc = vc0 + tc;
    const bool active = tr < rpp && vc < nvr;
    float acc[2][VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[0][e] = acc[1][e] = 0.f;
    if (active) {
      for (long n = r_begin + tr; n < r_end; n += (long)rpp * UNR) {
        float t[UNR][VE], rw[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const long nn = n + (long)u * rpp;
          const long nc = nn < r_end ? nn : r_end - 1;            // unconditional, clamped
          ldv<DT, VE>(x, (long)b * bs + nc * ld + vc * VE, t[u]);
          rw[u] = roww ? roww[(long)b * roww_bs + nc] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) rw[u] = n + (long)u * rpp < r_end ? rw[u] : 0.f;
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
          for (int e = 0; e < VE; ++e) { acc[0][e] += rw[u] * t[u][e]; acc[1][e] += t[u][e] > 0.f ? rw[u] : 0.f; }
      }
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[0][e] *= scale;
    }
    float* const dst[2] = {out + (long)b * out_bs, out_pos + (long)b * out_pos_bs};
    flush_strip<2, VE>(acc, lds, C, vc * VE, active, dst, tr, rpp, vc0 * VE, imin_d(C, (vc0 + tpr) * VE));
  }
}
void colsum_batched_pos(const Ctx& ctx, const void* x, long ld, long bs, int B, int N, int C, const float* roww, long roww_bs,
                        float scale, float* out, long out_bs, float* out_pos, long out_pos_bs) {
  int ve = col_ve(ctx, C);
  if (ld % ve != 0 || bs % ve != 0) ve = 1;
  int cap = 768;
  COL_CAPACITY(cap, ctx, ve, colsum_pos_k, strip_lds(C, ve));
  if (cap > 800) cap = 800;
  ColGeom g = col_geom(UNR1, C, ve, N, B, cap, true);
  COL_DISPATCH(ctx, ve, colsum_pos_k, dim3(g.chunks, B), strip_lds(C, ve), x, ld, bs, N, C, roww, roww_bs, scale, g.tpr,
               g.rpp, g.rpc, out, out_bs, out_pos, out_pos_bs);
}

// ---- BatchNorm statistics ----------------------------------------------------------------------------
// acc3[0..C) = shift (row 0), acc3[C..2C) += sum(x - shift), acc3[2C..3C) += sum((x - shift)^2)
template <int DT, int VE>
__global__ __launch_bounds__(256) void bn_stats_k(const void* x, long rows, int C, int tpr, int rpp, int rpc, float* acc3) {
  constexpr int UNR = UNR1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  STRIP_PROLOGUE(lmin_d(rows, (long)(blockIdx.x + 1) * rpc), (long)blockIdx.x * rpc)
  for (int vc0 = 0; vc0 < nvr; vc0 += tpr) {
    const int vc = vc0 + tc;
    const bool active = tr < rpp && vc < nvr;
    float acc[2][VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[0][e] = acc[1][e] = 0.f;
    if (active) {
      float sft[VE];
      ldv<DT, VE>(x, (long)vc * VE, sft);
      if (blockIdx.x == 0 && tr == 0) {
#pragma unroll
        for (int e = 0; e < VE; ++e) acc3[vc * VE + e] = sft[e];
      }
      for (long r = r_begin + tr; r < r_end; r += (long)rpp * UNR) {
        float t[UNR][VE];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const long rr = r + (long)u * rpp;
          ldv<DT, VE>(x, (rr < r_end ? rr : r_end - 1) * C + vc * VE, t[u]);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const bool ok = r + (long)u * rpp < r_end;
#pragma unroll
          for (int e = 0; e < VE; ++e) t[u][e] = ok ? t[u][e] : sft[e];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
          for (int e = 0; e < VE; ++e) { const float d = t[u][e] - sft[e]; acc[0][e] += d; acc[1][e] += d * d; }
      }
    }
    float* const dst[2] = {acc3 + C, acc3 + 2 * C};
    flush_strip<2, VE>(acc, lds, C, vc * VE, active, dst, tr, rpp, vc0 * VE, imin_d(C, (vc0 + tpr) * VE));
  }
}

void bn_stats(const Ctx& ctx, const void* x, long rows, int C, float* acc) {
  const int ve = col_ve(ctx, C);
  ColGeom g = col_geom(UNR1, C, ve, rows, 1, 768);
  COL_DISPATCH(ctx, ve, bn_stats_k, dim3(g.chunks), strip_lds(C, ve), x, rows, C, g.tpr, g.rpp, g.rpc, acc);
}

__global__ void bn_finalize_k(const float* acc, long rows, int C, const float* w, const float* b, float* run_mean,
                              float* run_var, float momentum, float eps, int training, float* mean, float* rstd,
                              float* sc, float* sh) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float m, v;
  if (training) {
    const float s1 = acc[C + c] / rows, s2 = acc[2 * C + c] / rows;
    m = acc[c] + s1;
    v = fmaxf(s2 - s1 * s1, 0.f);
    const float unb = rows > 1 ? v * ((float)rows / (float)(rows - 1)) : v;
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * m;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * unb;
  } else {
    m = run_mean[c];
    v = run_var[c];
  }
  const float rs = rsqrtf(v + eps);
  mean[c] = m;
  rstd[c] = rs;
  const float s = w[c] * rs;
  sc[c] = s;
  sh[c] = b[c] - m * s;
}

void bn_finalize(const Ctx& ctx, const float* acc, long rows, int C, const float* w, const float* b, float* run_mean,
                 float* run_var, float momentum, float eps, int training, float* mean, float* rstd, float* sc, float* sh) {
  hipLaunchKernelGGL(bn_finalize_k, dim3((C + 255) / 256), dim3(256), 0, STREAM(ctx), acc, rows, C, w, b, run_mean, run_var,
                     momentum, eps, training, mean, rstd, sc, sh);
}

// ---- per-channel affine (+relu) -----------------------------------------------------------------------
template <int DT, int VE>
__global__ __launch_bounds__(256) void affine_act_k(const void* x, void* y, long rows, int C, int tpr, int rpp, int rpc,
                                                    const float* sc, const float* sh, int relu) {
  constexpr int UNR = UNR1;
  STRIP_PROLOGUE(lmin_d(rows, (long)(blockIdx.x + 1) * rpc), (long)blockIdx.x * rpc)
  for (int vc0 = 0; vc0 < nvr; vc0 += tpr) {
    const int vc = vc0 + tc;
    if (!(tr < rpp && vc < nvr)) continue;
    float a[VE], bsh[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) { a[e] = 1.f; bsh[e] = 0.f; }
    if (sc) { ldv<DT_F32, VE>(sc, vc * VE, a); ldv<DT_F32, VE>(sh, vc * VE, bsh); }      // one branch, vector loads
    for (long r = r_begin + tr; r < r_end; r += (long)rpp * UNR) {
      float t[UNR][VE];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const long rr = r + (long)u * rpp;
        ldv<DT, VE>(x, (rr < r_end ? rr : r_end - 1) * C + vc * VE, t[u]);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const long rr = r + (long)u * rpp;
        if (rr < r_end) {
#pragma unroll
          for (int e = 0; e < VE; ++e) { float v = t[u][e] * a[e] + bsh[e]; t[u][e] = relu ? fmaxf(v, 0.f) : v; }
          stv<DT, VE>(y, rr * C + vc * VE, t[u]);
        }
      }
    }
  }
}

// the same pass with BN's finalisation inside (BnFin, prims.h)
template <int DT, int VE>
__global__ __launch_bounds__(256) void affine_act_bn_k(const void* x, void* y, long rows, int C, int tpr, int rpp, int rpc,
                                                       const BnFin fin, int relu) {
  constexpr int UNR = UNR1;
  STRIP_PROLOGUE(lmin_d(rows, (long)(blockIdx.x + 1) * rpc), (long)blockIdx.x * rpc)
  for (int vc0 = 0; vc0 < nvr; vc0 += tpr) {
    const int vc = vc0 + tc;
    if (!(tr < rpp && vc < nvr)) continue;
    float a[VE], bsh[VE];
    bn_fin_vec<VE>(fin, C, vc * VE, blockIdx.x == 0 && tr == 0, a, bsh);
    for (long r = r_begin + tr; r < r_end; r += (long)rpp * UNR) {
      float t[UNR][VE];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const long rr = r + (long)u * rpp;
        ldv<DT, VE>(x, (rr < r_end ? rr : r_end - 1) * C + vc * VE, t[u]);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const long rr = r + (long)u * rpp;
        if (rr < r_end) {
#pragma unroll
          for (int e = 0; e < VE; ++e) { float v = t[u][e] * a[e] + bsh[e]; t[u][e] = relu ? fmaxf(v, 0.f) : v; }
          stv<DT, VE>(y, rr * C + vc * VE, t[u]);
        }
      }
    }
  }
}

int bnfold_mode(int set) {
  static std::atomic<int> mode{getenv("DGSCT_NO_BNFOLD") ? 0 : 1};
  const int old = mode.load(std::memory_order_relaxed);
  if (set >= 0) mode.store(set ? 1 : 0, std::memory_order_relaxed);
  return old;
}

void affine_act_bn(const Ctx& ctx, const void* x, void* y, long rows, int C, const BnFin& fin, int relu) {
  if (!bnfold_mode(-1) || rows < 1) {
    bn_finalize(ctx, fin.acc, fin.rows, C, fin.w, fin.b, fin.run_mean, fin.run_var, fin.momentum, fin.eps, fin.training, fin.mean,
                fin.rstd, fin.sc, fin.sh);
    affine_act(ctx, x, y, rows, C, fin.sc, fin.sh, relu);
    return;
  }
  const int ve = col_ve(ctx, C);
  ColGeom g = col_geom(UNR1, C, ve, rows, 1);
  if (stream_cap()) { int cap = 4096; COL_CAPACITY(cap, ctx, ve, affine_act_bn_k, 0); g = col_geom(UNR1, C, ve, rows, 1, cap, true); }
  COL_DISPATCH(ctx, ve, affine_act_bn_k, dim3(g.chunks), 0, x, y, rows, C, g.tpr, g.rpp, g.rpc, fin, relu);
}

void affine_act(const Ctx& ctx, const void* x, void* y, long rows, int C, const float* sc, const float* sh, int relu) {
  const int ve = col_ve(ctx, C);
  ColGeom g = col_geom(UNR1, C, ve, rows, 1);
  if (stream_cap()) { int cap = 4096; COL_CAPACITY(cap, ctx, ve, affine_act_k, 0); g = col_geom(UNR1, C, ve, rows, 1, cap, true); }
  COL_DISPATCH(ctx, ve, affine_act_k, dim3(g.chunks), 0, x, y, rows, C, g.tpr, g.rpp, g.rpc, sc, sh, relu);
}

// ---- BatchNorm backward --------------------------------------------------------------------------------
template <int DT, int VE>
__global__ __launch_bounds__(256) void bn_bwd_stats_k(const void* dy, const void* x, long rows, int C, const float* mean,
                                                      const float* rstd, const float* sc, const float* sh, int relu,
                                                      int tpr, int rpp, int rpc, float* sums, float* part) {
  constexpr int UNR = UNR2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  STRIP_PROLOGUE(lmin_d(rows, (long)(blockIdx.x + 1) * rpc), (long)blockIdx.x * rpc)
  for (int vc0 = 0; vc0 < nvr; vc0 += tpr) {
    const int vc = vc0 + tc;
    const bool active = tr < rpp && vc < nvr;
    float acc[2][VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[0][e] = acc[1][e] = 0.f;
    if (active) {
      float m[VE], rs[VE], a[VE], bsh[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) { const int c = vc * VE + e; m[e] = mean[c]; rs[e] = rstd[c]; a[e] = sc[c]; bsh[e] = sh[c]; }
      for (long r = r_begin + tr; r < r_end; r += (long)rpp * UNR) {
        float g[UNR][VE], t[UNR][VE];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const long rr = r + (long)u * rpp;
          const long rc = rr < r_end ? rr : r_end - 1;
          ldv<DT, VE>(dy, rc * C + vc * VE, g[u]);
          ldv<DT, VE>(x, rc * C + vc * VE, t[u]);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const bool ok = r + (long)u * rpp < r_end;
#pragma unroll
          for (int e = 0; e < VE; ++e) { g[u][e] = ok ? g[u][e] : 0.f; t[u][e] = ok ? t[u][e] : 0.f; }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            float gg = g[u][e];
            if (relu && !(t[u][e] * a[e] + bsh[e] > 0.f)) gg = 0.f;
            acc[0][e] += gg;
            acc[1][e] += gg * (t[u][e] - m[e]) * rs[e];
          }
      }
    }
    float* const dst[2] = {sums, sums + C};
    flush_strip<2, VE>(acc, lds, C, vc * VE, active, dst, tr, rpp, vc0 * VE, imin_d(C, (vc0 + tpr) * VE), part);
  }
}

static inline long strip_part_min() {      // tuning hook: below this many atomics the extra finishing launch costs more (A/B: tools/ab_partmin.sh)
  static const long v = getenv("DGSCT_STRIP_PART_MIN") ? atol(getenv("DGSCT_STRIP_PART_MIN")) : 150000;
  return v;
}
static inline bool strip_part() {
  static const bool on = !(getenv("DGSCT_ROW_PART") && atoi(getenv("DGSCT_ROW_PART")) == 0);
  return on;
}
void bn_bwd_stats(const Ctx& ctx, const void* dy, const void* x, long rows, int C, const float* mean, const float* rstd,
                  const float* sc, const float* sh, int relu, float* sums, float* part, long part_floats) {
  const int ve = col_ve(ctx, C);
  ColGeom g = col_geom(UNR2, C, ve, rows, 1, 768);
  // (workgroups x channels of atomics below ~150 k cost less than the finishing launch: measured 9.9 -> 9.0 + 4.7 us)
  if (!strip_part() || (long)g.chunks * 2 * C > part_floats || (long)g.chunks * 2 * C < strip_part_min()) part = nullptr;
  COL_DISPATCH(ctx, ve, bn_bwd_stats_k, dim3(g.chunks), strip_lds(C, ve), dy, x, rows, C, mean, rstd, sc, sh, relu,
               g.tpr, g.rpp, g.rpc, sums, part);
  if (part) {
    PartTable t; t.NQ = 2; t.C = C;
    t.d[0] = PartDesc{0, 1, g.chunks, 1, sums, 0, 1.f};
    t.d[1] = PartDesc{1, 1, g.chunks, 1, sums + C, 0, 1.f};
    part_reduce(ctx.stream, part, t, 2);
  }
}

// dx = k1*gg - k2 - x*k3 with k1 = sc, k3 = sc*rstd*s1/R, k2 = sc*s0/R - mean*k3      (training)
template <int DT, int VE>
__global__ __launch_bounds__(256) void bn_bwd_apply_k(const void* dy, const void* x, void* dx, long rows, int C, int tpr,
                                                      int rpp, int rpc, const float* mean, const float* rstd,
                                                      const float* sc, const float* sh, const float* sums, int relu,
                                                      int has_bn, int training) {
  constexpr int UNR = UNR2;
  STRIP_PROLOGUE(lmin_d(rows, (long)(blockIdx.x + 1) * rpc), (long)blockIdx.x * rpc)
  const float inv = 1.f / (float)rows;
  for (int vc0 = 0; vc0 < nvr; vc0 += tpr) {
    const int vc = vc0 + tc;
    if (!(tr < rpp && vc < nvr)) continue;
    float a[VE], bsh[VE], k1[VE], k2[VE], k3[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      const int c = vc * VE + e;
      if (has_bn) {
        a[e] = sc[c]; bsh[e] = sh[c]; k1[e] = sc[c];
        if (training) { k3[e] = sc[c] * rstd[c] * sums[C + c] * inv; k2[e] = sc[c] * sums[c] * inv - mean[c] * k3[e]; }
        else { k3[e] = 0.f; k2[e] = 0.f; }
      } else { a[e] = 1.f; bsh[e] = 0.f; k1[e] = 1.f; k2[e] = 0.f; k3[e] = 0.f; }
    }
    for (long r = r_begin + tr; r < r_end; r += (long)rpp * UNR) {
      float g[UNR][VE], t[UNR][VE];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const long rr = r + (long)u * rpp;
        const long rc = rr < r_end ? rr : r_end - 1;
        ldv<DT, VE>(dy, rc * C + vc * VE, g[u]);
        ldv<DT, VE>(x, rc * C + vc * VE, t[u]);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const long rr = r + (long)u * rpp;
        if (rr < r_end) {
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            float gg = g[u][e];
            if (relu && !(t[u][e] * a[e] + bsh[e] > 0.f)) gg = 0.f;
            g[u][e] = k1[e] * gg - k2[e] - t[u][e] * k3[e];
          }
          stv<DT, VE>(dx, rr * C + vc * VE, g[u]);
        }
      }
    }
  }
}

void bn_bwd_apply(const Ctx& ctx, const void* dy, const void* x, void* dx, long rows, int C, const float* mean,
                  const float* rstd, const float* sc, const float* sh, const float* sums, int relu, int has_bn, int training) {
  const int ve = col_ve(ctx, C);
  ColGeom g = col_geom(UNR2, C, ve, rows, 1);
  if (stream_cap()) { int cap = 4096; COL_CAPACITY(cap, ctx, ve, bn_bwd_apply_k, 0); g = col_geom(UNR2, C, ve, rows, 1, cap, true); }
  COL_DISPATCH(ctx, ve, bn_bwd_apply_k, dim3(g.chunks), 0, dy, x, dx, rows, C, g.tpr, g.rpp, g.rpc, mean, rstd, sc, sh, sums,
               relu, has_bn, training);
}

// ---- scale_cols: y = x * (add + colw[b][c]) -----------------------------------------------------------
template <int DT, int VE>
__global__ __launch_bounds__(256) void scale_cols_k(const void* x, void* y, int N, int C, int tpr, int rpp, int rpc,
                                                    const float* colw, float add) {
  constexpr int UNR = UNR1;
  const int b = blockIdx.y;
  STRIP_PROLOGUE(lmin_d(N, (long)(blockIdx.x + 1) * rpc), (long)blockIdx.x * rpc)
  for (int vc0 = 0; vc0 < nvr; vc0 += tpr) {
    const int vc = vc0 + tc;
    if (!(tr < rpp && vc < nvr)) continue;
    float cw[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) cw[e] = add + colw[(long)b * C + vc * VE + e];
    for (long n = r_begin + tr; n < r_end; n += (long)rpp * UNR) {
      float t[UNR][VE];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const long nn = n + (long)u * rpp;
        ldv<DT, VE>(x, ((long)b * N + (nn < r_end ? nn : r_end - 1)) * C + vc * VE, t[u]);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const long nn = n + (long)u * rpp;
        if (nn < r_end) {
#pragma unroll
          for (int e = 0; e < VE; ++e) t[u][e] *= cw[e];
          stv<DT, VE>(y, ((long)b * N + nn) * C + vc * VE, t[u]);
        }
      }
    }
  }
}
void scale_cols(const Ctx& ctx, const void* x, void* y, int B, int N, int C, const float* colw, float add) {
  const int ve = col_ve(ctx, C);
  ColGeom g = col_geom(UNR1, C, ve, N, B);
  COL_DISPATCH(ctx, ve, scale_cols_k, dim3(g.chunks, B), 0, x, y, N, C, g.tpr, g.rpp, g.rpc, colw, add);
}

// ---- outer_rows: y[b][n][c] = roww[b][n] * colw[b][c] (write-only stream) ----------------------------------
template <int DT, int VE>
__global__ __launch_bounds__(256) void outer_rows_k(void* y, int N, int C, int tpr, int rpp, int rpc, const float* roww,
                                                    const float* colw) {
  const int b = blockIdx.y;
  STRIP_PROLOGUE(lmin_d(N, (long)(blockIdx.x + 1) * rpc), (long)blockIdx.x * rpc)
  for (int vc0 = 0; vc0 < nvr; vc0 += tpr) {
    const int vc = vc0 + tc;
    if (!(tr < rpp && vc < nvr)) continue;
    float cw[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) cw[e] = colw[(long)b * C + vc * VE + e];
    for (long n = r_begin + tr; n < r_end; n += rpp) {
      const float w = roww[(long)b * N + n];
      float t[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) t[e] = w * cw[e];
      stv<DT, VE>(y, ((long)b * N + n) * C + vc * VE, t);
    }
  }
}
void outer_rows(const Ctx& ctx, const float* roww, const float* colw, int B, int N, int C, void* y) {
  const int ve = col_ve(ctx, C);
  ColGeom g = col_geom(UNR1, C, ve, N, B);
  COL_DISPATCH(ctx, ve, outer_rows_k, dim3(g.chunks, B), 0, y, N, C, g.tpr, g.rpp, g.rpc, roww, colw);
}

// ---- relu_bwd_scale: y = (x > 0) * roww[b][n] * colw[b][c] * colw2[c] * scale; optional colsum_out[c] += sum y ------
template <int DT, int VE>
__global__ __launch_bounds__(256) void relu_bwd_scale_k(const void* x, void* y, int N, int C, int tpr, int rpp, int rpc,
                                                        const float* roww, const void* colw, int cdt,
                                                        const float* colw2, float scale, float* colsum_out, float* part) {
  constexpr int UNR = UNR1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.y;
  STRIP_PROLOGUE(lmin_d(N, (long)(blockIdx.x + 1) * rpc), (long)blockIdx.x * rpc)
  for (int vc0 = 0; vc0 < nvr; vc0 += tpr) {
    const int vc = vc0 + tc;
    const bool active = tr < rpp && vc < nvr;
    float acc[1][VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[0][e] = 0.f;
    if (active) {
      float cw[VE];
      ldv_rt<VE>(colw, cdt, (long)b * C + vc * VE, cw);
#pragma unroll
      for (int e = 0; e < VE; ++e) cw[e] *= scale;
      if (colw2) {
        float t2[VE];
        ldv<DT_F32, VE>(colw2, vc * VE, t2);
#pragma unroll
        for (int e = 0; e < VE; ++e) cw[e] *= t2[e];
      }
      for (long n = r_begin + tr; n < r_end; n += (long)rpp * UNR) {
        float t[UNR][VE], rw[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const long nn = n + (long)u * rpp;
          const long nc = nn < r_end ? nn : r_end - 1;
          ldv<DT, VE>(x, ((long)b * N + nc) * C + vc * VE, t[u]);
          rw[u] = roww ? roww[(long)b * N + nc] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const long nn = n + (long)u * rpp;
          if (nn < r_end) {
#pragma unroll
            for (int e = 0; e < VE; ++e) t[u][e] = t[u][e] > 0.f ? rw[u] * cw[e] : 0.f;
            stv<DT, VE>(y, ((long)b * N + nn) * C + vc * VE, t[u]);
            if (colsum_out) {      // sum what was STORED (the rounded values the weight-gradient GEMM will read)
#pragma unroll
              for (int e = 0; e < VE; ++e) acc[0][e] += DT == DT_BF16 ? bf2f(f2bf(t[u][e])) : t[u][e];
            }
          }
        }
      }
    }
    if (colsum_out) {
      float* const dst[1] = {colsum_out};
      flush_strip<1, VE>(acc, lds, C, vc * VE, active, dst, tr, rpp, vc0 * VE, imin_d(C, (vc0 + tpr) * VE), part);
    }
  }
}
void relu_bwd_scale(const Ctx& ctx, const void* x, void* y, int B, int N, int C, const float* roww, const void* colw, int cdt,
                    const float* colw2, float scale, float* colsum_out, float* part, long part_floats) {
  const int ve = col_ve(ctx, C);
  int cap = 4096;
  if (colsum_out) COL_CAPACITY(cap, ctx, ve, relu_bwd_scale_k, strip_lds(C, ve));
  ColGeom g = col_geom(UNR1, C, ve, N, B, cap, colsum_out != nullptr);
  // partial sums pay from ~150 k atomics up (480 workgroups x 512 channels: 26.1 -> 16.4 + 4.7 us; x 128 channels: no gain)
  if (!colsum_out || !strip_part() || (long)g.chunks * B * C > part_floats || (long)g.chunks * B * C < strip_part_min()) part = nullptr;
  COL_DISPATCH(ctx, ve, relu_bwd_scale_k, dim3(g.chunks, B), strip_lds(C, ve), x, y, N, C, g.tpr, g.rpp, g.rpc, roww,
               colw, cdt, colw2, scale, colsum_out, part);
  if (part) {
    PartTable t; t.NQ = 1; t.C = C;
    t.d[0] = PartDesc{0, 1, g.chunks * B, 1, colsum_out, 0, 1.f};
    if (ctx.late) { ctx.late->part = part; ctx.late->t = t; ctx.late->n = 1; }   // a bias gradient: nothing downstream reads it
    else part_reduce(ctx.stream, part, t, 1);
  }
}

// ---- xc_bwd: dX1 += dXc*(1+ch); dch += sum_n dXc*X1 ------------------------------------------------------
template <int DT, int VE>
__global__ __launch_bounds__(256) void xc_bwd_k(const void* dXc, const void* X1, void* dX1, int N, int C, const float* ch,
                                                int tpr, int rpp, int rpc, float* dch) {
  constexpr int UNR = UNR3;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.y;
  STRIP_PROLOGUE(lmin_d(N, (long)(blockIdx.x + 1) * rpc), (long)blockIdx.x * rpc)
  for (int vc0 = 0; vc0 < nvr; vc0 += tpr) {
    const int vc = vc0 + tc;
    const bool active = tr < rpp && vc < nvr;
    float acc[1][VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[0][e] = 0.f;
    if (active) {
      float cv[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) cv[e] = 1.f + ch[(long)b * C + vc * VE + e];
      for (long n = r_begin + tr; n < r_end; n += (long)rpp * UNR) {
        float g[UNR][VE], x[UNR][VE], d[UNR][VE];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const long nn = n + (long)u * rpp;
          const long o = ((long)b * N + (nn < r_end ? nn : r_end - 1)) * C + vc * VE;
          ldv<DT, VE>(dXc, o, g[u]); ldv<DT, VE>(X1, o, x[u]); ldv<DT, VE>(dX1, o, d[u]);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const long nn = n + (long)u * rpp;
          if (nn < r_end) {
#pragma unroll
            for (int e = 0; e < VE; ++e) { acc[0][e] += g[u][e] * x[u][e]; d[u][e] += g[u][e] * cv[e]; }
            stv<DT, VE>(dX1, ((long)b * N + nn) * C + vc * VE, d[u]);
          }
        }
      }
    }
    float* const dst[1] = {dch + (long)b * C};
    flush_strip<1, VE>(acc, lds, C, vc * VE, active, dst, tr, rpp, vc0 * VE, imin_d(C, (vc0 + tpr) * VE));
  }
}
void xc_bwd(const Ctx& ctx, const void* dXc, const void* X1, void* dX1, int B, int N, int C, const float* ch, float* dch) {
  const int ve = col_ve(ctx, C);
  int cap = 768;
  COL_CAPACITY(cap, ctx, ve, xc_bwd_k, strip_lds(C, ve));
  ColGeom g = col_geom(UNR3, C, ve, N, B, cap, true);
  COL_DISPATCH(ctx, ve, xc_bwd_k, dim3(g.chunks, B), strip_lds(C, ve), dXc, X1, dX1, N, C, ch, g.tpr, g.rpp, g.rpc, dch);
}

// ---- sum over the batch axis of small fp32 tensors: out[i] (+)= scale * sum_b in[b*bs + i] ---------------
// grid (n/256, batch slices); every slice adds its partial with one atomic, so `out` must hold the value to
// accumulate onto (zero for a plain sum: the callers' gradient buffer is pre-zeroed).
__global__ __launch_bounds__(256) void sum_batch_k(const float* in, long bs, int B, long n, float* out, float scale, int bper) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int b0 = blockIdx.y * bper, b1 = imin_d(B, b0 + bper);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;          // four independent loads in flight (the loop is pure latency)
  int b = b0;
  for (; b + 3 < b1; b += 4) {
    s0 += in[(long)b * bs + i]; s1 += in[(long)(b + 1) * bs + i]; s2 += in[(long)(b + 2) * bs + i]; s3 += in[(long)(b + 3) * bs + i];
  }
  for (; b < b1; ++b) s0 += in[(long)b * bs + i];
  unsafeAtomicAdd(out + i, ((s0 + s1) + (s2 + s3)) * scale);
}
void sum_batch(const Ctx& ctx, const float* in, long bs, int B, long n, float* out, float scale, int accumulate) {
  if (!accumulate) (void)hipMemsetAsync(out, 0, (size_t)n * sizeof(float), STREAM(ctx));
  const int bx = (int)cdiv(n, 256);
  int slices = (int)cdiv(1024, bx);
  if (slices > 128) slices = 128;      // every slice ends in one atomic per element: > ~128 per address costs more than the loop
  if (slices > B) slices = B;
  if (slices < 1) slices = 1;
  const int bper = (int)cdiv(B, slices);
  slices = (int)cdiv(B, bper);
  hipLaunchKernelGGL(sum_batch_k, dim3(bx, slices), dim3(256), 0, STREAM(ctx), in, bs, B, n, out, scale, bper);
}

}  // namespace dgsct
