// Grouped projections with a tiny per-group width (bottleneck adapters at the early stages: C = 96/128, ds = C/8 = 12/16,
// g = 2 -> dg = ds/g = 6/8).  As MFMA GEMMs these have N or K = 6..8: a 32-wide tile is 80 % padding and the launch is
// a pure HBM stream that the GEMM engine runs at ~1.1 TB/s.  Here they are row kernels on the vector units: a
// power-of-two group of lanes owns one token row (16 B per lane), the dg x VE weight block of the lane lives in
// registers, and the row never leaves the register file between load, dot products and store.
//   narrow: y[r][gi*dg + jl] = sum_cl x[r][gi*cg + cl] * W(gi, jl, cl)     [rows][C]  -> [rows][ds]
//   wide:   y[r][gi*cg + cl] = sum_jl x[r][gi*dg + jl] * W(gi, jl, cl)     [rows][ds] -> [rows][C]  (+ BatchNorm sums)
// W(gi, jl, cl) = W[gi*sg + jl*sj + cl*sc] addresses the fp32 master weight in either role (forward: down/up sampler,
// backward: the same tensors transposed), so no transposed copy exists.   Reference: net_trans.py:629-643.
#include <cstdlib>
#include <hip/hip_runtime.h>
#include "prims.h"
#include "device_util.h"
#include "err.h"

namespace dgsct {

#define STREAM(ctx) ((hipStream_t)(ctx).stream)
static inline long cdivl(long a, long b) { return (a + b - 1) / b; }

constexpr int PROJ_DG_MAX = 16;     // max per-group narrow width (weight block of a lane in registers: DG x VE floats)

struct ProjGeom { int ve, gs, rpp, rpc, chunks; };
static bool proj_geom(int mode, int C, int ds, int g, long rows, long target_wgs, ProjGeom& pg) {
  if (g < 1 || C % g || ds % g) return false;
  const int cg = C / g, dg = ds / g;
  pg.ve = mode == DT_BF16 ? 8 : 4;
  if (dg < 1 || dg > PROJ_DG_MAX || cg % pg.ve || (dg & 1)) return false;
  const int nvec = C / pg.ve;
  pg.gs = 1;
  while (pg.gs < nvec) pg.gs <<= 1;
  if (pg.gs > 64 || ds > pg.gs || (long)ds * cg > 4096) return false;
  pg.rpp = 256 / pg.gs;
  long chunks = target_wgs;
  const long maxc = cdivl(rows, pg.rpp);
  if (chunks > maxc) chunks = maxc;
  if (chunks < 1) chunks = 1;
  pg.rpc = (int)(cdivl(cdivl(rows, chunks), pg.rpp) * pg.rpp);
  pg.chunks = (int)cdivl(rows, pg.rpc);
  return true;
}

bool gproj_supported(int mode, int C, int ds, int g) {
  static const int off = getenv("DGSCT_NO_GPROJ") ? atoi(getenv("DGSCT_NO_GPROJ")) : 0;
  ProjGeom pg;
  return !off && proj_geom(mode, C, ds, g, 1 << 20, 1024, pg);
}

// ---- narrow ---------------------------------------------------------------------------------------------------------
// Per row: every lane forms its dg partial dot products; the lanes of a channel group are combined through LDS (the
// row group sits inside one wavefront, whose DS instructions execute in order: no workgroup barrier).
// UNR rows per lane per trip: all loads are issued before the first use (latency hiding).  DG = 8 (stage 0: dg = 6, 8)
// keeps 4 rows in flight; DG = 16 (stage 1: dg = 12, 16) has a 128-register weight block and runs 2 rows at 2 WG/CU.

template <int DT, int VE, int PROJ_DG, int PROJ_UNR>
__global__ __launch_bounds__(256, PROJ_DG > 8 ? 2 : 3) void gproj_narrow_k(const void* x, long rows, int C, int ds, int g, const float* W, long sg,
                                                      long sj, long sc, int gs, int rpc, void* y) {
  // [UNR][row slot][lane][DG] partial dot products (+8 floats per row slot: spreads the slots over the banks)
  __shared__ __attribute__((aligned(16))) float lds[PROJ_UNR * (256 * PROJ_DG + 128 * 8)];
  const int gl = threadIdx.x & (gs - 1), sub = threadIdx.x / gs, rpp = 256 / gs;
  const int cg = C / g, dg = ds / g, lpg = cg / VE;
  const int col = gl * VE;
  const bool valid = col < C;
  const int colc = valid ? col : 0;
  const int gi = colc / cg, cl0 = colc - gi * cg;
  // weights: staged through LDS by the whole workgroup (ds*cg <= 4096 floats), then each lane keeps its dg x VE block
  // in registers.  (Per-lane global loads of 64 scattered floats compile to 64 serial branch+load+wait round trips.)
  for (int i = threadIdx.x; i < ds * cg; i += 256) {
    const int cl = i % cg, j = i / cg, gq = j / dg, jl = j - gq * dg;
    lds[i] = W[gq * sg + jl * sj + (long)cl * sc];
  }
  __syncthreads();
  float w[PROJ_DG][VE];
#pragma unroll
  for (int jl = 0; jl < PROJ_DG; ++jl)
#pragma unroll
    for (int e = 0; e < VE; ++e) w[jl][e] = (valid && jl < dg) ? lds[(gi * dg + jl) * cg + cl0 + e] : 0.f;
  __syncthreads();                                    // lds is reused for the partial sums below
  const int slot_stride = gs * PROJ_DG + 8;
  float* const slot0 = lds + sub * slot_stride;
  const int unr_stride = 256 * PROJ_DG + 128 * 8;
  const int oq = gl / dg, ojl = gl - oq * dg;           // the output element this lane reduces (gl < ds)
  const long r0 = (long)blockIdx.x * rpc;
  const long r_end = lmin_d(rows, r0 + rpc);
  for (long rb = r0; rb < r_end; rb += (long)rpp * PROJ_UNR) {
    // Unconditional loads from clamped addresses: a per-element "load or zero" makes hipcc branch around every load and
    // wait vmcnt(0) after each one, which serialises the four HBM round trips of a trip.
    float t[PROJ_UNR][VE];
#pragma unroll
    for (int u = 0; u < PROJ_UNR; ++u) {
      const long row = rb + (long)u * rpp + sub;
      ldv<DT, VE>(x, (row < r_end ? row : r_end - 1) * C + colc, t[u]);
    }
#pragma unroll
    for (int u = 0; u < PROJ_UNR; ++u) {
      const bool ok = valid && rb + (long)u * rpp + sub < r_end;
#pragma unroll
      for (int e = 0; e < VE; ++e) t[u][e] = ok ? t[u][e] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < PROJ_UNR; ++u) {
      float p[PROJ_DG];
#pragma unroll
      for (int jl = 0; jl < PROJ_DG; ++jl) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < VE; ++e) a += t[u][e] * w[jl][e];
        p[jl] = a;
      }
      float4* dst = reinterpret_cast<float4*>(slot0 + u * unr_stride + gl * PROJ_DG);
#pragma unroll
      for (int q4 = 0; q4 < PROJ_DG / 4; ++q4) dst[q4] = make_float4(p[4 * q4], p[4 * q4 + 1], p[4 * q4 + 2], p[4 * q4 + 3]);
    }
    // the row group lives inside one wavefront: its DS instructions execute in order, only the compiler must not move
    // the reads above the writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int u = 0; u < PROJ_UNR; ++u) {
      const long row = rb + (long)u * rpp + sub;
      if (gl < ds && row < r_end) {
        const float* src = slot0 + u * unr_stride + oq * lpg * PROJ_DG + ojl;
        float s = 0.f;
        for (int i = 0; i < lpg; ++i) s += src[i * PROJ_DG];
        ste<DT>(y, row * ds + gl, s);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

void gproj_narrow(const Ctx& ctx, const void* x, long rows, int C, int ds, int g, const float* W, long sg, long sj, long sc,
                  void* y) {
  ProjGeom pg;
  if (!proj_geom(ctx.mode, C, ds, g, rows, 2048, pg)) { set_error("gproj_narrow: unsupported shape C=%d ds=%d g=%d", C, ds, g); return; }
  const bool big = ds / g > 8;
#define NARROW_(DT_, VE_, DG_, UNR_) \
  hipLaunchKernelGGL((gproj_narrow_k<DT_, VE_, DG_, UNR_>), dim3(pg.chunks), dim3(256), 0, STREAM(ctx), x, rows, C, ds, g, W, sg, sj, sc, \
                     pg.gs, pg.rpc, y)
  if (ctx.mode == DT_BF16) { if (big) NARROW_(DT_BF16, 8, 16, 2); else NARROW_(DT_BF16, 8, 8, 4); }
  else { if (big) NARROW_(DT_F32, 4, 16, 2); else NARROW_(DT_F32, 4, 8, 4); }
#undef NARROW_
}

// ---- wide -----------------------------------------------------------------------------------------------------------
// stats != null: the bn_stats accumulators of y AS STORED (rounded to E): stats[0..C) = y[0][c] (shift),
// stats[C..2C) += sum (y - shift), stats[2C..3C) += sum (y - shift)^2.
template <int DT, int VE, int PROJ_DG, int PROJ_UNR>
__global__ __launch_bounds__(256, PROJ_DG > 8 ? 2 : 3) void gproj_wide_k(const void* x, long rows, int C, int ds, int g, const float* W, long sg,
                                                    long sj, long sc, int gs, int rpc, void* y, float* stats) {
  __shared__ float lds[4096];
  const int gl = threadIdx.x & (gs - 1), sub = threadIdx.x / gs, rpp = 256 / gs;
  const int cg = C / g, dg = ds / g;
  const int col = gl * VE;
  const bool valid = col < C;
  const int colc = valid ? col : 0;
  const int gi = colc / cg, cl0 = colc - gi * cg;
  for (int i = threadIdx.x; i < ds * cg; i += 256) {      // weights through LDS (see gproj_narrow_k)
    const int cl = i % cg, j = i / cg, gq = j / dg, jl = j - gq * dg;
    lds[i] = W[gq * sg + jl * sj + (long)cl * sc];
  }
  __syncthreads();
  float w[VE][PROJ_DG];
#pragma unroll
  for (int e = 0; e < VE; ++e)
#pragma unroll
    for (int jl = 0; jl < PROJ_DG; ++jl) w[e][jl] = (valid && jl < dg) ? lds[(gi * dg + jl) * cg + cl0 + e] : 0.f;
  __syncthreads();                                        // lds is reused by the statistics flush
  auto project = [&](long row, float (&o)[VE]) {
    float xin[PROJ_DG];
#pragma unroll
    for (int jl = 0; jl < PROJ_DG; ++jl) {
      const float v = lde<DT>(x, row * ds + gi * dg + (jl < dg ? jl : 0));
      xin[jl] = jl < dg ? v : 0.f;
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      float s = 0.f;
#pragma unroll
      for (int jl = 0; jl < PROJ_DG; ++jl) s += xin[jl] * w[e][jl];
      o[e] = s;
    }
  };
  auto rounded = [](float v) { return DT == DT_BF16 ? bf2f(f2bf(v)) : v; };
  float sft[VE], acc[2][VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) { sft[e] = 0.f; acc[0][e] = acc[1][e] = 0.f; }
  if (stats && valid) {
    float o[VE];
    project(0, o);
#pragma unroll
    for (int e = 0; e < VE; ++e) sft[e] = rounded(o[e]);
    if (blockIdx.x == 0 && sub == 0) {
#pragma unroll
      for (int e = 0; e < VE; ++e) stats[col + e] = sft[e];
    }
  }
  const long r0 = (long)blockIdx.x * rpc;
  const long r_end = lmin_d(rows, r0 + rpc);
  if (valid) {
    for (long rb = r0 + sub; rb < r_end; rb += (long)rpp * PROJ_UNR) {
      // all narrow-row loads of the trip first (dg values = 12..32 bytes per row, shared by the lanes of a group)
      float xin[PROJ_UNR][PROJ_DG];
#pragma unroll
      for (int u = 0; u < PROJ_UNR; ++u) {
        const long row = rb + (long)u * rpp;
        const bool ok = row < r_end;
        if (DT == DT_BF16) {
          const unsigned* px = reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(x) + (ok ? row : r0) * ds + gi * dg);
#pragma unroll
          for (int j2 = 0; j2 < PROJ_DG / 2; ++j2) {
            const unsigned v = px[2 * j2 < dg ? j2 : 0];            // unconditional load, clamped index (w is 0 beyond dg)
            xin[u][2 * j2] = __uint_as_float(v << 16);
            xin[u][2 * j2 + 1] = __uint_as_float(v & 0xffff0000u);
          }
        } else {
          const float* px = reinterpret_cast<const float*>(x) + (ok ? row : r0) * ds + gi * dg;
#pragma unroll
          for (int jl = 0; jl < PROJ_DG; ++jl) xin[u][jl] = px[jl < dg ? jl : 0];
        }
      }
#pragma unroll
      for (int u = 0; u < PROJ_UNR; ++u) {
        const long row = rb + (long)u * rpp;
        if (row >= r_end) continue;
        float o[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          float a = 0.f;
#pragma unroll
          for (int jl = 0; jl < PROJ_DG; ++jl) a += xin[u][jl] * w[e][jl];
          o[e] = a;
        }
        stv<DT, VE>(y, row * C + col, o);
        if (stats) {
#pragma unroll
          for (int e = 0; e < VE; ++e) { const float d = rounded(o[e]) - sft[e]; acc[0][e] += d; acc[1][e] += d * d; }
        }
      }
    }
  }
  if (stats) {          // combine the row slots of the workgroup in LDS (store / barrier / column sum), one atomic per channel
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      __syncthreads();
      if (valid) {
#pragma unroll
        for (int e = 0; e < VE; e += 4)
          *reinterpret_cast<float4*>(&lds[sub * C + col + e]) = make_float4(acc[q][e], acc[q][e + 1], acc[q][e + 2], acc[q][e + 3]);
      }
      __syncthreads();
      for (int i = threadIdx.x; i < C; i += 256) {
        float s = 0.f;
        for (int r = 0; r < rpp; ++r) s += lds[r * C + i];
        unsafeAtomicAdd(stats + C + q * C + i, s);
      }
    }
  }
}

void gproj_wide(const Ctx& ctx, const void* x, long rows, int C, int ds, int g, const float* W, long sg, long sj, long sc,
                void* y, float* stats) {
  ProjGeom pg;
  long target = 2048;
  if (stats) {          // a reduction: one full round of resident workgroups (see wg_capacity)
    const bool big0 = ds / g > 8;
    const void* fn = ctx.mode == DT_BF16
        ? (big0 ? reinterpret_cast<const void*>(&gproj_wide_k<DT_BF16, 8, 16, 2>) : reinterpret_cast<const void*>(&gproj_wide_k<DT_BF16, 8, 8, 8>))
        : (big0 ? reinterpret_cast<const void*>(&gproj_wide_k<DT_F32, 4, 16, 2>) : reinterpret_cast<const void*>(&gproj_wide_k<DT_F32, 4, 8, 4>));
    target = wg_capacity(fn, 0);
    if (target > 1024) target = 1024;
  }
  if (!proj_geom(ctx.mode, C, ds, g, rows, target, pg) || C > 512) {
    set_error("gproj_wide: unsupported shape C=%d ds=%d g=%d", C, ds, g);
    return;
  }
  const bool big = ds / g > 8;
#define WIDE_(DT_, VE_, DG_, UNR_) \
  hipLaunchKernelGGL((gproj_wide_k<DT_, VE_, DG_, UNR_>), dim3(pg.chunks), dim3(256), 0, STREAM(ctx), x, rows, C, ds, g, W, sg, sj, sc, \
                     pg.gs, pg.rpc, y, stats)
  if (ctx.mode == DT_BF16) { if (big) WIDE_(DT_BF16, 8, 16, 2); else WIDE_(DT_BF16, 8, 8, 8); }
  else { if (big) WIDE_(DT_F32, 4, 16, 2); else WIDE_(DT_F32, 4, 8, 4); }
#undef WIDE_
}

}  // namespace dgsct
