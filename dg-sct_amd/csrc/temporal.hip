// Gate application of the post-backbone TemporalAttention (reference
// DG-SCT/AVE/nets/net_trans.py:240-251): the two scalar-per-timestep sigmoid gates and the three outputs in one pass.
//   ga = sigmoid(akv . wa + ba), gv = sigmoid(vkv . wv + bv)                      (nn.Sequential(Linear(d_model, 1), Sigmoid))
//   out_v = vq * (1 + gamma * ga), out_a = aq * (1 + gamma * gv), gate = ga * gv
// Rows are (timestep, clip) pairs of d_model = 256 fp32 features (the reference runs this head in fp32): one wavefront per row,
// float4 loads, DPP/shuffle row reductions; the backward's per-channel sums (d wa, d wv) are reduced over the rows of a
// workgroup in registers/LDS and leave as one atomic per channel per workgroup.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "prims.h"
#include "device_util.h"
#include "err.h"

namespace dgsct {

struct TGateArgs {
  int R, D; float gamma;
  const float *akv, *vkv, *vq, *aq, *wa, *ba, *wv, *bv;
  float *out_v, *out_a, *gate, *ga, *gv;
  const float *dOv, *dOa, *dg;
  float *dakv, *dvkv, *dvq, *daq, *dwa, *dba, *dwv, *dbv;
};

__global__ __launch_bounds__(256) void tgate_fwd_k(const TGateArgs p) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.R) return;
  const long o = (long)row * p.D;
  float sa = 0.f, sv = 0.f;
  for (int c = lane * 4; c < p.D; c += 256) {
    const float4 a = *reinterpret_cast<const float4*>(p.akv + o + c), w = *reinterpret_cast<const float4*>(p.wa + c);
    const float4 v = *reinterpret_cast<const float4*>(p.vkv + o + c), u = *reinterpret_cast<const float4*>(p.wv + c);
    sa += a.x * w.x + a.y * w.y + a.z * w.z + a.w * w.w;
    sv += v.x * u.x + v.y * u.y + v.z * u.z + v.w * u.w;
  }
  group_sum2(sa, sv, 64);
  const float ga = 1.f / (1.f + expf(-(sa + *p.ba))), gv = 1.f / (1.f + expf(-(sv + *p.bv)));
  const float fv = 1.f + p.gamma * ga, fa = 1.f + p.gamma * gv;
  for (int c = lane * 4; c < p.D; c += 256) {
    float4 x = *reinterpret_cast<const float4*>(p.vq + o + c), y = *reinterpret_cast<const float4*>(p.aq + o + c);
    x.x *= fv; x.y *= fv; x.z *= fv; x.w *= fv;
    y.x *= fa; y.y *= fa; y.z *= fa; y.w *= fa;
    *reinterpret_cast<float4*>(p.out_v + o + c) = x;
    *reinterpret_cast<float4*>(p.out_a + o + c) = y;
  }
  if (lane == 0) { p.gate[row] = ga * gv; p.ga[row] = ga; p.gv[row] = gv; }
}

// rows are strided over the workgroups so that each thread keeps its channels: thread -> 4 channels (lane*4 + 256*i), wave -> rows
__global__ __launch_bounds__(256) void tgate_bwd_k(const TGateArgs p) {
  __shared__ float red[2][4][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float accA[16], accV[16];                      // per-channel sums of this lane: up to D = 1024 (4 chunks of 4)
#pragma unroll
  for (int i = 0; i < 16; ++i) { accA[i] = 0.f; accV[i] = 0.f; }
  float sba = 0.f, sbv = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < p.R; row += gridDim.x * 4) {
    const long o = (long)row * p.D;
    const float ga = p.ga[row], gv = p.gv[row];
    float da = 0.f, dv = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane * 4 + 256 * i;
      if (c >= p.D) break;
      const float4 gov = *reinterpret_cast<const float4*>(p.dOv + o + c), x = *reinterpret_cast<const float4*>(p.vq + o + c);
      const float4 goa = *reinterpret_cast<const float4*>(p.dOa + o + c), y = *reinterpret_cast<const float4*>(p.aq + o + c);
      da += gov.x * x.x + gov.y * x.y + gov.z * x.z + gov.w * x.w;
      dv += goa.x * y.x + goa.y * y.y + goa.z * y.z + goa.w * y.w;
      const float fv = 1.f + p.gamma * ga, fa = 1.f + p.gamma * gv;
      *reinterpret_cast<float4*>(p.dvq + o + c) = make_float4(gov.x * fv, gov.y * fv, gov.z * fv, gov.w * fv);
      *reinterpret_cast<float4*>(p.daq + o + c) = make_float4(goa.x * fa, goa.y * fa, goa.z * fa, goa.w * fa);
    }
    group_sum2(da, dv, 64);
    const float dgr = p.dg ? p.dg[row] : 0.f;
    const float dpa = (p.gamma * da + dgr * gv) * ga * (1.f - ga), dpv = (p.gamma * dv + dgr * ga) * gv * (1.f - gv);
    if (lane == 0) { sba += dpa; sbv += dpv; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane * 4 + 256 * i;
      if (c >= p.D) break;
      const float4 a = *reinterpret_cast<const float4*>(p.akv + o + c), w = *reinterpret_cast<const float4*>(p.wa + c);
      const float4 v = *reinterpret_cast<const float4*>(p.vkv + o + c), u = *reinterpret_cast<const float4*>(p.wv + c);
      *reinterpret_cast<float4*>(p.dakv + o + c) = make_float4(dpa * w.x, dpa * w.y, dpa * w.z, dpa * w.w);
      *reinterpret_cast<float4*>(p.dvkv + o + c) = make_float4(dpv * u.x, dpv * u.y, dpv * u.z, dpv * u.w);
      accA[4 * i] += dpa * a.x; accA[4 * i + 1] += dpa * a.y; accA[4 * i + 2] += dpa * a.z; accA[4 * i + 3] += dpa * a.w;
      accV[4 * i] += dpv * v.x; accV[4 * i + 1] += dpv * v.y; accV[4 * i + 2] += dpv * v.z; accV[4 * i + 3] += dpv * v.w;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c >= p.D) break;
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][wave][c + e] = accA[4 * i + e]; red[1][wave][c + e] = accV[4 * i + e]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < p.D; c += 256) {
    unsafeAtomicAdd(p.dwa + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    unsafeAtomicAdd(p.dwv + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
  if (lane == 0) { unsafeAtomicAdd(p.dba, sba); unsafeAtomicAdd(p.dbv, sbv); }
}

static bool tgate_ok(int R, int D) {
  if (R <= 0 || D <= 0 || D % 4 || D > 1024) { set_error("temporal gate: D=%d must be a multiple of 4 and <= 1024, R=%d > 0", D, R); return false; }
  return true;
}
void temporal_gate_fwd(const Ctx& ctx, int R, int D, float gamma, const float* akv, const float* vkv, const float* vq, const float* aq,
                       const float* wa, const float* ba, const float* wv, const float* bv, float* out_v, float* out_a, float* gate,
                       float* ga, float* gv) {
  if (!tgate_ok(R, D)) return;
  TGateArgs p{};
  p.R = R; p.D = D; p.gamma = gamma; p.akv = akv; p.vkv = vkv; p.vq = vq; p.aq = aq; p.wa = wa; p.ba = ba; p.wv = wv; p.bv = bv;
  p.out_v = out_v; p.out_a = out_a; p.gate = gate; p.ga = ga; p.gv = gv;
  hipLaunchKernelGGL(tgate_fwd_k, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)ctx.stream, p);
}
void temporal_gate_bwd(const Ctx& ctx, int R, int D, float gamma, const float* akv, const float* vkv, const float* vq, const float* aq,
                       const float* wa, const float* wv, const float* ga, const float* gv, const float* dOv, const float* dOa,
                       const float* dg, float* dakv, float* dvkv, float* dvq, float* daq, float* dwa, float* dba, float* dwv, float* dbv) {
  if (!tgate_ok(R, D)) return;
  TGateArgs p{};
  p.R = R; p.D = D; p.gamma = gamma; p.akv = akv; p.vkv = vkv; p.vq = vq; p.aq = aq; p.wa = wa; p.wv = wv; p.ga = const_cast<float*>(ga); p.gv = const_cast<float*>(gv);
  p.dOv = dOv; p.dOa = dOa; p.dg = dg; p.dakv = dakv; p.dvkv = dvkv; p.dvq = dvq; p.daq = daq; p.dwa = dwa; p.dba = dba; p.dwv = dwv; p.dbv = dbv;
  hipStream_t s = (hipStream_t)ctx.stream;
  (void)hipMemsetAsync(dwa, 0, (size_t)D * 4, s);
  (void)hipMemsetAsync(dwv, 0, (size_t)D * 4, s);
  (void)hipMemsetAsync(dba, 0, 4, s);
  (void)hipMemsetAsync(dbv, 0, 4, s);
  int wgs = (R + 3) / 4; if (wgs > 64) wgs = 64;
  hipLaunchKernelGGL(tgate_bwd_k, dim3(wgs), dim3(256), 0, s, p);
}

// ---- per-frame scalar gate on a feature block (TemporalAttention of AVVP mgn.py:155-156 and of the AVS decoder scales
// PVT_AVSModel.py:572-577): y[r][i] = x[r][i] * (1 + gamma * g[r]), x one frame's [inner] block ([128] features or a [C,H,W] map).
// Forward: one pass, 16-byte accesses.  Backward: dx = dy * (1 + gamma g[r]) and dg[r] = gamma * sum_i dy x in the same pass
// (a frame is split over `chunks` workgroups; one fp32 atomic per workgroup into the pre-zeroed dg).
template <int DT>
__global__ __launch_bounds__(256) void frame_scale_fwd_k(const void* x, const float* g, void* y, long inner, float gamma, int chunks) {
  constexpr int V = El<DT>::VMAX;
  const long r = blockIdx.y;
  const float sc = 1.f + gamma * g[r];
  const long n = inner / V, per = (n + chunks - 1) / chunks;
  const long i0 = (long)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) {
    float v[V];
    ldv<DT, V>(x, r * inner + i * V, v);
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] *= sc;
    stv<DT, V>(y, r * inner + i * V, v);
  }
  if (blockIdx.x == chunks - 1)
    for (long i = n * V + threadIdx.x; i < inner; i += 256) ste<DT>(y, r * inner + i, lde<DT>(x, r * inner + i) * sc);
}
template <int DT>
__global__ __launch_bounds__(256) void frame_scale_bwd_k(const void* x, const float* g, const void* dy, void* dx, float* dg, long inner,
                                                         float gamma, int chunks) {
  constexpr int V = El<DT>::VMAX;
  __shared__ float red[8];
  const long r = blockIdx.y;
  const float sc = 1.f + gamma * g[r];
  const long n = inner / V, per = (n + chunks - 1) / chunks;
  const long i0 = (long)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
  float acc = 0.f;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) {
    float a[V], b[V];
    ldv<DT, V>(dy, r * inner + i * V, a);
    ldv<DT, V>(x, r * inner + i * V, b);
#pragma unroll
    for (int e = 0; e < V; ++e) { acc += a[e] * b[e]; a[e] *= sc; }
    if (dx) stv<DT, V>(dx, r * inner + i * V, a);
  }
  if (blockIdx.x == chunks - 1)
    for (long i = n * V + threadIdx.x; i < inner; i += 256) {
      const float a = lde<DT>(dy, r * inner + i);
      acc += a * lde<DT>(x, r * inner + i);
      if (dx) ste<DT>(dx, r * inner + i, a * sc);
    }
  const float t = block_sum(acc, red);
  if (threadIdx.x == 0 && dg) unsafeAtomicAdd(dg + r, gamma * t);
}
static int fs_chunks(int rows, long inner, int V) {
  long c = (1024 + rows - 1) / rows;                    // >= ~1024 workgroups, each with >= 2048 vectors
  const long maxc = inner / V / 2048 + 1;
  if (c > maxc) c = maxc;
  return (int)(c < 1 ? 1 : c);
}
static inline bool fs_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
void frame_scale_fwd(const Ctx& ctx, int rows, long inner, float gamma, const void* x, const float* g, void* y) {
  if (rows <= 0 || inner <= 0) return;
  if (!fs_al16(x) || !fs_al16(y)) { set_error("frame_scale: x / y must be 16-byte aligned"); return; }
  const bool al = (inner % (ctx.mode == DT_BF16 ? 8 : 4)) == 0;
  if (!al) { set_error("frame_scale: the per-frame block (%ld elements) must be a multiple of 16 bytes", inner); return; }
  const int ch = fs_chunks(rows, inner, ctx.mode == DT_BF16 ? 8 : 4);
  if (ctx.mode == DT_BF16) hipLaunchKernelGGL(frame_scale_fwd_k<DT_BF16>, dim3(ch, rows), dim3(256), 0, (hipStream_t)ctx.stream, x, g, y, inner, gamma, ch);
  else hipLaunchKernelGGL(frame_scale_fwd_k<DT_F32>, dim3(ch, rows), dim3(256), 0, (hipStream_t)ctx.stream, x, g, y, inner, gamma, ch);
}
void frame_scale_bwd(const Ctx& ctx, int rows, long inner, float gamma, const void* x, const float* g, const void* dy, void* dx, float* dg) {
  if (rows <= 0 || inner <= 0) return;
  if (!fs_al16(x) || !fs_al16(dy) || (dx && !fs_al16(dx))) { set_error("frame_scale: x / dy / dx must be 16-byte aligned"); return; }
  const bool al = (inner % (ctx.mode == DT_BF16 ? 8 : 4)) == 0;
  if (!al) { set_error("frame_scale: the per-frame block (%ld elements) must be a multiple of 16 bytes", inner); return; }
  hipStream_t s = (hipStream_t)ctx.stream;
  if (dg) (void)hipMemsetAsync(dg, 0, (size_t)rows * 4, s);
  const int ch = fs_chunks(rows, inner, ctx.mode == DT_BF16 ? 8 : 4);
  if (ctx.mode == DT_BF16) hipLaunchKernelGGL(frame_scale_bwd_k<DT_BF16>, dim3(ch, rows), dim3(256), 0, s, x, g, dy, dx, dg, inner, gamma, ch);
  else hipLaunchKernelGGL(frame_scale_bwd_k<DT_F32>, dim3(ch, rows), dim3(256), 0, s, x, g, dy, dx, dg, inner, gamma, ch);
}

}  // namespace dgsct
