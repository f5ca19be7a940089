// Gate application of the post-backbone TemporalAttention (reference
// DG-SCT/AVE/nets/net_trans.py:240-251): the two scalar-per-timestep sigmoid gates and the three outputs in one pass.
//   ga = sigmoid(akv . wa + ba), gv = sigmoid(vkv . wv + bv)                      (nn.Sequential(Linear(d_model, 1), Sigmoid))
//   out_v = vq * (1 + gamma * ga), out_a = aq * (1 + gamma * gv), gate = ga * gv
// Rows are (timestep, clip) pairs of d_model = 256 fp32 features (the reference runs this head in fp32): one wavefront per row,
// float4 loads, DPP/shuffle row reductions; the backward's per-channel sums (d wa, d wv) are reduced over the rows of a
// workgroup in registers/LDS and leave as one atomic per channel per workgroup.
#include <hip/hip_runtime.h>
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

}  // namespace dgsct
