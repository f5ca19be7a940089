// Device-side helpers shared by the gfx950 kernels: bf16 <-> fp32, 16-byte vector element access,
// sub-wavefront (power-of-two lane group) reductions.  Wavefront = 64 lanes on CDNA4.
#pragma once
#include <hip/hip_runtime.h>
#include "prims.h"

namespace dgsct {

// Resident-workgroup capacity of the device for one kernel instantiation (256-thread workgroups): occupancy x CUs.
// Reduction kernels size their grid to ONE full round of resident workgroups: 800 workgroups on 768 slots take two
// rounds (the second one 4 % full), and every extra workgroup costs an LDS combine + one global atomic per channel.
inline int wg_capacity(const void* fn, size_t shmem) {
  struct Key { const void* f; size_t s; };
  static thread_local Key keys[64];
  static thread_local int vals[64];
  static thread_local int n = 0;
  for (int i = 0; i < n; ++i)
    if (keys[i].f == fn && keys[i].s == shmem) return vals[i];
  int per_cu = 0, dev = 0, cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, shmem) != hipSuccess || per_cu < 1) per_cu = 2;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int cap = per_cu * cus;
  if (n < 64) { keys[n] = Key{fn, shmem}; vals[n] = cap; ++n; }
  return cap;
}

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
// round-to-nearest-even (same rounding as torch.Tensor.to(torch.bfloat16)): gfx950's v_cvt_pk_bf16_f32, one instruction per PAIR --
// the integer add-and-shift version was 5-6 VALU instructions per element, a third of the instruction stream of the row kernels
typedef __attribute__((ext_vector_type(2))) float du_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 du_bf16x2_t;
__device__ __forceinline__ unsigned f2bf2(float lo, float hi) {
  const du_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, du_bf16x2_t));
}
__device__ __forceinline__ unsigned short f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

template <int DT> struct El;
template <> struct El<DT_F32>  { static constexpr int ES = 4; static constexpr int VMAX = 4; };
template <> struct El<DT_BF16> { static constexpr int ES = 2; static constexpr int VMAX = 8; };

template <int DT>
__device__ __forceinline__ float lde(const void* p, long i) {
  if (DT == DT_F32) return reinterpret_cast<const float*>(p)[i];
  return bf2f(reinterpret_cast<const unsigned short*>(p)[i]);
}
template <int DT>
__device__ __forceinline__ void ste(void* p, long i, float v) {
  if (DT == DT_F32) reinterpret_cast<float*>(p)[i] = v;
  else reinterpret_cast<unsigned short*>(p)[i] = f2bf(v);
}
__device__ __forceinline__ float lde_rt(const void* p, int dt, long i) {
  return dt == DT_F32 ? reinterpret_cast<const float*>(p)[i] : bf2f(reinterpret_cast<const unsigned short*>(p)[i]);
}
__device__ __forceinline__ void ste_rt(void* p, int dt, long i, float v) {
  if (dt == DT_F32) reinterpret_cast<float*>(p)[i] = v;
  else reinterpret_cast<unsigned short*>(p)[i] = f2bf(v);
}

// VE consecutive elements starting at element index i (i*ES must be VE*ES-aligned for VE > 1).
template <int DT, int VE>
__device__ __forceinline__ void ldv(const void* p, long i, float (&v)[VE]) {
  if (DT == DT_F32) {
    const float* q = reinterpret_cast<const float*>(p) + i;
    if (VE == 4) { float4 t = *reinterpret_cast<const float4*>(q); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else {
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] = q[e];
    }
  } else {
    const unsigned short* q = reinterpret_cast<const unsigned short*>(p) + i;
    if (VE == 8) {
      uint4 t = *reinterpret_cast<const uint4*>(q);
      unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
    } else if (VE == 4) {
      uint2 t = *reinterpret_cast<const uint2*>(q);
      unsigned w[2] = {t.x, t.y};
#pragma unroll
      for (int e = 0; e < 2; ++e) { v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
    } else {
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] = bf2f(q[e]);
    }
  }
}
template <int DT, int VE>
__device__ __forceinline__ void stv(void* p, long i, const float (&v)[VE]) {
  if (DT == DT_F32) {
    float* q = reinterpret_cast<float*>(p) + i;
    if (VE == 4) *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
    else {
#pragma unroll
      for (int e = 0; e < VE; ++e) q[e] = v[e];
    }
  } else {
    unsigned short* q = reinterpret_cast<unsigned short*>(p) + i;
    if (VE == 8) {
      unsigned w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = f2bf2(v[2 * e], v[2 * e + 1]);
      *reinterpret_cast<uint4*>(q) = make_uint4(w[0], w[1], w[2], w[3]);
    } else if (VE == 4) {
      unsigned w[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) w[e] = f2bf2(v[2 * e], v[2 * e + 1]);
      *reinterpret_cast<uint2*>(q) = make_uint2(w[0], w[1]);
    } else {
#pragma unroll
      for (int e = 0; e < VE; ++e) q[e] = f2bf(v[e]);
    }
  }
}
// run-time element type, ONE branch around a vector load (lde_rt per element is a branch + load + wait each)
template <int VE>
__device__ __forceinline__ void ldv_rt(const void* p, int dt, long i, float (&v)[VE]) {
  if (dt == DT_F32) ldv<DT_F32, VE>(p, i, v);
  else ldv<DT_BF16, VE>(p, i, v);
}
// raw (still packed) VE-element vector in a uint4 -- lets a kernel issue next-row loads early at 4 registers apiece
template <int DT, int VE>
__device__ __forceinline__ uint4 ldraw(const void* p, long i) {
  constexpr int BYTES = VE * El<DT>::ES;
  const char* q = reinterpret_cast<const char*>(p) + i * El<DT>::ES;
  if (BYTES == 16) return *reinterpret_cast<const uint4*>(q);
  if (BYTES == 8) { const uint2 t = *reinterpret_cast<const uint2*>(q); return make_uint4(t.x, t.y, 0, 0); }
  uint4 r = make_uint4(0, 0, 0, 0);
  if (BYTES == 4) r.x = *reinterpret_cast<const unsigned*>(q);
  else r.x = *reinterpret_cast<const unsigned short*>(q);
  return r;
}
template <int DT, int VE>
__device__ __forceinline__ void unpack(const uint4& r, float (&v)[VE]) {
  const unsigned w[4] = {r.x, r.y, r.z, r.w};
  if (DT == DT_F32) {
#pragma unroll
    for (int e = 0; e < VE; ++e) v[e] = __uint_as_float(w[e]);
  } else {
#pragma unroll
    for (int e = 0; e < VE; ++e) v[e] = (e & 1) ? __uint_as_float(w[e >> 1] & 0xffff0000u) : __uint_as_float(w[e >> 1] << 16);
  }
}
// VE consecutive fp32 values
template <int VE>
__device__ __forceinline__ void ldf(const float* p, long i, float (&v)[VE]) {
  const float* q = p + i;
  if (VE % 4 == 0) {
#pragma unroll
    for (int e = 0; e < VE; e += 4) { float4 t = *reinterpret_cast<const float4*>(q + e); v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w; }
  } else {
#pragma unroll
    for (int e = 0; e < VE; ++e) v[e] = q[e];
  }
}

// sum over a power-of-two group of GS lanes (GS <= 64) that is aligned inside the wavefront
// Reductions over aligned groups of gs = 2^k lanes (result in every lane of the group).  Within a 16-lane row the
// exchange is DPP (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: operand modifiers of the add itself); only the
// 16 / 32 steps go through ds_bpermute.  The all-bpermute loop cost 6 LDS-crossbar round trips (each followed by
// s_waitcnt lgkmcnt(0)) per sum, twice per row of the LayerNorm kernels.  Every lane of a quad / half-row / row holds
// bit-identical partial sums after each step, so the result equals the xor butterfly's.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float group_sum(float x, int gs) {
  if (gs >= 16) {
    x += dpp_f<0xB1>(x);      // quad_perm [1,0,3,2]
    x += dpp_f<0x4E>(x);      // quad_perm [2,3,0,1]
    x += dpp_f<0x141>(x);     // row_half_mirror
    x += dpp_f<0x140>(x);     // row_mirror
    if (gs >= 32) x += __shfl_xor(x, 16, 64);
    if (gs >= 64) x += __shfl_xor(x, 32, 64);
    return x;
  }
  for (int o = gs >> 1; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
// two independent sums at once: the exchanges of one overlap the adds of the other, and the bpermute pair shares a wait
__device__ __forceinline__ void group_sum2(float& a, float& b, int gs) {
  if (gs >= 16) {
    a += dpp_f<0xB1>(a);  b += dpp_f<0xB1>(b);
    a += dpp_f<0x4E>(a);  b += dpp_f<0x4E>(b);
    a += dpp_f<0x141>(a); b += dpp_f<0x141>(b);
    a += dpp_f<0x140>(a); b += dpp_f<0x140>(b);
    if (gs >= 32) { const float ta = __shfl_xor(a, 16, 64), tb = __shfl_xor(b, 16, 64); a += ta; b += tb; }
    if (gs >= 64) { const float ta = __shfl_xor(a, 32, 64), tb = __shfl_xor(b, 32, 64); a += ta; b += tb; }
    return;
  }
  for (int o = gs >> 1; o > 0; o >>= 1) { const float ta = __shfl_xor(a, o, 64), tb = __shfl_xor(b, o, 64); a += ta; b += tb; }
}
__device__ __forceinline__ float group_max(float x, int gs) {
  if (gs >= 16) {
    x = fmaxf(x, dpp_f<0xB1>(x));
    x = fmaxf(x, dpp_f<0x4E>(x));
    x = fmaxf(x, dpp_f<0x141>(x));
    x = fmaxf(x, dpp_f<0x140>(x));
    if (gs >= 32) x = fmaxf(x, __shfl_xor(x, 16, 64));
    if (gs >= 64) x = fmaxf(x, __shfl_xor(x, 32, 64));
    return x;
  }
  for (int o = gs >> 1; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}
// sum over the 256-thread workgroup; result valid in every thread.  red: >= 4 floats of LDS.
__device__ __forceinline__ float block_sum(float x, float* red) {
  x = group_sum(x, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float x, float* red) {
  x = group_max(x, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ int imin_d(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ long lmin_d(long a, long b) { return a < b ? a : b; }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// scale / shift of the VE BatchNorm channels c0 .. c0 + VE - 1 from the batch sums (bn_finalize_k's arithmetic, see BnFin in prims.h);
// `owner`: exactly one thread of the launch per channel also stores the four vectors and updates the running statistics.
// All inputs are fetched as vectors BEFORE anything is stored: channel by channel, the owner's stores (which may alias the inputs as
// far as the compiler knows) turned the loads of every thread of the launch into VE serial round trips (+20 us per kernel).
template <int VE>
__device__ __forceinline__ void bn_fin_vec(const BnFin& f, int C, int c0, bool owner, float (&s)[VE], float (&t)[VE]) {
  float m[VE], v[VE], w[VE], b[VE];
  ldf<VE>(f.w, c0, w);
  ldf<VE>(f.b, c0, b);
  if (f.training) {
    float sft[VE], s1[VE], s2[VE];
    ldf<VE>(f.acc, c0, sft);
    ldf<VE>(f.acc, (long)C + c0, s1);
    ldf<VE>(f.acc, 2L * C + c0, s2);
    const float rows = (float)f.rows;
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      const float a1 = s1[e] / rows, a2 = s2[e] / rows;       // (divisions, as bn_finalize_k: the variance below cancels, a last-bit difference shows)
      m[e] = sft[e] + a1;
      v[e] = fmaxf(a2 - a1 * a1, 0.f);
    }
  } else {
    ldf<VE>(f.run_mean, c0, m);
    ldf<VE>(f.run_var, c0, v);
  }
  float rs[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    rs[e] = rsqrtf(v[e] + f.eps);
    s[e] = w[e] * rs[e];
    t[e] = b[e] - m[e] * s[e];
  }
  if (owner) {
    if (f.training) {
      float rm[VE], rv[VE];
      ldf<VE>(f.run_mean, c0, rm);
      ldf<VE>(f.run_var, c0, rv);
      const float k = f.rows > 1 ? (float)f.rows / (float)(f.rows - 1) : 1.f;
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        f.run_mean[c0 + e] = (1.f - f.momentum) * rm[e] + f.momentum * m[e];
        f.run_var[c0 + e] = (1.f - f.momentum) * rv[e] + f.momentum * (v[e] * k);
      }
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) { f.mean[c0 + e] = m[e]; f.rstd[c0 + e] = rs[e]; f.sc[c0 + e] = s[e]; f.sh[c0 + e] = t[e]; }
  }
}

// ---- second stage of the per-channel sums: dst[j][c] += scale * sum_{k<K} part[(j*K + k)][q][c] ---------------------
static __global__ __launch_bounds__(256) void part_reduce_k(const float* part, PartTable t) {
  const PartDesc d = t.d[blockIdx.z];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if ((int)blockIdx.y >= d.J * d.S || c >= t.C) return;
  const int j = blockIdx.y / d.S, sp = blockIdx.y - j * d.S;
  const int kb = (d.K + d.S - 1) / d.S, k0 = sp * kb, k1 = k0 + kb < d.K ? k0 + kb : d.K;
  const long rs = (long)t.NQ * t.C;
  const float* p = part + ((long)j * d.K) * rs + (long)d.q * t.C + c;
  // (a dependent chain of launches waits for this kernel: every load of a thread's <= ~12 partials is in flight at once -- with 4 per
  //  trip and up to 23 trips' worth the launch took 14 us for 7 MB)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
  int k = k0;
  for (; k + 7 < k1; k += 8) {
    const float v0 = p[k * rs], v1 = p[(k + 1) * rs], v2 = p[(k + 2) * rs], v3 = p[(k + 3) * rs];
    const float v4 = p[(k + 4) * rs], v5 = p[(k + 5) * rs], v6 = p[(k + 6) * rs], v7 = p[(k + 7) * rs];
    s0 += v0; s1 += v1; s2 += v2; s3 += v3; s4 += v4; s5 += v5; s6 += v6; s7 += v7;
  }
  for (; k + 3 < k1; k += 4) { s0 += p[k * rs]; s1 += p[(k + 1) * rs]; s2 += p[(k + 2) * rs]; s3 += p[(k + 3) * rs]; }
  for (; k < k1; ++k) s0 += p[k * rs];
  if (k1 > k0) unsafeAtomicAdd(d.dst + (long)j * d.dst_stride + c, d.scale * (((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7))));
}
static inline void part_reduce(void* stream, const float* part, PartTable& t, int n) {
  if (n == 0) return;
  int ymax = 1;
  for (int i = 0; i < n; ++i) {
    PartDesc& d = t.d[i];
    d.S = d.K / 12; if (d.S < 1) d.S = 1; if (d.S > 64) d.S = 64;
    if (d.J * d.S > ymax) ymax = d.J * d.S;
  }
  hipLaunchKernelGGL(part_reduce_k, dim3((t.C + 255) / 256, ymax, n), dim3(256), 0, (hipStream_t)stream, part, t);
}

}  // namespace dgsct
