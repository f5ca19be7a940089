// fp8 (OCP e4m3) MFMA projections -- BASELINE.json configs[4] "bf16 + fp8-MFMA projections" (SURVEY.md section 7).
//
//   D[m][n] = act( inv_scale * sum_k fp8(A[m][k]) * W8[n][k] + bias[n] )        D, A: bf16 token-major; W8: fp8 [N][K]
//
// Used (dtype DGSCT_BF16_FP8) for the three big weight-stationary forward projections of the adapter: fc (second remap
// GEMM), fc_affine_video_1 and fc_affine_video_2 (net_trans.py:554, 594, 602).  Weights are quantised once per
// parameter update by dgsct_prepare with a per-tensor scale s = 448 / max|W| (inv_scale = 1 / s); activations are
// O(1) token maps and are converted un-scaled (saturating at +-448) ON THE WAY INTO LDS, so no fp8 copy of an activation
// is ever written to HBM and an LDS tile is half the bytes of its bf16 counterpart.  v_mfma_f32_32x32x16_fp8_fp8, fp32
// accumulate; 128 x 128 x 64 tiles, 4 wavefronts (2 x 2), register prefetch of the next k-tile; token rows leave through
// LDS as 16-byte stores.
#include <hip/hip_runtime.h>
#include "prims.h"
#include "device_util.h"
#include "err.h"

namespace dgsct {

namespace {
typedef __attribute__((ext_vector_type(16))) float f8_f32x16;
constexpr int F8_BM = 128, F8_BN = 128, F8_BK = 64, F8_PITCH = F8_BK + 8;     // LDS pitch in bytes (fp8): 72 -> conflict-free b64 rows

__device__ __forceinline__ float clamp448(float x) { return fminf(fmaxf(x, -448.f), 448.f); }
// 8 bf16 (one 16-byte chunk) -> 8 fp8 e4m3 (8 bytes)
__device__ __forceinline__ uint2 bf16x8_to_fp8(const uint4& v) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(__uint_as_float(w[0] << 16)), clamp448(__uint_as_float(w[0] & 0xffff0000u)), lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(__uint_as_float(w[1] << 16)), clamp448(__uint_as_float(w[1] & 0xffff0000u)), lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(__uint_as_float(w[2] << 16)), clamp448(__uint_as_float(w[2] & 0xffff0000u)), hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(__uint_as_float(w[3] << 16)), clamp448(__uint_as_float(w[3] & 0xffff0000u)), hi, true);
  return make_uint2((unsigned)lo, (unsigned)hi);
}
}  // namespace

struct Fp8Args {
  int M, N, K;
  const unsigned short* A; long lda;
  const unsigned char* W8;             // [N][K]
  const float* inv_scale; const float* bias; int relu;
  unsigned short* D; long ldd;
  const float* r1_m; const float* r1_n; int m_mod;      // + r1_m[m % m_mod] * r1_n[n]  (rank-1 bias of the conv remap)
};

__global__ __launch_bounds__(256, 2) void gemm_fp8_k(const Fp8Args p) {
  constexpr int STG = 4 * 32 * (64 + 4) * 4;                                  // epilogue staging: 4 waves x [32][64 + 4] fp32
  constexpr int OPND = 2 * F8_BM * F8_PITCH;
  __shared__ __attribute__((aligned(16))) char smem[STG > OPND ? STG : OPND];
  char* sA = smem;
  char* sB = smem + F8_BM * F8_PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_m = (p.M + F8_BM - 1) / F8_BM;
  const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
  const int m0 = tm * F8_BM, n0 = tn * F8_BN;
  f8_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // staging maps: A tile 128 rows x 64 bf16 = 1024 16-byte chunks (4 per thread); W tile 128 rows x 64 fp8 = 512 chunks (2 per thread)
  uint4 ra[4], rb[2];
  auto load = [&](int kt) {
    const int k0 = kt * F8_BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256, r = c >> 3, k = (c & 7) * 8;
      const int rr = m0 + r < p.M ? m0 + r : p.M - 1;
      const int kk = k0 + k < p.K ? k0 + k : p.K - 8;
      ra[i] = *reinterpret_cast<const uint4*>(p.A + (long)rr * p.lda + kk);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256, r = c >> 2, k = (c & 3) * 16;
      const int rr = n0 + r < p.N ? n0 + r : p.N - 1;
      const int kk = k0 + k < p.K ? k0 + k : p.K - 16;
      rb[i] = *reinterpret_cast<const uint4*>(p.W8 + (long)rr * p.K + kk);
    }
  };
  auto store = [&](int kt) {
    const int k0 = kt * F8_BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256, r = c >> 3, k = (c & 7) * 8;
      uint2 v = bf16x8_to_fp8(ra[i]);
      if (m0 + r >= p.M || k0 + k >= p.K) v = make_uint2(0, 0);
      *reinterpret_cast<uint2*>(sA + r * F8_PITCH + k) = v;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256, r = c >> 2, k = (c & 3) * 16;
      uint4 v = rb[i];
      if (n0 + r >= p.N || k0 + k >= p.K) v = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint2*>(sB + r * F8_PITCH + k) = make_uint2(v.x, v.y);
      *reinterpret_cast<uint2*>(sB + r * F8_PITCH + k + 8) = make_uint2(v.z, v.w);
    }
  };
  const int nkt = (p.K + F8_BK - 1) / F8_BK;
  load(0);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    store(kt);
    __syncthreads();
    if (kt + 1 < nkt) load(kt + 1);
#pragma unroll
    for (int kk = 0; kk < F8_BK / 16; ++kk) {
      long af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = *reinterpret_cast<const long*>(sA + ((wm * 2 + i) * 32 + (lane & 31)) * F8_PITCH + kk * 16 + (lane >> 5) * 8);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bf[j] = *reinterpret_cast<const long*>(sB + ((wn * 2 + j) * 32 + (lane & 31)) * F8_PITCH + kk * 16 + (lane >> 5) * 8);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
  // epilogue: 32-row blocks through LDS, whole token rows out as 16-byte stores
  const float isc = *p.inv_scale;
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * 68);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + (wn * 2 + j) * 32 + (lane & 31);
      const float bn = (p.bias && n < p.N) ? p.bias[n] : 0.f;
      const float r1n = (p.r1_n && n < p.N) ? p.r1_n[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = isc * acc[i][j][r] + bn;
        if (p.r1_m) {
          const int m = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < p.M) v += p.r1_m[m % p.m_mod] * r1n;
        }
        if (p.relu) v = fmaxf(v, 0.f);
        stg[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 68 + j * 32 + (lane & 31)] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int c = it * 64 + lane, row = c >> 3, cc = c & 7;
      const int m = m0 + (wm * 2 + i) * 32 + row, n = n0 + wn * 64 + cc * 8;
      if (m >= p.M || n >= p.N) continue;
      float v[8];
      const float4 a = *reinterpret_cast<const float4*>(stg + row * 68 + cc * 8), b = *reinterpret_cast<const float4*>(stg + row * 68 + cc * 8 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      if (n + 8 <= p.N) stv<DT_BF16, 8>(p.D, (long)m * p.ldd + n, v);
      else
        for (int e = 0; e < 8 && n + e < p.N; ++e) p.D[(long)m * p.ldd + n + e] = f2bf(v[e]);
    }
  }
}

// ---- per-tensor quantisation of a weight: scale = 448 / max|W|, W8 = fp8(W * scale), inv_scale = 1 / scale ---------------
__global__ __launch_bounds__(256) void fp8_amax_k(const float* w, long n, unsigned* amax_bits) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
  m = group_max(m, 64);
  if ((threadIdx.x & 63) == 0) atomicMax(amax_bits, __float_as_uint(m));      // non-negative floats order like their bit patterns
}
__global__ __launch_bounds__(256) void fp8_quant_k(const float* w, long n, const unsigned* amax_bits, unsigned char* out, float* inv_scale) {
  const float amax = __uint_as_float(*amax_bits);
  const float scale = amax > 0.f ? 448.f / amax : 1.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) *inv_scale = 1.f / scale;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    float x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = i + e < n ? clamp448(w[i + e] * scale) : 0.f;
    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], 0, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], pk, true);
    if (i + 4 <= n) *reinterpret_cast<int*>(out + i) = pk;
    else
      for (int e = 0; e < 4 && i + e < n; ++e) out[i + e] = (unsigned char)(pk >> (8 * e));
  }
}
void fp8_quantize(const Ctx& ctx, const float* w, long n, void* out8, float* inv_scale, void* amax_scratch) {
  hipStream_t s = (hipStream_t)ctx.stream;
  (void)hipMemsetAsync(amax_scratch, 0, 4, s);
  int g = (int)((n + 1023) / 1024); if (g > 512) g = 512; if (g < 1) g = 1;
  hipLaunchKernelGGL(fp8_amax_k, dim3(g), dim3(256), 0, s, w, n, (unsigned*)amax_scratch);
  hipLaunchKernelGGL(fp8_quant_k, dim3(g), dim3(256), 0, s, w, n, (const unsigned*)amax_scratch, (unsigned char*)out8, inv_scale);
}
void gemm_fp8(const Ctx& ctx, int M, int N, int K, const void* A, long lda, const void* W8, const float* inv_scale, const float* bias,
              int relu, void* D, long ldd, const float* r1_m, const float* r1_n, int m_mod) {
  if (K % 16 != 0 || lda % 8 != 0 || ldd % 8 != 0 || K < 16) {
    set_error("gemm_fp8: K=%d must be a multiple of 16 and the leading dimensions multiples of 8", K);
    return;
  }
  Fp8Args p{M, N, K, (const unsigned short*)A, lda, (const unsigned char*)W8, inv_scale, bias, relu, (unsigned short*)D, ldd, r1_m, r1_n,
            m_mod > 0 ? m_mod : 1};
  const int tiles = ((M + F8_BM - 1) / F8_BM) * ((N + F8_BN - 1) / F8_BN);
  hipLaunchKernelGGL(gemm_fp8_k, dim3(tiles), dim3(256), 0, (hipStream_t)ctx.stream, p);
}

}  // namespace dgsct
