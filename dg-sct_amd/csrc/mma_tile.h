// LDS-tile building blocks of the fused (flash-style) kernels: staging of row-major global slabs into LDS and
// 32x32 MFMA tile products between two LDS images, for both arithmetic modes of the library:
//   DT_BF16: v_mfma_f32_32x32x16_bf16 (operand fragments = 8 bf16 per lane), fp32 accumulate
//   DT_F32 : v_mfma_f32_32x32x2_f32   (operand fragments = 1 float per lane; exact fp32 FMA chain) -- the parity path
// An LDS image is always ROW-MAJOR [rows][cols] with a byte pitch.  It can feed an MFMA operand two ways:
//   KM ("k-major")  : the operand's M/N index is the image ROW, the contraction runs ALONG the row
//                     (bf16: one ds_read_b128 per fragment; pitch = cols*2 + 16 keeps the 16-lane groups conflict-free)
//   MN ("mn-major") : the operand's M/N index is the image COLUMN, the contraction runs ACROSS rows
//                     (bf16: two ds_read_b64_tr_b16 transpose reads per fragment; pitch == 64 (mod 128) bytes)
// so a token-major activation slab [tokens][channels] serves `X . W^T` (KM) and `P^T . X` (MN) without ever being
// transposed in memory.
#pragma once
#include <hip/hip_runtime.h>
#include "device_util.h"

namespace dgsct {

typedef __attribute__((ext_vector_type(8))) __bf16 mt_bf16x8;
typedef __attribute__((ext_vector_type(4))) short mt_s16x4;
typedef __attribute__((ext_vector_type(8))) short mt_s16x8;
typedef __attribute__((ext_vector_type(16))) float mt_f32x16;

__host__ __device__ constexpr int mt_mn_pitch_bf16(int cols) {
  int b = cols * 2;
  int v = (b / 128) * 128 + 64;
  return v >= b ? v : v + 128;
}
template <int MODE> struct MT {
  static constexpr int ES = MODE == DT_BF16 ? 2 : 4;
  static constexpr int VE = 16 / ES;
  static constexpr int KSTEP = MODE == DT_BF16 ? 16 : 2;
  __host__ __device__ static constexpr int km_pitch(int cols) { return MODE == DT_BF16 ? cols * 2 + 16 : cols * 4 + 16; }
  __host__ __device__ static constexpr int mn_pitch(int cols) { return MODE == DT_BF16 ? mt_mn_pitch_bf16(cols) : cols * 4; }
};

// ---- global (element type of MODE) -> LDS image, [TR][TC] elements, 16-byte chunks.  Rows >= rows_valid and columns >=
// cols_valid arrive as zeros.  Loads are UNCONDITIONAL from clamped addresses and the mask is applied to the register
// (a guarded load compiles to branch + load + s_waitcnt vmcnt(0) per chunk: serial round trips; gemm.hip FAST staging).
// cols_valid - c0 and ld must be multiples of the chunk width (8 bf16 / 4 fp32) -- checked on the host.
template <int MODE, int TR, int TC>
__device__ __forceinline__ void stage_tile(char* lds, int pitch, const void* g, long ld, int r0, int rows_valid, int c0,
                                           int cols_valid, int tid) {
  constexpr int ES = MT<MODE>::ES, VE = MT<MODE>::VE;
  constexpr int CPR = TC / VE, NCH = TR * CPR, NIT = (NCH + 255) / 256;
  static_assert(TC % VE == 0, "tile width must be whole 16-byte chunks");
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = it * 256 + tid;
    if (NCH % 256 != 0 && i >= NCH) break;
    const int r = i / CPR, c = (i % CPR) * VE;
    const bool ok = r0 + r < rows_valid && c0 + c < cols_valid;
    const int rr = r0 + r < rows_valid ? r0 + r : rows_valid - 1;
    const int cc = c0 + c < cols_valid ? c0 + c : cols_valid - VE;
    uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(g) + ((long)rr * ld + cc) * ES);
    if (!ok) v = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(lds + r * pitch + c * ES) = v;
  }
}
// fp32 global [rows][ld] -> LDS image(s) of MODE's element type.  bf16: `hi` = bf16(x) and (LO) `lo` = bf16(x - hi): the
// pair carries ~16 mantissa bits through two bf16 MFMAs (used for the un-scaled attention logits, whose softmax amplifies
// operand rounding).  fp32 mode: plain copy into `hi`.
template <int MODE, int TR, int TC, bool LO>
__device__ __forceinline__ void stage_tile_f32(char* hi, char* lo, int pitch, const float* g, long ld, int r0, int rows_valid,
                                               int c0, int cols_valid, int tid) {
  constexpr int CPR = TC / 4, NCH = TR * CPR, NIT = (NCH + 255) / 256;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = it * 256 + tid;
    if (NCH % 256 != 0 && i >= NCH) break;
    const int r = i / CPR, c = (i % CPR) * 4;
    const bool ok = r0 + r < rows_valid && c0 + c < cols_valid;
    const int rr = r0 + r < rows_valid ? r0 + r : rows_valid - 1;
    const int cc = c0 + c < cols_valid ? c0 + c : cols_valid - 4;
    float4 v = *reinterpret_cast<const float4*>(g + (long)rr * ld + cc);
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == DT_F32) {
      *reinterpret_cast<float4*>(hi + r * pitch + c * 4) = v;
    } else {
      const unsigned short h0 = f2bf(v.x), h1 = f2bf(v.y), h2 = f2bf(v.z), h3 = f2bf(v.w);
      *reinterpret_cast<uint2*>(hi + r * pitch + c * 2) = make_uint2((unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16));
      if (LO) {
        const unsigned short l0 = f2bf(v.x - bf2f(h0)), l1 = f2bf(v.y - bf2f(h1)), l2 = f2bf(v.z - bf2f(h2)), l3 = f2bf(v.w - bf2f(h3));
        *reinterpret_cast<uint2*>(lo + r * pitch + c * 2) = make_uint2((unsigned)l0 | ((unsigned)l1 << 16), (unsigned)l2 | ((unsigned)l3 << 16));
      }
    }
  }
}

// ---- operand fragments -------------------------------------------------------------------------------------------
// bf16 KM: lane l holds image[row0 + (l & 31)][16 kk + 8 (l >> 5) .. +7]
__device__ __forceinline__ mt_bf16x8 mt_frag_km(const char* img, int pitch, int row0, int kk, int lane) {
  return *reinterpret_cast<const mt_bf16x8*>(img + (row0 + (lane & 31)) * pitch + (kk * 16 + (lane >> 5) * 8) * 2);
}
// bf16 MN: lane l holds image[16 kk + 8 (l >> 5) .. +7][col0 + (l & 31)] through two transpose reads (each 16-lane group
// reads a [4 k][16 col] block; the lane supplies the address of 4 consecutive columns of ONE k-row and receives 4
// consecutive k of ONE column)
__device__ __forceinline__ mt_bf16x8 mt_frag_mn(const char* img, int pitch, int col0, int kk, int lane) {
  const int kbase = kk * 16 + (lane >> 5) * 8 + ((lane & 15) >> 2);
  const int c = col0 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
  const char* p0 = img + kbase * pitch + c * 2;
  mt_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) mt_s16x4*)(p0));
  mt_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) mt_s16x4*)(p0 + 4 * pitch));
  mt_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(mt_bf16x8, v);
}

// acc[32 x 32] += sum_{k < K} A(m, k) * B(n, k).  A: image `A`, pitch pa, its 32 M-indices start at a0 (rows if AKM, columns
// otherwise); B likewise.  K is a multiple of 16 (bf16) / 2 (fp32).  Accumulator element r of a lane:
// m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), n = lane & 31.
template <int MODE, bool AKM, bool BKM>
__device__ __forceinline__ void mma_tile(mt_f32x16& acc, const char* A, int pa, int a0, const char* B, int pb, int b0, int K,
                                         int lane) {
  if (MODE == DT_BF16) {
    for (int kk = 0; kk < K / 16; ++kk) {
      const mt_bf16x8 af = AKM ? mt_frag_km(A, pa, a0, kk, lane) : mt_frag_mn(A, pa, a0, kk, lane);
      const mt_bf16x8 bf = BKM ? mt_frag_km(B, pb, b0, kk, lane) : mt_frag_mn(B, pb, b0, kk, lane);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
    }
  } else {
    for (int kk = 0; kk < K / 2; ++kk) {
      const int k = kk * 2 + (lane >> 5);
      const float af = AKM ? *reinterpret_cast<const float*>(A + (a0 + (lane & 31)) * pa + k * 4)
                           : *reinterpret_cast<const float*>(A + k * pa + (a0 + (lane & 31)) * 4);
      const float bf = BKM ? *reinterpret_cast<const float*>(B + (b0 + (lane & 31)) * pb + k * 4)
                           : *reinterpret_cast<const float*>(B + k * pb + (b0 + (lane & 31)) * 4);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);
    }
  }
}
__device__ __forceinline__ int mt_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

}  // namespace dgsct
