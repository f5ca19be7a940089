// Batched / grouped / split-K MFMA GEMM engine for gfx950 (MI355X, CDNA4), hand-written.
//
//   D[b][m][n] = epilogue( sum_k A[b][m][k] * B[b][n][k] )          (see prims.h: struct Gemm)
//
// * BF16 mode: v_mfma_f32_32x32x16_bf16, fp32 accumulate.  K-major operands are staged to LDS as
//   [rows][32+8] (80-byte rows: conflict-free ds_read_b128 fragment reads); MN-major operands
//   (token-contraction GEMMs: Wn.Y, every weight gradient) are staged as [32 k-rows][rows+pad]
//   and read with ds_read_b64_tr_b16, the CDNA4 LDS transpose read, so no transposed copy of an
//   activation is ever written to HBM.  Row pitch = 64 (mod 128) bytes keeps the four k-rows of a
//   tr-read lane group on disjoint banks.
// * F32 mode (the parity path): v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain); LDS holds
//   [16 k-rows][rows+4] so either operand layout is one ds_read_b32 per fragment.
// * 256 threads = 4 wavefronts (64 lanes) per workgroup, wave grid WGM x WGN, each wave owns
//   TM x TN 32x32 accumulator tiles.  Global->register prefetch of tile t+1 overlaps the MFMAs of
//   tile t (register staging: the operand roles here need transposes / zero-fill guards that the
//   lane-linear LDS-DMA path cannot express).
// * blockIdx.x is remapped so that the 8 XCDs each walk a contiguous range of tiles (private L2s).
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <vector>
#include "prims.h"
#include "device_util.h"
#include <cmath>

#include "err.h"
#include "gemm_int.h"

namespace dgsct {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct GemmK {
  int M, N, K, KB;
  int tiles_m, tiles_n, kflat, kt_total, kt_per_split; unsigned kinv;
  int nfast;                       // tile order of the work list: n-tiles fastest (else m-tiles)
  int xgm, xnb;                    // batched GEMM with a SHARED A: m-tiles per group / batches per XCD (0: plain order)
  const char* A; long lda, a_bs, a_kbs; int a_vec;
  const char* B; long ldb, b_bs, b_kbs; int b_vec;
  char* D; int ddt; long ldd, dbs;
  float alpha; const float* alpha_ptr;
  const float* bias_m; const float* bias_n; long bias_n_bs; int m_mod;
  const float* r1_m; const float* r1_n;
  int act;
  const char* R; int rdt; long ldr, rbs; float beta;
  const char* R2;
  const char* mask; long ldmask, maskbs;
  const float* sm_scale; float* sm_dot;
  int atomic;
  int wide;
};

// pitch (bytes) of one k-row of an MN-major bf16 LDS tile holding `rows` elements: >= rows*2, == 64 (mod 128)
__host__ __device__ constexpr int mn_pitch_bf16(int rows) {
  int b = rows * 2;
  int v = (b / 128) * 128 + 64;
  return v >= b ? v : v + 128;
}

template <int MODE, bool KM, int ROWS>
struct TileGeom {
  static constexpr int ES = MODE == DT_BF16 ? 2 : 4;
  static constexpr int BKT = MODE == DT_BF16 ? 64 : 16;
  static constexpr int VE = 16 / ES;                       // elements per 16-byte chunk
  static constexpr int NCHUNK = ROWS * BKT / VE;           // chunks per tile
  static constexpr int NLD = (NCHUNK + 255) / 256;         // chunks per thread
  // LDS pitch (bytes)
  static constexpr int PITCH = MODE == DT_BF16 ? (KM ? (BKT + 8) * 2 : mn_pitch_bf16(ROWS)) : (ROWS + 4) * 4;
  static constexpr int LDS_BYTES = MODE == DT_BF16 ? (KM ? ROWS * PITCH : BKT * PITCH) : BKT * PITCH;
};

// ---- global -> register staging of one operand tile --------------------------------------------
// The contraction index is FLAT: kf in [0, Kflat), Kflat = KB*K; kf -> (kb, k) = (kf / K, kf % K) by a
// multiply-high with kinv = ceil(2^32 / K) (exact for kf*K < 2^32), address += kb*kbs.  A k-tile may therefore
// straddle two frames of a two-level contraction (no padding waste when K = 96 and the tile is 64 deep).
__device__ __forceinline__ void split_k(int kf, int K, int Kflat, unsigned kinv, int& kb, int& k) {
  if (Kflat == K) { kb = 0; k = kf; }
  else { kb = kinv ? (int)__umulhi((unsigned)kf, kinv) : kf; k = kf - kb * K; }
}
template <int MODE, bool KM, int ROWS, int NLD_>
__device__ __forceinline__ void stage_load(uint4 (&reg)[NLD_], const char* base, long ld, long kbs, int r0, int rows_total,
                                           int kf0, int K, int Kflat, unsigned kinv, int vec, int tid) {
  using G = TileGeom<MODE, KM, ROWS>;
  constexpr int ES = G::ES, VE = G::VE;
#pragma unroll
  for (int i = 0; i < G::NLD; ++i) {
    int c = tid + i * 256;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (c < G::NCHUNK) {
      int r, k;
      if (KM) { r = c / (G::BKT / VE); k = (c % (G::BKT / VE)) * VE; }
      else    { k = c / (ROWS / VE);   r = (c % (ROWS / VE)) * VE; }
      const int rg = r0 + r, kf = kf0 + k;
      if (rg < rows_total && kf < Kflat) {
        int kb, kk;
        split_k(kf, K, Kflat, kinv, kb, kk);
        const char* bp = base + (long)kb * kbs * ES;
        if (KM) {
          const char* p = bp + ((long)rg * ld + kk) * ES;
          if (vec && kk + VE <= K) v = *reinterpret_cast<const uint4*>(p);
          else {
            unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < VE; ++e)
              if (kf + e < Kflat) {
                int kbe, kke;
                split_k(kf + e, K, Kflat, kinv, kbe, kke);
                const char* q = base + ((long)kbe * kbs + (long)rg * ld + kke) * ES;
                if (ES == 4) w[e] = *reinterpret_cast<const unsigned*>(q);
                else w[e >> 1] |= (unsigned)*reinterpret_cast<const unsigned short*>(q) << ((e & 1) * 16);
              }
            v = make_uint4(w[0], w[1], w[2], w[3]);
          }
        } else {
          const char* p = bp + ((long)kk * ld + rg) * ES;
          if (vec && rg + VE <= rows_total) v = *reinterpret_cast<const uint4*>(p);
          else {
            unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < VE; ++e)
              if (rg + e < rows_total) {
                if (ES == 4) w[e] = reinterpret_cast<const unsigned*>(p)[e];
                else w[e >> 1] |= (unsigned)reinterpret_cast<const unsigned short*>(p)[e] << ((e & 1) * 16);
              }
            v = make_uint4(w[0], w[1], w[2], w[3]);
          }
        }
      }
    }
    reg[i] = v;
  }
}

// ---- register -> LDS ---------------------------------------------------------------------------
template <int MODE, bool KM, int ROWS, int NLD_>
__device__ __forceinline__ void stage_store(const uint4 (&reg)[NLD_], char* lds, int tid) {
  using G = TileGeom<MODE, KM, ROWS>;
  constexpr int VE = G::VE;
#pragma unroll
  for (int i = 0; i < G::NLD; ++i) {
    int c = tid + i * 256;
    if (c < G::NCHUNK) {
      int r, k;
      if (KM) { r = c / (G::BKT / VE); k = (c % (G::BKT / VE)) * VE; }
      else    { k = c / (ROWS / VE);   r = (c % (ROWS / VE)) * VE; }
      if (MODE == DT_BF16) {
        if (KM) *reinterpret_cast<uint4*>(lds + r * G::PITCH + k * 2) = reg[i];
        else    *reinterpret_cast<uint4*>(lds + k * G::PITCH + r * 2) = reg[i];
      } else {
        if (KM) {   // transpose on the way in: LDS is [k][row]
          float* l = reinterpret_cast<float*>(lds);
          const int pitch = G::PITCH / 4;
          l[(k + 0) * pitch + r] = __uint_as_float(reg[i].x);
          l[(k + 1) * pitch + r] = __uint_as_float(reg[i].y);
          l[(k + 2) * pitch + r] = __uint_as_float(reg[i].z);
          l[(k + 3) * pitch + r] = __uint_as_float(reg[i].w);
        } else {
          *reinterpret_cast<uint4*>(lds + k * G::PITCH + r * 4) = reg[i];
        }
      }
    }
  }
}

// ---- FAST staging (bf16, 16-byte chunks never straddle an edge: host-checked) ----------------------------------
// Straight-line code: every chunk is loaded UNCONDITIONALLY from a clamped (always valid) address and the edge mask is
// applied when the registers are written to LDS.  The guarded generic path above compiles to branch + load +
// s_waitcnt vmcnt(0) per chunk (hipcc materialises the select at the merge point), i.e. the chunks of a k-tile are
// fetched one after the other and the register prefetch overlaps nothing.
template <bool KM, int ROWS, int NLD_>
__device__ __forceinline__ void stage_load_fast(uint4 (&reg)[NLD_], const char* base, long ld, long kbs, int r0, int rows_total,
                                                int kf0, int K, int Kflat, unsigned kinv, int tid) {
  using G = TileGeom<DT_BF16, KM, ROWS>;
  constexpr int VE = G::VE;
#pragma unroll
  for (int i = 0; i < G::NLD; ++i) {
    int c = tid + i * 256;
    if (G::NCHUNK % 256 != 0) c = c < G::NCHUNK ? c : G::NCHUNK - 1;
    int r, k;
    if (KM) { r = c / (G::BKT / VE); k = (c % (G::BKT / VE)) * VE; }
    else    { k = c / (ROWS / VE);   r = (c % (ROWS / VE)) * VE; }
    int rg = r0 + r, kf = kf0 + k;
    if (KM) { rg = rg < rows_total ? rg : rows_total - 1; kf = kf < Kflat ? kf : Kflat - VE; }
    else    { rg = rg < rows_total ? rg : rows_total - VE; kf = kf < Kflat ? kf : Kflat - 1; }
    // branch-free (kf, K) -> (frame, k): kinv = ceil(2^32 / K) gives frame 0 for kf < K, so one-level contractions need no
    // special case (a branch here makes hipcc drain vmcnt at its merge point)
    const int kb = (int)__umulhi((unsigned)kf, kinv);
    const int kk = kf - kb * K;
    const char* p = base + ((long)kb * kbs + (KM ? (long)rg * ld + kk : (long)kk * ld + rg)) * 2;
    reg[i] = *reinterpret_cast<const uint4*>(p);
  }
}
template <bool KM, int ROWS, int NLD_>
__device__ __forceinline__ void stage_store_fast(const uint4 (&reg)[NLD_], char* lds, int r0, int rows_total, int kf0, int Kflat,
                                                 int tid) {
  using G = TileGeom<DT_BF16, KM, ROWS>;
  constexpr int VE = G::VE;
#pragma unroll
  for (int i = 0; i < G::NLD; ++i) {
    const int c = tid + i * 256;
    if (G::NCHUNK % 256 != 0 && c >= G::NCHUNK) continue;
    int r, k;
    if (KM) { r = c / (G::BKT / VE); k = (c % (G::BKT / VE)) * VE; }
    else    { k = c / (ROWS / VE);   r = (c % (ROWS / VE)) * VE; }
    const bool ok = r0 + r < rows_total && kf0 + k < Kflat;
    uint4 v = reg[i];
    if (!ok) v = make_uint4(0, 0, 0, 0);
    if (KM) *reinterpret_cast<uint4*>(lds + r * G::PITCH + k * 2) = v;
    else    *reinterpret_cast<uint4*>(lds + k * G::PITCH + r * 2) = v;
  }
}

// ---- LDS-DMA staging (STAGE 2: bf16, FAST conditions + K a multiple of the 64-deep k-tile) ---------------------------
// `global_load_lds_dwordx4` writes 64 lanes x 16 B = 1 KiB CONTIGUOUS bytes of LDS per wave instruction (wave-uniform
// base + lane*16), the global source address is per lane.  So the LDS image is dense (no padding) and bank conflicts
// are avoided by choosing WHICH global chunk each lane fetches (swizzle on the source side):
//  * K-major tile [ROWS][64 k] = 128-byte rows, 8 chunks: chunk c of row r lives at slot c ^ ((r >> 1) & 7).  A 16-lane
//    ds_read_b128 group reads 16 consecutive rows of one k-chunk: row parity picks the 128-byte half of the 256-byte bank
//    line and (r >> 1) & 7 the slot -> 16 distinct 16-byte slots, conflict-free.
//  * MN-major tile [64 k][ROWS] = ROWS*2-byte k-rows: a transpose-read group touches 4 consecutive k-rows x 32 B; with a
//    192-byte (ROWS = 96) or 64-byte (32) pitch those fall on different bank granules already, for 128 / 256 / 512-byte
//    pitches chunk rc of k-row k lives at slot rc ^ (2 * (k & 3)).
// No staging registers, no ds_write pass; the tile needs no edge masking because row reads are clamped (garbage rows
// only feed accumulators that are never stored) and K is a whole number of k-tiles.
template <bool KM, int ROWS>
struct GldsGeom {
  static constexpr int CPR = KM ? 8 : ROWS / 8;          // 16-byte chunks per LDS row
  static constexpr int NSLOT = ROWS * 8;
  static constexpr int NI = NSLOT / 256;                 // DMA instructions per thread
  static constexpr int LDS_BYTES = NSLOT * 16;
  static constexpr bool SWZ_MN = !(ROWS == 32 || ROWS == 96);
  static_assert(NSLOT % 256 == 0, "tile must be a whole number of 4-wave DMA rounds");
};
template <bool KM, int ROWS>
__device__ __forceinline__ void glds_tile(char* lds, const char* base, long ld, long kbs, int r0, int rows_total, int kf0, int K,
                                          unsigned kinv, int wave, int lane) {
  using G = GldsGeom<KM, ROWS>;
#pragma unroll
  for (int j = 0; j < G::NI; ++j) {
    const int sbase = (j * 4 + wave) * 64;               // wave-uniform slot base
    const int s = sbase + lane;
    const int q = s / G::CPR, cpos = s % G::CPR;
    int rg, kf;
    if (KM) {
      const int c = cpos ^ ((q >> 1) & 7);
      rg = r0 + q; rg = rg < rows_total ? rg : rows_total - 1;
      kf = kf0 + c * 8;
    } else {
      const int rc = G::SWZ_MN ? (cpos ^ (2 * (q & 3))) : cpos;
      rg = r0 + rc * 8; rg = rg < rows_total ? rg : rows_total - 8;
      kf = kf0 + q;
    }
    const int kb = (int)__umulhi((unsigned)kf, kinv);
    const int kk = kf - kb * K;
    const char* gp = base + ((long)kb * kbs + (KM ? (long)rg * ld + kk : (long)kk * ld + rg)) * 2;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                     (__attribute__((address_space(3))) void*)(lds + sbase * 16), 16, 0, 0);
  }
}
template <bool KM, int ROWS>
__device__ __forceinline__ bf16x8_t frag_glds(const char* lds, int row0, int kk, int lane) {
  using G = GldsGeom<KM, ROWS>;
  if (KM) {
    const int row = row0 + (lane & 31);
    const int cc = kk * 2 + (lane >> 5);
    return *reinterpret_cast<const bf16x8_t*>(lds + row * 128 + ((cc ^ ((row >> 1) & 7)) * 16));
  } else {
    constexpr int PITCH = ROWS * 2;
    const int kbase = kk * 16 + (lane >> 5) * 8 + ((lane & 15) >> 2);
    const int r = row0 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    const int sw = G::SWZ_MN ? 2 * (kbase & 3) : 0;       // (kbase + 4) & 3 == kbase & 3: same swizzle for both reads
    const char* p0 = lds + kbase * PITCH + (((r >> 3) ^ sw) * 16) + (r & 7) * 2;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 4 * PITCH));
    s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
  }
}

// ---- LDS -> MFMA fragment (bf16) ---------------------------------------------------------------
// 32x32x16 operand fragment: lane l holds X[row = l&31][k = 8*(l>>5) .. +7] of the 16-deep k-step kk.
template <bool KM, int ROWS>
__device__ __forceinline__ bf16x8_t frag_bf16(const char* lds, int row0, int kk, int lane) {
  using G = TileGeom<DT_BF16, KM, ROWS>;
  if (KM) {
    const char* p = lds + (row0 + (lane & 31)) * G::PITCH + (kk * 16 + (lane >> 5) * 8) * 2;
    return *reinterpret_cast<const bf16x8_t*>(p);
  } else {
    // two transposed reads of a [4 k][16 rows] block per 16-lane group (ds_read_b64_tr_b16):
    // the lane supplies the address of 4 consecutive rows of ONE k-row, and receives 4 consecutive
    // k of ONE row: result[j] = LDS[k = kbase + j][row = rbase + (lane&15)].
    const int kbase = kk * 16 + (lane >> 5) * 8 + ((lane & 15) >> 2);
    const int r = row0 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    const char* p0 = lds + kbase * G::PITCH + r * 2;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t*)(p0));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t*)(p0 + 4 * G::PITCH));
    s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
  }
}

// Occupancy: 3 workgroups per CU (168 VGPRs) everywhere except the 128x128 tile with one K-major and one MN-major
// operand, whose two address streams + two prefetch sets + 64 accumulators need ~200 registers (spilled 120 at 168).
// STAGE: 0 = guarded generic staging, 1 = FAST register staging, 2 = LDS-DMA staging (glds)
template <int MODE, bool AK, bool BK, int WGM, int WGN, int TM, int TN, int STAGE>
__global__ __launch_bounds__(256, TM * TN >= 8 ? 1 : ((TM * TN >= 6 || (STAGE != 2 && TM * TN == 4 && AK != BK)) ? 2 : 3))
void gemm_kernel(const GemmK p) {
  constexpr bool FAST = STAGE == 1;
  constexpr bool GLDS = STAGE == 2;
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  using GA = TileGeom<MODE, AK, BM>;
  using GB = TileGeom<MODE, BK, BN>;
  constexpr int BKT = GA::BKT;
  constexpr int ES = GA::ES;
  constexpr int STG_BYTES = 4 * 32 * (TN * 32 + 4) * 4;      // wide-epilogue staging, 4 waves
  constexpr int A_BYTES = GLDS ? GldsGeom<AK, BM>::LDS_BYTES : GA::LDS_BYTES;
  constexpr int OPND_BYTES = A_BYTES + (GLDS ? GldsGeom<BK, BN>::LDS_BYTES : GB::LDS_BYTES);
  __shared__ __attribute__((aligned(16))) char smem[OPND_BYTES > STG_BYTES ? OPND_BYTES : STG_BYTES];
  char* ldsA = smem;
  char* ldsB = smem + A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  // XCD-aware tile order: hardware round-robins consecutive workgroup ids over the 8 XCDs; give each XCD a
  // contiguous run of tiles so operand panels shared by neighbouring tiles stay in ONE private L2 (bijective).
  int tm, tn, b, zs = blockIdx.z;
  if (p.xgm > 0) {
    // Batched product with ONE A for every batch (the remap: Wn . Y_b): the 8 XCDs split the BATCHES (workgroup id mod
    // 8 is the XCD), and an XCD walks its batches m-group by m-group, so the workgroups resident on it at any time are
    // xgm m-panels of A x (resident / xgm) batches of B -- the mix that minimises what its private L2 has to fetch
    // (in plain launch order every XCD streams ALL of A once per ~3 batches: 7-13x the algorithmic bytes, measured).
    const int L = blockIdx.x + gridDim.x * blockIdx.y;
    const int c = L & 7, sl = L >> 3;
    const int G = p.xgm * p.xnb;
    const int mg = sl / G, rem = sl - mg * G;
    int gs = p.tiles_m - mg * p.xgm; gs = gs < p.xgm ? gs : p.xgm;
    const int bb = rem / gs;
    tm = __builtin_amdgcn_readfirstlane(mg * p.xgm + (rem - bb * gs)); tn = 0;      // (the divisions run on the VALU:
    b = __builtin_amdgcn_readfirstlane(c * p.xnb + bb);                              //  back to scalar registers)
  } else {
    // ONE work list over (split, batch, tile), cut into 8 contiguous runs: hardware dispatches workgroups in linear
    // (x, y, z) order round-robin over the XCDs, so workgroup L lands on XCD L & 7 and is the (L >> 3)-th to start there.
    // What an XCD has resident at any time is then a window of ~64-96 neighbouring list entries:
    //  * all tiles of ONE split (or one batch) -- its operand slices cross the fabric once, not once per XCD (per-(y, z)
    //    remapping left 6 of a 48-tile split on every XCD: 178 MB fetched for 41 MB of operands at 512 x 384 x 23040);
    //  * with `nfast`, all n-tiles of an m-panel next to each other: a [23040 x 512] activation against a 512 x 512
    //    weight fetched its activation once per n-tile (96 MB for 24) when m ran fastest.
    const int ntile = p.tiles_m * p.tiles_n;
    if (p.nfast < 0) {               // DGSCT_GEMM_NFAST=-2: the round-1 order (per (y, z) plane, m fastest), kept for A/B runs
      int t = blockIdx.x;
      const int q = ntile >> 3, r = ntile & 7, x = t & 7, y = t >> 3;
      t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
      tm = t % p.tiles_m; tn = t / p.tiles_m; b = blockIdx.y;
    } else {
    const unsigned total = (unsigned)ntile * gridDim.y * gridDim.z;
    const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned q = total >> 3, r = total & 7, x = L & 7, y = L >> 3;
    const unsigned pp = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    const unsigned rest = pp / (unsigned)ntile;
    const int t = (int)(pp - rest * (unsigned)ntile);
    zs = __builtin_amdgcn_readfirstlane((int)(rest / gridDim.y));
    b = __builtin_amdgcn_readfirstlane((int)(rest - (unsigned)zs * gridDim.y));
    if (p.nfast) { tm = __builtin_amdgcn_readfirstlane(t / p.tiles_n); tn = __builtin_amdgcn_readfirstlane(t - tm * p.tiles_n); }
    else { tn = __builtin_amdgcn_readfirstlane(t / p.tiles_m); tm = __builtin_amdgcn_readfirstlane(t - tn * p.tiles_m); }
    }
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int kt_begin = zs * p.kt_per_split;
  int kt_end = kt_begin + p.kt_per_split;
  if (kt_end > p.kt_total) kt_end = p.kt_total;

  const char* Ab = p.A + (long)b * p.a_bs * ES;
  const char* Bb = p.B + (long)b * p.b_bs * ES;

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Two register sets in flight: the loads of k-tiles t+1 and t+2 are outstanding while tile t is in the MFMAs, so a
  // tile's global-memory latency is covered by two tiles of math (small GEMMs here are latency-, not throughput-bound).
  uint4 ra0[GA::NLD], rb0[GB::NLD], ra1[GA::NLD], rb1[GB::NLD];
  auto compute = [&]() {
    if (MODE == DT_BF16) {
#pragma unroll
      for (int kk = 0; kk < BKT / 16; ++kk) {
        bf16x8_t af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[i] = GLDS ? frag_glds<AK, BM>(ldsA, (wm * TM + i) * 32, kk, lane) : frag_bf16<AK, BM>(ldsA, (wm * TM + i) * 32, kk, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bf[j] = GLDS ? frag_glds<BK, BN>(ldsB, (wn * TN + j) * 32, kk, lane) : frag_bf16<BK, BN>(ldsB, (wn * TN + j) * 32, kk, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    } else {
      const float* la = reinterpret_cast<const float*>(ldsA);
      const float* lb = reinterpret_cast<const float*>(ldsB);
      constexpr int PA = GA::PITCH / 4, PB = GB::PITCH / 4;
#pragma unroll
      for (int kk = 0; kk < BKT / 2; ++kk) {
        float af[TM], bf[TN];
        const int k = kk * 2 + (lane >> 5);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = la[k * PA + (wm * TM + i) * 32 + (lane & 31)];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = lb[k * PB + (wn * TN + j) * 32 + (lane & 31)];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
  };
#define DGSCT_PREFETCH(RA, RB, KT)                                                                                     \
  do {                                                                                                                 \
    if constexpr (FAST) {                                                                                              \
      stage_load_fast<AK, BM>(RA, Ab, p.lda, p.a_kbs, m0, p.M, (KT) * BKT, p.K, p.kflat, p.kinv, tid);                   \
      stage_load_fast<BK, BN>(RB, Bb, p.ldb, p.b_kbs, n0, p.N, (KT) * BKT, p.K, p.kflat, p.kinv, tid);                   \
    } else {                                                                                                           \
      stage_load<MODE, AK, BM>(RA, Ab, p.lda, p.a_kbs, m0, p.M, (KT) * BKT, p.K, p.kflat, p.kinv, p.a_vec, tid);         \
      stage_load<MODE, BK, BN>(RB, Bb, p.ldb, p.b_kbs, n0, p.N, (KT) * BKT, p.K, p.kflat, p.kinv, p.b_vec, tid);         \
    }                                                                                                                  \
  } while (0)
#define DGSCT_STORE(RA, RB, KT)                                                                                        \
  do {                                                                                                                 \
    if constexpr (FAST) {                                                                                              \
      stage_store_fast<AK, BM>(RA, ldsA, m0, p.M, (KT) * BKT, p.kflat, tid);                                             \
      stage_store_fast<BK, BN>(RB, ldsB, n0, p.N, (KT) * BKT, p.kflat, tid);                                             \
    } else {                                                                                                           \
      stage_store<MODE, AK, BM>(RA, ldsA, tid);                                                                        \
      stage_store<MODE, BK, BN>(RB, ldsB, tid);                                                                        \
    }                                                                                                                  \
  } while (0)
  if constexpr (GLDS) {
    // one LDS buffer, two barriers per k-tile: the DMA of a workgroup is not overlapped with its own MFMAs -- the 3-4
    // resident workgroups of the CU overlap each other
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      __syncthreads();                     // previous tile's fragment reads are done
      glds_tile<AK, BM>(ldsA, Ab, p.lda, p.a_kbs, m0, p.M, kt * BKT, p.K, p.kinv, wave, lane);
      glds_tile<BK, BN>(ldsB, Bb, p.ldb, p.b_kbs, n0, p.N, kt * BKT, p.K, p.kinv, wave, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute();
    }
  } else {
  constexpr bool ONE_SET = TM * TN >= 6 && TM * TN < 8;   // 256 x 96: two sets (88 VGPRs) + 96 accumulators spill at 2 WG/CU
  if constexpr (ONE_SET) {
    if (kt_begin < kt_end) DGSCT_PREFETCH(ra0, rb0, kt_begin);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      __syncthreads();
      DGSCT_STORE(ra0, rb0, kt);
      __syncthreads();
      if (kt + 1 < kt_end) DGSCT_PREFETCH(ra0, rb0, kt + 1);
      compute();
    }
  } else {
  if (kt_begin < kt_end) DGSCT_PREFETCH(ra0, rb0, kt_begin);
  if (kt_begin + 1 < kt_end) DGSCT_PREFETCH(ra1, rb1, kt_begin + 1);
  for (int kt = kt_begin; kt < kt_end; kt += 2) {
    __syncthreads();                       // previous tile's fragment reads are done
    DGSCT_STORE(ra0, rb0, kt);
    __syncthreads();
    if (kt + 2 < kt_end) DGSCT_PREFETCH(ra0, rb0, kt + 2);
    compute();
    if (kt + 1 < kt_end) {
      __syncthreads();
      DGSCT_STORE(ra1, rb1, kt + 1);
      __syncthreads();
      if (kt + 3 < kt_end) DGSCT_PREFETCH(ra1, rb1, kt + 3);
      compute();
    }
  }
  }
  }
#undef DGSCT_PREFETCH
#undef DGSCT_STORE

  // ---- epilogue: accumulator element r of tile (i,j): row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31
  const float alpha = p.alpha * (p.alpha_ptr ? *p.alpha_ptr : 1.f);
  char* Db = p.D + (long)b * p.dbs * (p.ddt == DT_F32 ? 4 : 2);
  const char* Rb = p.R ? p.R + (long)b * p.rbs * (p.rdt == DT_F32 ? 4 : 2) : nullptr;
  const char* R2b = p.R2 ? p.R2 + (long)b * p.rbs * (p.rdt == DT_F32 ? 4 : 2) : nullptr;
  const char* Mb = p.mask ? p.mask + (long)b * p.maskbs * ES : nullptr;
  const float* bias_n = p.bias_n ? p.bias_n + (long)b * p.bias_n_bs : nullptr;
  if constexpr (WGM == 1 && TM == 1 && TN == 1) {
    // Column-wise softmax epilogues (ACT_SOFTMAX / ACT_SOFTMAX_BWD) over the M <= 32 rows of the tile, output written
    // TRANSPOSED: D[n][m].  The GEMM is issued as logits^T = tok . X^T, so the softmax axis (the latent tokens) runs
    // along the accumulator REGISTERS of a lane -- 16 values per lane, the other 16 in lane^32 -- and one token's
    // softmax costs ~35 VALU ops + 2 cross-half exchanges (a row-wise formulation needs 160 ds_bpermutes per tile).
    if (p.act == ACT_SOFTMAX || p.act == ACT_SOFTMAX_BWD) {
      const int n = n0 + wn * 32 + (lane & 31);
      const int h = lane >> 5;
      const bool nok = n < p.N;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = alpha * acc[0][0][r];
      const int esz = p.ddt == DT_F32 ? 4 : 2;
      char* drow = Db + (long)(nok ? n : 0) * p.ldd * esz;
      float dot_total = 0.f;
      if (p.act == ACT_SOFTMAX) {
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int t = (r & 3) + 8 * (r >> 2) + 4 * h;
          if (t < p.M) mx = fmaxf(mx, v[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int t = (r & 3) + 8 * (r >> 2) + 4 * h;
          v[r] = t < p.M ? __expf(v[r] - mx) : 0.f;
          sum += v[r];
        }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] *= inv;
      } else {
        const char* prow = Mb + (long)(nok ? n : 0) * p.ldmask * ES;
        float pv[16];
        float dot = 0.f;
        if (MODE == DT_BF16 && (p.ldmask & 3) == 0 && p.ldmask >= 32) {
          // the lane's 16 probabilities are 4 runs of 4 consecutive latent tokens: four 8-byte loads issued together
          // (the per-element lde_rt compiled to 16 load + s_waitcnt vmcnt(0) pairs, i.e. 16 serial round trips)
          uint2 w[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) w[q] = *reinterpret_cast<const uint2*>(prow + (8 * q + 4 * h) * 2);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int t0 = 8 * q + 4 * h;
            pv[4 * q] = t0 < p.M ? __uint_as_float(w[q].x << 16) : 0.f;
            pv[4 * q + 1] = t0 + 1 < p.M ? __uint_as_float(w[q].x & 0xffff0000u) : 0.f;
            pv[4 * q + 2] = t0 + 2 < p.M ? __uint_as_float(w[q].y << 16) : 0.f;
            pv[4 * q + 3] = t0 + 3 < p.M ? __uint_as_float(w[q].y & 0xffff0000u) : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) dot += pv[r] * v[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int t = (r & 3) + 8 * (r >> 2) + 4 * h;
            pv[r] = (t < p.M) ? lde_rt(prow, MODE, t) : 0.f;
            dot += pv[r] * v[r];
          }
        }
        dot += __shfl_xor(dot, 32, 64);
        const float sc = p.sm_scale ? *p.sm_scale : 1.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = sc * pv[r] * (v[r] - dot);
        if (nok && h == 0) dot_total = dot;
      }
      if (nok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {                    // 4 consecutive rows t = 8q + 4h + {0..3} per store
          const int t0 = 8 * q + 4 * h;
          if (t0 + 4 <= p.M && (p.ldd & 3) == 0) {
            if (p.ddt == DT_F32) {
              *reinterpret_cast<float4*>(drow + (long)t0 * 4) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            } else {
              uint2 w;
              w.x = f2bf2(v[4 * q], v[4 * q + 1]);
              w.y = f2bf2(v[4 * q + 2], v[4 * q + 3]);
              *reinterpret_cast<uint2*>(drow + (long)t0 * 2) = w;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (t0 + e < p.M) ste_rt(drow, p.ddt, t0 + e, v[4 * q + e]);
          }
        }
      }
      if (p.act == ACT_SOFTMAX_BWD && p.sm_dot) {        // one atomic per workgroup
        __syncthreads();
        const float tsum = block_sum(dot_total, reinterpret_cast<float*>(smem));
        if (tid == 0) unsafeAtomicAdd(p.sm_dot, tsum);
      }
      return;
    }
  }
  if (p.wide) {
    // Wide path: stage each wave's 32 x (TN*32) block through LDS (fp32) and write full token rows with 16-byte
    // stores (32-byte for fp32 out); the residual is read the same way.  An MFMA accumulator holds a COLUMN per lane,
    // so the direct path below can only issue 2-byte stores to 64-byte row segments -- 4-6x slower on the
    // token-major [rows][C] outputs that dominate this workload (they are HBM-bound GEMMs).
    constexpr int SP = TN * 32 + 4;                     // staging pitch (floats); 16-byte aligned rows
    constexpr int CPR = TN * 4;                         // 8-column chunks per row
    float* stg = reinterpret_cast<float*>(smem) + wave * (32 * SP);
    const int ncol0 = n0 + wn * TN * 32;
    // Loads of the epilogue are issued up front, unconditionally, from clamped indices: inside the per-element / per-chunk
    // conditionals below each one was a branch + load + s_waitcnt vmcnt(0) (16 serial round trips per tile for the rank-1
    // row factor of the remap bias, one per chunk for the residual).  Register budget: the residual prefetch only where the
    // variant has room (the 128 x 128 tiles at 3 workgroups per CU spilled with it).
    constexpr bool PRE_R_OK = TM * TN < 4 || (TM * TN == 4 && AK != BK && STAGE != 2);
    const bool pre_r = PRE_R_OK && Rb && !R2b && p.rdt == DT_BF16;   // residual rows as 16-byte chunks (the common case: dX1 += ...)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float r1m[16];
      if (p.r1_m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          m = m < p.M ? m : p.M - 1;
          r1m[r] = p.r1_m[p.m_mod > 0 ? m % p.m_mod : m];
        }
      }
      uint4 rpre[PRE_R_OK ? (32 * CPR) / 64 : 1];
      if (pre_r) {
#pragma unroll
        for (int it = 0; it < (PRE_R_OK ? (32 * CPR) / 64 : 1); ++it) {
          const int c = it * 64 + lane;
          int m = m0 + (wm * TM + i) * 32 + c / CPR, n = ncol0 + (c % CPR) * 8;
          m = m < p.M ? m : p.M - 1;
          n = n + 8 <= p.N ? n : (p.N >= 8 ? p.N - 8 : 0);
          rpre[it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(Rb) + (long)m * p.ldr + n);
        }
      }
      __syncthreads();                                  // operand tiles / previous block fully consumed
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = ncol0 + j * 32 + (lane & 31);
        const bool nok = n < p.N;
        const float bn = (bias_n && nok) ? bias_n[n] : 0.f;
        const float r1n = (p.r1_n && nok) ? p.r1_n[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          float v = alpha * acc[i][j][r] + bn;
          if (p.r1_m) v += r1m[r] * r1n;
          if (p.bias_m) {                               // (no caller in the adapter schedule: left as a guarded load)
            const int m = m0 + (wm * TM + i) * 32 + row;
            if (m < p.M) v += p.bias_m[p.m_mod > 0 ? m % p.m_mod : m];
          }
          if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
          else if (p.act == ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
          stg[row * SP + j * 32 + (lane & 31)] = v;
        }
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < (32 * CPR) / 64; ++it) {
        const int c = it * 64 + lane;
        const int row = c / CPR, cc = c % CPR;
        const int m = m0 + (wm * TM + i) * 32 + row;
        const int n = ncol0 + cc * 8;
        if (m >= p.M || n >= p.N) continue;
        float v[8];
        {
          const float4 a = *reinterpret_cast<const float4*>(stg + row * SP + cc * 8);
          const float4 c4 = *reinterpret_cast<const float4*>(stg + row * SP + cc * 8 + 4);
          v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c4.x; v[5] = c4.y; v[6] = c4.z; v[7] = c4.w;
        }
        const long od = (long)m * p.ldd + n;
        if (n + 8 <= p.N) {
          if (pre_r) {
            float rv[8];
            unpack<DT_BF16, 8>(rpre[PRE_R_OK ? it : 0], rv);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += p.beta * rv[e];
          } else if (Rb) {
            float rv[8];
            const long orr = (long)m * p.ldr + n;
            if (p.rdt == DT_F32) ldv<DT_F32, 4>(Rb, orr, *reinterpret_cast<float(*)[4]>(rv)), ldv<DT_F32, 4>(Rb, orr + 4, *reinterpret_cast<float(*)[4]>(rv + 4));
            else ldv<DT_BF16, 8>(Rb, orr, rv);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += p.beta * rv[e];
            if (R2b) {
              if (p.rdt == DT_F32) ldv<DT_F32, 4>(R2b, orr, *reinterpret_cast<float(*)[4]>(rv)), ldv<DT_F32, 4>(R2b, orr + 4, *reinterpret_cast<float(*)[4]>(rv + 4));
              else ldv<DT_BF16, 8>(R2b, orr, rv);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
          }
          if (p.ddt == DT_F32) { stv<DT_F32, 4>(Db, od, *reinterpret_cast<const float(*)[4]>(v)); stv<DT_F32, 4>(Db, od + 4, *reinterpret_cast<const float(*)[4]>(v + 4)); }
          else stv<DT_BF16, 8>(Db, od, v);
        } else {
          for (int e = 0; e < 8 && n + e < p.N; ++e) {
            float x = v[e];
            if (Rb) x += p.beta * lde_rt(Rb, p.rdt, (long)m * p.ldr + n + e);
            if (R2b) x += lde_rt(R2b, p.rdt, (long)m * p.ldr + n + e);
            ste_rt(Db, p.ddt, od + e, x);
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
    if (n >= p.N) continue;
    const float bn = bias_n ? bias_n[n] : 0.f;
    const float r1n = p.r1_n ? p.r1_n[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= p.M) continue;
        float v = alpha * acc[i][j][r] + bn;
        if (p.bias_m || p.r1_m) {
          const int mm = p.m_mod > 0 ? m % p.m_mod : m;
          if (p.bias_m) v += p.bias_m[mm];
          if (p.r1_m) v += p.r1_m[mm] * r1n;
        }
        if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
        else if (p.act == ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
        if (Mb) {
          const long o = (long)m * p.ldmask + n;
          const float mv = MODE == DT_BF16 ? bf2f(reinterpret_cast<const unsigned short*>(Mb)[o])
                                           : reinterpret_cast<const float*>(Mb)[o];
          if (!(mv > 0.f)) v = 0.f;
        }
        if (Rb) {
          const long o = (long)m * p.ldr + n;
          v += p.beta * (p.rdt == DT_F32 ? reinterpret_cast<const float*>(Rb)[o]
                                         : bf2f(reinterpret_cast<const unsigned short*>(Rb)[o]));
          if (R2b) v += lde_rt(R2b, p.rdt, o);
        }
        const long o = (long)m * p.ldd + n;
        if (p.atomic) unsafeAtomicAdd(reinterpret_cast<float*>(Db) + o, v);
        else if (p.ddt == DT_F32) reinterpret_cast<float*>(Db)[o] = v;
        else reinterpret_cast<unsigned short*>(Db)[o] = f2bf(v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Optional per-launch timing of the GEMM family (bench.py's roofline leg): HIP events recorded on the
// launch stream around every gemm_kernel launch of this thread while enabled.  Off by default.
struct ProfRec { hipEvent_t e0, e1; double flops; int M, N, K, KB, batch, splitk, cfg, ak, bk, atomic, wide; double bytes; };
// (process-wide: autograd runs backward on its own thread)
static std::atomic<bool> g_prof_on{false};
static std::mutex g_prof_mu;
static std::deque<ProfRec>* g_prof = nullptr;

static ProfRec* prof_begin(hipStream_t s, double flops, const ProfRec& shape) {
  if (!g_prof_on.load(std::memory_order_relaxed)) return nullptr;
  ProfRec r = shape;
  r.flops = flops;
  (void)hipEventCreate(&r.e0);
  (void)hipEventCreate(&r.e1);
  (void)hipEventRecord(r.e0, s);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof) g_prof = new std::deque<ProfRec>();
  g_prof->push_back(r);
  return &g_prof->back();          // deque: stable addresses
}
static void prof_end(ProfRec* r, hipStream_t s) {
  if (r) (void)hipEventRecord(r->e1, s);
}
void* gemm_prof_begin(void* stream, double flops, const GemmProfShape& h) {
  ProfRec shp{};
  shp.M = h.M; shp.N = h.N; shp.K = h.K; shp.KB = h.KB; shp.batch = h.batch; shp.splitk = h.splitk; shp.cfg = h.cfg;
  shp.ak = h.ak; shp.bk = h.bk; shp.atomic = h.atomic; shp.wide = h.wide; shp.bytes = h.bytes;
  return prof_begin((hipStream_t)stream, flops, shp);
}
void gemm_prof_end(void* rec, void* stream) { prof_end((ProfRec*)rec, (hipStream_t)stream); }
void gemm_prof_enable(int on) { g_prof_on.store(on != 0); }
// Synchronises on the recorded events; returns launches, sum of durations (ms) and of useful FLOPs; clears the log.
void gemm_prof_collect(long* launches, double* total_ms, double* total_flops) {
  long n = 0; double ms = 0, fl = 0;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  FILE* dump = nullptr;
  if (const char* path = getenv("DGSCT_PROF_DUMP")) dump = fopen(path, "w");
  if (dump) fprintf(dump, "M,N,K,KB,batch,splitk,cfg,ak,bk,atomic,wide,flops,bytes,ms\n");
  if (g_prof) {
    for (auto& r : *g_prof) {
      (void)hipEventSynchronize(r.e1);
      float t = 0.f;
      if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) { ms += t; fl += r.flops; ++n; }
      if (dump) fprintf(dump, "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%.0f,%.0f,%.5f\n", r.M, r.N, r.K, r.KB, r.batch, r.splitk, r.cfg,
                        r.ak, r.bk, r.atomic, r.wide, r.flops, r.bytes, t);
      (void)hipEventDestroy(r.e0);
      (void)hipEventDestroy(r.e1);
    }
    g_prof->clear();
  }
  if (dump) fclose(dump);
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
}

// ---- per-call time line (diagnostics): one event pair per adapter forward / backward call on ITS stream, so the overlap of the
// two adapter streams can be read without a tracer in the way (rocprofv3 --kernel-trace makes the late stages host-bound).
struct CallRec { hipEvent_t e0, e1; int kind, N, C; void* stream; };
static std::atomic<int> g_call_on{0};
static std::deque<CallRec>* g_calls = nullptr;
int call_prof_mode(int set) {
  const int old = g_call_on.load();
  if (set == 0 || set == 1) g_call_on.store(set);
  return old;
}
void* call_prof_begin(void* stream, int kind, int N, int C) {
  if (!g_call_on.load(std::memory_order_relaxed)) return nullptr;
  CallRec r{nullptr, nullptr, kind, N, C, stream};
  (void)hipEventCreate(&r.e0);
  (void)hipEventCreate(&r.e1);
  (void)hipEventRecord(r.e0, (hipStream_t)stream);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_calls) g_calls = new std::deque<CallRec>();
  g_calls->push_back(r);
  return &g_calls->back();
}
void call_prof_end(void* rec) {
  if (rec) (void)hipEventRecord(((CallRec*)rec)->e1, (hipStream_t)((CallRec*)rec)->stream);
}
// "kind N C stream t0_us t1_us" per call, times relative to the first call's start; clears the log
void call_prof_dump(const char* path) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  FILE* f = fopen(path, "w");
  if (g_calls && !g_calls->empty()) {
    hipEvent_t base = g_calls->front().e0;
    for (auto& r : *g_calls) {
      (void)hipEventSynchronize(r.e1);
      float a = 0.f, b = 0.f;
      (void)hipEventElapsedTime(&a, base, r.e0);
      (void)hipEventElapsedTime(&b, base, r.e1);
      if (f) fprintf(f, "%s %d %d %p %.1f %.1f\n", r.kind ? "bwd" : "fwd", r.N, r.C, r.stream, a * 1e3, b * 1e3);
    }
    for (auto& r : *g_calls) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_calls->clear();
  }
  if (f) fclose(f);
}

template <int MODE, int WGM, int WGN, int TM, int TN, int STAGE>
static void launch_lay(const GemmK& k, int ak, int bk, dim3 grid, hipStream_t s) {
  if (ak && bk)       hipLaunchKernelGGL((gemm_kernel<MODE, true, true, WGM, WGN, TM, TN, STAGE>), grid, dim3(256), 0, s, k);
  else if (ak && !bk) hipLaunchKernelGGL((gemm_kernel<MODE, true, false, WGM, WGN, TM, TN, STAGE>), grid, dim3(256), 0, s, k);
  else if (!ak && bk) hipLaunchKernelGGL((gemm_kernel<MODE, false, true, WGM, WGN, TM, TN, STAGE>), grid, dim3(256), 0, s, k);
  else                hipLaunchKernelGGL((gemm_kernel<MODE, false, false, WGM, WGN, TM, TN, STAGE>), grid, dim3(256), 0, s, k);
}
// stage: 0 generic, 1 FAST, 2 glds
template <int MODE, int WGM, int WGN, int TM, int TN>
static void launch_cfg(const GemmK& k, int ak, int bk, dim3 grid, hipStream_t s, int stage) {
  if constexpr (MODE == DT_BF16) {
    if (stage == 2) { launch_lay<MODE, WGM, WGN, TM, TN, 2>(k, ak, bk, grid, s); return; }
    if (stage == 1) { launch_lay<MODE, WGM, WGN, TM, TN, 1>(k, ak, bk, grid, s); return; }
  }
  launch_lay<MODE, WGM, WGN, TM, TN, 0>(k, ak, bk, grid, s);
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// what-if switch (TIMING ONLY, results wrong): split-K partial tiles leave as plain stores instead of fp32 atomics -- what the atomics cost
static std::atomic<int> g_noatomic{0}, g_cfgx{0};
int gemm_cfgx_mode(int set) { const int old = g_cfgx.load(); if (set >= 0) g_cfgx.store(set); return old; }
int gemm_noatomic_mode(int set) { const int old = g_noatomic.load(); if (set >= 0) g_noatomic.store(set); return old; }

template <int MODE>
static void gemm_mode(const Ctx& ctx, const Gemm& g) {
  constexpr int ES = MODE == DT_BF16 ? 2 : 4;
  constexpr int VE = 16 / ES;
  constexpr int BKT = MODE == DT_BF16 ? 64 : 16;
  if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return;
  GemmK k;
  k.M = g.M; k.N = g.N; k.K = g.K; k.KB = g.KB;
  k.A = (const char*)g.A.p; k.lda = g.A.ld; k.a_bs = g.A.bs; k.a_kbs = g.A.kbs;
  k.B = (const char*)g.B.p; k.ldb = g.B.ld; k.b_bs = g.B.bs; k.b_kbs = g.B.kbs;
  k.a_vec = aligned16(g.A.p) && g.A.ld % VE == 0 && g.A.bs % VE == 0 && g.A.kbs % VE == 0;
  k.b_vec = aligned16(g.B.p) && g.B.ld % VE == 0 && g.B.bs % VE == 0 && g.B.kbs % VE == 0;
  k.D = (char*)g.D; k.ddt = g.ddt; k.ldd = g.ldd; k.dbs = g.dbs;
  k.alpha = g.alpha; k.alpha_ptr = g.alpha_ptr;
  k.bias_m = g.bias_m; k.bias_n = g.bias_n; k.bias_n_bs = g.bias_n_bs; k.m_mod = g.m_mod;
  k.r1_m = g.r1_m; k.r1_n = g.r1_n; k.act = g.act;
  k.R = (const char*)g.R; k.rdt = g.rdt; k.ldr = g.ldr; k.rbs = g.rbs; k.beta = g.beta;
  k.R2 = g.R ? (const char*)g.R2 : nullptr;
  k.mask = (const char*)g.mask; k.ldmask = g.ldmask; k.maskbs = g.maskbs;
  k.sm_scale = g.sm_scale; k.sm_dot = g.sm_dot;
  k.atomic = g.atomic;
  if constexpr (MODE == DT_BF16) {
    if (gemm_skinny_try(ctx, g)) return;                        // [BT, C] gate-MLP products: K split over the waves (gemm_skinny.hip)
    if (gemm_tall_try(ctx, g)) return;                          // weight gradients over the token rows of stages 0-1 (gemm_tall.hip)
    if (gemm8_try(ctx, g)) return;                              // deep products: 8-wave LDS-DMA pipelined kernel (gemm8.hip)
  }
  const bool rowwise = g.act == ACT_SOFTMAX || g.act == ACT_SOFTMAX_BWD;
  {
    const int dv = g.ddt == DT_F32 ? 4 : 8, rv = g.rdt == DT_F32 ? 4 : 8;
    bool w = !g.atomic && !g.mask && !rowwise && aligned16(g.D) && g.ldd % dv == 0 && g.dbs % dv == 0 && g.N >= 8;
    if (g.R) w = w && aligned16(g.R) && g.ldr % rv == 0 && g.rbs % rv == 0 && (!g.R2 || aligned16(g.R2));
    k.wide = w;
  }
  k.kflat = g.K * g.KB;
  // frame index of a flat contraction index: floor(kf / K) = umulhi(kf, ceil(2^32 / K)), exact while kf * K < 2^32.
  // One-level contractions (KB == 1) can be far deeper than that (a weight gradient over 655 360 token rows): they get
  // kinv = 0, i.e. frame 0 for every kf, which is also what the branch-free fast paths rely on.
  k.kinv = (g.KB > 1 && g.K > 1) ? (unsigned)((0x100000000ULL + (unsigned long long)g.K - 1) / (unsigned long long)g.K) : 0u;
  if (g.KB > 1 && (unsigned long long)g.K * g.KB * g.K >= 0x100000000ULL) { set_error("gemm: two-level contraction too deep for the 32-bit frame split"); return; }
  k.kt_total = (k.kflat + BKT - 1) / BKT;

  // tile configuration (measured on MI355X with tools/gemm_bench.py over the shapes of the adapter stack):
  //  * deep contractions (K >= 1024: the remap GEMMs) want the big tiles: 128x96 when N is a multiple of 96 (the
  //    remap widths), else 128x128;
  //  * shallow ones (most of this workload: K = 32..768) are bound by per-workgroup latency and by wave
  //    quantisation over the 256 CUs x 3 resident 128x128 workgroups: 64x64 tiles (6 resident) win unless the 128x128
  //    grid fills its last round well and K is not tiny.
  int cfg, force_split = 0;
  const long kflat = (long)g.K * g.KB;
  auto tiles = [&](int bm, int bn) { return (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * g.batch; };
  if (g.N <= 32) cfg = 2;                                       // 128 x 32
  else if (g.M <= 32) cfg = 3;                                  // 32 x 128
  else if (kflat <= 64 && g.M >= 1024 && g.N <= 128 && g.N % 128 != 0) cfg = 2;   // token x tk products: N = 96 as 3 x 32
  else if (!g.atomic && g.N % 96 == 0 && g.N % 128 != 0 && g.M >= 128 && kflat >= 64 && tiles(128, 96) >= 256)
    cfg = (g.A.kmajor && g.M % 256 == 0 && kflat >= 1024) ? 5 : 1;   // 128 x 96: no padded columns; 256 x 96 (wave tile 64 x 96:
                                                                     // 0.85 instead of 1.37 KB of LDS reads per MFMA) for the deep ones
  else if (kflat >= 1024 && !g.atomic) {
    if (tiles(128, 128) < 64) cfg = 4;                          // B x C gate GEMMs (M = 160): 8-16 big tiles leave the chip idle
    else if (g.M >= 96 && g.N >= 96) cfg = 0;                   // 128 x 128
    else cfg = 4;
  } else if (g.atomic && g.batch == 1 && g.M > 64 && g.M <= 128 && g.N > 64 && g.N <= 128 && kflat >= 131072) {
    // stage-0 weight gradients (C x C over 370-650 k token rows): ONE tile spanning the whole output, split-K over one
    // workgroup per CU.  Four 64 x 64 tiles read every operand panel twice (425 MB fetched for 189 MB of operands) and
    // 32-row tiles three times: 98.6 -> 67.2 us at 96 x 96 x 655 360, 67.5 -> 60.8 at 128 x 128 x 368 640
    // (tools/gemm_wgrad_probe.py).
    cfg = g.N <= 96 ? 1 : 0;
    static const int fs = getenv("DGSCT_GEMM_FORCESPLIT") ? atoi(getenv("DGSCT_GEMM_FORCESPLIT")) : 256;
    force_split = fs;
  } else if (g.atomic && g.M >= 1024 && g.N >= 1024 && kflat >= 8192) cfg = 0;   // dWn: a plain big GEMM
  else if (g.atomic && g.M <= 128 && g.M % 64 != 0 && g.M % 32 == 0) cfg = 3;     // 96-row weight gradients: 3 x 32 rows
  else {
    const long w0 = tiles(128, 128);
    const long last = w0 % 768;
    const bool fills = w0 >= 192 && (last == 0 || last >= 576 || w0 >= 6144);
    cfg = (kflat > 256 && g.M >= 128 && g.N >= 128 && fills && !g.atomic) ? 0 : 4;
  }
  // experiment ("cfgx" bit 1): the per-frame batched products of the late stages with a short contraction (dY[b] = Wn^T dT[b],
  // T1[b] = Wn Y[b]: K = 36 .. 256) on 128 x 128 tiles instead of 64 x 64
  if ((g_cfgx.load(std::memory_order_relaxed) & 1) && !g.atomic && g.batch >= 16 && g.M >= 128 && g.N >= 128 && kflat <= 256 && cfg == 4) cfg = 0;
  if (const char* e = getenv("DGSCT_GEMM_CFG")) { const int c = atoi(e); if (c >= 0 && c <= 5) cfg = c; }   // tuning hook
  if (rowwise) {
    if (g.M > 32 || g.R || g.atomic || g.splitk > 1 || g.bias_m || g.bias_n) {
      set_error("gemm: softmax epilogue needs M <= 32 and a plain, unsplit output");
      return;
    }
    cfg = 3;                                                      // 32 x 128: the softmax axis in one tile's registers
  }
  const int ak = g.A.kmajor, bk = g.B.kmajor;
  // FAST staging: 16-byte chunks are aligned and never straddle a matrix edge or a frame of a two-level contraction
  static const bool no_fast = getenv("DGSCT_GEMM_NOFAST") != nullptr;
  const bool fast_ok = MODE == DT_BF16 && !no_fast && k.a_vec && k.b_vec && g.K % VE == 0 && (long)g.K * g.KB >= VE &&
                    (ak || g.M % VE == 0) && (bk || g.N % VE == 0) && g.M >= VE && g.N >= VE;
  // LDS-DMA staging (DGSCT_GEMM_GLDS=1: every eligible GEMM, =2: contractions >= 512 deep).  Measured on MI355X: -12 % on
  // the two-level / MN-major remap GEMMs in isolation (tools/gemm_bench.py), 688 vs 730 TFLOP/s at 4096^3, and no
  // difference on the whole step (77.1 / 77.6 / 76.8 ms off / deep-only / all) -- with 3 workgroups per CU the
  // two-deep register prefetch already hides the staging, so it stays off by default.
  static const int glds_mode = getenv("DGSCT_GEMM_GLDS") ? atoi(getenv("DGSCT_GEMM_GLDS")) : 0;
  const bool glds_ok = fast_ok && kflat % BKT == 0 && (glds_mode == 1 || (glds_mode == 2 && kflat >= 512));
  const int fast = glds_ok ? 2 : (fast_ok ? 1 : 0);
  if (cfg >= 5 && !(MODE == DT_BF16 && fast_ok)) cfg = 1;        // big tiles exist for FAST bf16 only
  static const int BMs[6] = {128, 128, 128, 32, 64, 256}, BNs[6] = {128, 96, 32, 128, 64, 96};
  k.tiles_m = (g.M + BMs[cfg] - 1) / BMs[cfg];
  k.tiles_n = (g.N + BNs[cfg] - 1) / BNs[cfg];
  int splitk = g.splitk;
  if (g.atomic && splitk <= 0) { if (const char* e = getenv("DGSCT_GEMM_SPLITK")) splitk = atoi(e); }   // tuning hook
  if (!g.atomic) splitk = 1;
  else if (splitk <= 0) {                                       // auto: aim for >= 1024 workgroups, >= 4 k-tiles each
    long wg = (long)k.tiles_m * k.tiles_n * g.batch;
    splitk = force_split ? force_split : (int)((1024 + wg - 1) / wg);
    // Every split ends in one fp32 atomic per output element, and atomics on one address serialise at the memory side
    // (~0.1 us each): 480 splits of a 36 x 64 weight gradient spent 50 of 58 us there.  Step time vs cap (B=16): none 67.5,
    // 64: 67.0, 40: 66.6, 32: 66.7, 24: 67.7, 16: 72.2 ms.  DGSCT_GEMM_MAXSPLIT=0 removes the cap.
    static const int max_split = getenv("DGSCT_GEMM_MAXSPLIT") ? atoi(getenv("DGSCT_GEMM_MAXSPLIT")) : 40;
    {   // ... but a workgroup should not walk more than ~32 k-tiles either (655 360-row contractions: 40 splits = 256 k-tiles
        // each, 414 us for a 48 x 6 weight gradient that took 126 us at 512 splits)
      const int deep = k.kt_total / 32;
      const int cap = max_split > deep ? max_split : deep;
      if (max_split > 0 && !force_split && splitk > cap) splitk = cap;
    }
    int maxs = k.kt_total / 4; if (maxs < 1) maxs = 1;
    if (splitk > maxs) splitk = maxs;
    if (splitk < 1) splitk = 1;
  }
  k.kt_per_split = (k.kt_total + splitk - 1) / splitk;
  splitk = (k.kt_total + k.kt_per_split - 1) / k.kt_per_split;
  dim3 grid(k.tiles_m * k.tiles_n, g.batch, splitk);
  k.xgm = k.xnb = 0;
  // n fastest when there are fewer n- than m-tiles: the window of tiles resident on an XCD then spans whole rows of tiles
  // (each big-operand panel is fetched once; the small operand once per XCD).  DGSCT_GEMM_NFAST=0|1 forces it.
  static const int nfast_env = getenv("DGSCT_GEMM_NFAST") ? atoi(getenv("DGSCT_GEMM_NFAST")) : -1;
  k.nfast = nfast_env >= 0 ? nfast_env : nfast_env == -2 ? -1 : (k.tiles_n < k.tiles_m);
  static const bool no_xgroup = getenv("DGSCT_GEMM_NOXGROUP") != nullptr;
  if (!no_xgroup && g.A.bs == 0 && g.B.bs != 0 && g.batch >= 16 && g.batch % 8 == 0 && k.tiles_n == 1 && k.tiles_m >= 2 && splitk == 1 &&
      kflat >= 512) {
    const int resident = (cfg == 5 ? 1 : (cfg == 0 && ak != bk) ? 2 : 3) * 32;          // workgroups per XCD (launch bounds)
    int gm = (int)(sqrt((double)resident * g.N / BMs[cfg]) + 0.5);
    if (gm < 1) gm = 1;
    if (gm > k.tiles_m) gm = k.tiles_m;
    const int ngroups = (k.tiles_m + gm - 1) / gm;
    k.xgm = (k.tiles_m + ngroups - 1) / ngroups;
    k.xnb = g.batch / 8;
  }
  if (g_noatomic.load(std::memory_order_relaxed)) k.atomic = 0;
  hipStream_t s = (hipStream_t)ctx.stream;
  ProfRec shp{};
  shp.M = g.M; shp.N = g.N; shp.K = g.K; shp.KB = g.KB; shp.batch = g.batch; shp.splitk = splitk; shp.cfg = cfg;
  shp.ak = ak; shp.bk = bk; shp.atomic = g.atomic; shp.wide = k.wide;
  shp.bytes = ((double)g.M * g.K * g.KB * (g.A.bs ? g.batch : 1) + (double)g.N * g.K * g.KB * (g.B.bs ? g.batch : 1)) * ES +
              (double)g.M * g.N * g.batch * (g.ddt == DT_F32 ? 4 : 2) * (g.R ? 2 : 1);
  ProfRec* rec = prof_begin(s, 2.0 * g.M * g.N * (double)g.K * g.KB * g.batch, shp);
  switch (cfg) {
    case 0: launch_cfg<MODE, 2, 2, 2, 2>(k, ak, bk, grid, s, fast); break;
    case 1: launch_cfg<MODE, 4, 1, 1, 3>(k, ak, bk, grid, s, fast); break;
    case 2: launch_cfg<MODE, 4, 1, 1, 1>(k, ak, bk, grid, s, fast); break;
    case 3: launch_cfg<MODE, 1, 4, 1, 1>(k, ak, bk, grid, s, fast); break;
    case 5: if constexpr (MODE == DT_BF16) {                                                                 // 256 x 96
        if (fast == 2) launch_lay<DT_BF16, 4, 1, 2, 3, 2>(k, ak, bk, grid, s);
        else launch_lay<DT_BF16, 4, 1, 2, 3, 1>(k, ak, bk, grid, s);
      } break;
    default: launch_cfg<MODE, 2, 2, 1, 1>(k, ak, bk, grid, s, fast); break;
  }
  prof_end(rec, s);
}

void gemm(const Ctx& ctx, const Gemm& g) {
  if (ctx.mode == DT_BF16) gemm_mode<DT_BF16>(ctx, g);
  else gemm_mode<DT_F32>(ctx, g);
}

}  // namespace dgsct
