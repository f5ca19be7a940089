// 8-wave LDS-DMA pipelined MFMA GEMM for the DEEP products of the adapter path (gfx950, bf16).
//
//   D[m][n] (+)= sum_k A[m][k] * B[n][k]        M % 256 == 0, contraction a whole number of 64-deep k-tiles
//
// What it is for (the six remap products of stage 0 are 53 % of the step's FLOPs):
//   * the batched remap products with ONE A for every frame -- Wn . Y[b], Wn^T . dT[b] -- whose per-frame width is only
//     Co or C (96..256 columns).  The frames are laid side by side: column n of the product is (frame n / Nsub,
//     column n % Nsub), so a 256 x 192 tile spans two 96-wide frames and the 160 frames give 80 column tiles;
//   * the remap weight gradient dWn = sum_b dT[b] . Y[b]^T (two-level contraction over (frame, channel), split-K);
//   * the C x C weight gradients of the late stages (contraction over 23 040 .. 40 960 token rows, split-K).
//
// Structure (cdna_hip_programming.md section 5, "glds vs register staging" / the 256^2 template, simplified to two phases):
//   * 512 threads = 8 wavefronts as 4 (M) x 2 (N); a wave owns 64 x BN/2 of the 256 x BN tile: 2 x 3 or 2 x 4
//     v_mfma_f32_32x32x16_bf16 accumulators (96 / 128 VGPRs);
//   * operands go global -> LDS by global_load_lds_dwordx4 (no staging registers, no ds_write pass) into TWO LDS
//     buffers; the k-tile t+2 is requested as soon as every wave has finished reading tile t, so a tile's DMA has the
//     whole MFMA phase of the tile in between to land: `s_waitcnt vmcnt(N)` with N = this wave's DMA instructions per
//     k-tile (never 0 inside the loop), raw s_barrier (a __syncthreads would drain the DMA queue);
//   * dense LDS images, conflict-free fragment reads by swizzling the SOURCE address of each 16-byte chunk:
//     K-major tile [rows][64 k]: chunk c of row r at slot c ^ ((r >> 1) & 7), read with ds_read_b128;
//     MN-major tile [64 k][rows]: chunk rc of k-row k at slot rc ^ swz_mn(k), read with ds_read_b64_tr_b16;
//   * epilogue: bf16 / fp32 rows through LDS as 16-byte stores (+ rank-1 and column bias of the remap), or fp32 atomics
//     for split-K.
// One workgroup per CU (112 / 128 KB of LDS): grids are sized in whole rounds of 256 workgroups where the shape allows.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "prims.h"
#include "device_util.h"
#include "err.h"
#include "gemm_int.h"

namespace dgsct {

typedef __attribute__((ext_vector_type(8))) __bf16 g8_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short g8_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short g8_s16x8_t;
typedef __attribute__((ext_vector_type(16))) float g8_f32x16_t;

// Structure experiments of round 4 (compile-time, default = the round-3 kernel: 8 waves, 256-row tiles).  Timings of the remap
// forward 2304 x (96 x 160) x 4096, K-major x MN-major, alone on the chip (tools/gemm_bench.py BIG=1, DGSCT_GEMM8_DBG switches):
//                                             whole kernel   no DMA in the loop   DMA + barriers only   MFMAs + barriers only
//   8 waves, 256 x 192 (64 x 96 per wave)        331 us           268 us               154 us                 196 us
//   4 waves, 256 x 192 (128 x 96 per wave,
//     accumulators in AGPRs, 0.58 KB of
//     fragment reads per MFMA instead of 0.83)   347              274                  174
//   4 waves, 128 x 192, TWO workgroups per CU
//     (80 KB of LDS each, independent barriers)  356              267                  184
// i.e. neither fewer fragment reads nor de-synchronised workgroups move it, and the operand DMA alone is half the kernel: the
// round-3 "per-CU delivery ceiling" reading was wrong (DMA only = 154 us), and so is "barrier lock-step".  What the three share
// is the MFMA stream itself; with operands that never change (the no-read runs) it finishes in 196 us, with real data the
// clock drops (MI355X_MICROARCH.md: 2.29 -> 1.87 GHz under bf16 MFMA load on random operands), so part of the 268 is DVFS.
#ifndef G8_WAVES
#define G8_WAVES 8
#endif
constexpr int G8_NW = G8_WAVES;
constexpr int G8_NT = G8_NW * 64;
// rows of the workgroup tile.  128 (with 4 waves of 64 x BN/2): 2 x (128 + 192) x 128 B = 80 KB of LDS per workgroup, so TWO
// independent workgroups share a CU and fill each other's barrier / LDS-latency bubbles (one 8-wave workgroup runs its two waves
// per SIMD in lock-step through the same barriers); the price is 1.43 x the operand DMA per FLOP.
#ifndef G8_BM
#define G8_BM 256
#endif
constexpr int G8_ROWS = G8_BM;

struct G8 {
  int M, Ntot, Nsub; unsigned ninv;       // column n -> (frame n / Nsub, n % Nsub); ninv = ceil(2^32 / Nsub)
  int K, kflat; unsigned kinv;            // flat contraction index kf -> (kf / K, kf % K); kinv = ceil(2^32 / K), 0: one level
  int tiles_m, tiles_n, kt_total, kt_per_split;
  const char* A; long lda, a_kbs;
  const char* B; long ldb, b_bs, b_kbs;
  char* D; int ddt; long ldd, dbs; int atomic;
  const float* r1_m; const float* r1_n; const float* bias_n;
  const char* R;                          // bf16 residual laid out like D (ldd, dbs), added in the row pass (plain stores only); NULL: none
  int gm;                                 // m-tiles per group of the work list (tile order, see the kernel)
  int stag;                               // staggered DMA issue of the two wave halves (PIPE loop; dgsct_test_tune "g8stag")
  int dbg;                                // DGSCT_GEMM8_DBG (timing experiments only, results are garbage): 1 no DMA in the loop, 2 no fragment reads, 4 no MFMA
};

template <bool KM, int ROWS>
struct G8Geom {
  static constexpr int CPR = KM ? 8 : ROWS / 8;          // 16-byte chunks per LDS row
  static constexpr int NSLOT = ROWS * 8;
  static constexpr int NI = NSLOT / G8_NT;               // DMA instructions per thread and k-tile
  static constexpr int BYTES = NSLOT * 16;
  static_assert(NSLOT % G8_NT == 0, "tile must be a whole number of workgroup-wide DMA rounds");
  static_assert(KM || ROWS == 128 || ROWS == 192 || ROWS == 256, "MN-major swizzle is worked out for 128-, 192- and 256-row tiles");
  // MN-major image: a 32-lane group of ds_read_b64_tr_b16 reads 4 consecutive k-rows x 64 contiguous bytes (2 x 16 rows); the four
  // 64-byte pieces must fall on different quarters of the 256-byte bank line.  k-rows are ROWS * 2 bytes apart:
  //   512 B (256 rows): all four on the same quarter -> chunk index ^ 4 (k & 3)          (quarters 0, 1, 2, 3)
  //   384 B (192 rows): quarters 0, 2, 0, 2          -> chunk index ^ 4 ((k >> 1) & 1)   (quarters 0, 2, 1, 3)
  // (the first version XOR-ed 2 (k & 3): that only swaps the two 32-byte halves of a piece -- 28-40 % SQ_LDS_BANK_CONFLICT)
  static __host__ __device__ constexpr int swz_mn(int k) { return (ROWS == 256 || ROWS == 128) ? 4 * (k & 3) : 4 * ((k >> 1) & 1); }
};

// one operand tile of one k-tile: global -> LDS.  `BATCHED`: rows are (frame, row-in-frame) pairs, frames bs apart.
template <bool KM, int ROWS, bool BATCHED, bool TWO>
__device__ __forceinline__ void g8_glds(char* lds, const char* base, long ld, long kbs, long bs, int nsub, unsigned ninv, int r0,
                                        int kf0, int K, unsigned kinv, int wave, int lane) {
  using G = G8Geom<KM, ROWS>;
#pragma unroll
  for (int j = 0; j < G::NI; ++j) {
    const int sbase = (j * G8_NW + wave) * 64;               // wave-uniform slot base: 64 lanes x 16 B = 1 KiB of LDS
    const int s = sbase + lane;
    const int q = s / G::CPR, cpos = s % G::CPR;
    int rg, kf;
    if (KM) {
      const int c = cpos ^ ((q >> 1) & 7);
      rg = r0 + q;
      kf = kf0 + c * 8;
    } else {
      const int rc = cpos ^ G::swz_mn(q);
      rg = r0 + rc * 8;
      kf = kf0 + q;
    }
    long off = 0;
    int kk = kf;
    if (TWO) {
      const int kb = (int)__umulhi((unsigned)kf, kinv);
      kk = kf - kb * K;
      off = (long)kb * kbs;
    }
    int rr = rg;
    if (BATCHED) {
      const int bb = (int)__umulhi((unsigned)rg, ninv);
      rr = rg - bb * nsub;
      off += (long)bb * bs;
    }
    off += KM ? (long)rr * ld + kk : (long)kk * ld + rr;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off * 2),
                                     (__attribute__((address_space(3))) void*)(lds + sbase * 16), 16, 0, 0);
  }
}

// ---- fragment reads ---------------------------------------------------------------------------------------------------
// 32x32x16 operand fragment of the 16-deep k-step kk: lane l holds X[row0 + (l & 31)][8 (l >> 5) .. + 7].
// K-major image: one ds_read_b128 at row * 128 + ((2 kk + h) ^ sw) * 16, sw = (row >> 1) & 7 (the same for every 32-row block);
// MN-major image: two ds_read_b64_tr_b16 (4 k-rows x 16 rows each) at kbase * PITCH + ((r >> 3) ^ sw) * 16 + (r & 7) * 2.
//
// hipcc (ROCm 7.2) puts an `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 INTRINSIC while an LDS-DMA is in flight (the
// intrinsic carries no memory operand, so the wait-count pass assumes it may read what the DMA writes) -- that drains the
// prefetched k-tile at the top of every iteration and the pipeline is gone.  Plain ds_read_b128 loads are not affected.  Variants
// with an MN-major operand therefore issue ALL their fragment reads as inline asm and order them by hand: the asm outputs only
// become visible to the MFMAs through `g8_tie` after an explicit `s_waitcnt lgkmcnt(0)`.
template <bool KM, int ROWS>
struct G8FragAddr {
  // per-lane byte offset of the fragment of 32-row block `blk` (k-step 0, buffer 0) inside the operand image
  static __device__ __forceinline__ unsigned base(int row0, int lane) {
    if (KM) return (unsigned)((row0 + (lane & 31)) * 128);
    constexpr int PITCH = ROWS * 2;
    const int kb0 = (lane >> 5) * 8 + ((lane & 15) >> 2);
    const int r = row0 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    const int sw = G8Geom<KM, ROWS>::swz_mn(kb0);          // (kb0 + 4, + 8, + 16 .. have the same k & 3)
    return (unsigned)(kb0 * PITCH + (((r >> 3) ^ sw) * 16) + (r & 7) * 2);
  }
};
// A fragment in flight.  K-major: ONE 128-bit asm output (never split before the wait: a sub-register copy in between would read
// registers the LDS has not written yet); MN-major: the two 8-byte halves the transpose reads return, joined after the wait.
// tools/check_gemm8_isa.py (run by the CPU tests) verifies on the generated ISA that nothing touches these registers between
// the read and the `s_waitcnt lgkmcnt(0)` that covers it, that the loop has no scratch traffic and no compiler-made vmcnt wait.
template <bool KM> struct G8F;
template <> struct G8F<true> { g8_s16x8_t v; };
template <> struct G8F<false> { g8_s16x4_t lo, hi; };

template <int OFF>
__device__ __forceinline__ void g8_rd128(unsigned a, G8F<true>& f) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.v) : "v"(a), "n"(OFF));
}
template <int OFF0, int OFF1>
__device__ __forceinline__ void g8_rdtr(unsigned a, G8F<false>& f) {
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(f.lo), "=&v"(f.hi) : "v"(a), "n"(OFF0), "n"(OFF1));
}
__device__ __forceinline__ void g8_tie(G8F<true>& f) { asm volatile("" : "+v"(f.v)); }
__device__ __forceinline__ void g8_tie(G8F<false>& f) { asm volatile("" : "+v"(f.lo), "+v"(f.hi)); }
__device__ __forceinline__ g8_bf16x8_t g8_val(const G8F<true>& f) { return __builtin_bit_cast(g8_bf16x8_t, f.v); }
__device__ __forceinline__ g8_bf16x8_t g8_val(const G8F<false>& f) {
  g8_s16x8_t v = __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(g8_bf16x8_t, v);
}
// compiler-scheduled K-major fragment (both operands K-major: no transpose reads, hipcc's own lgkmcnt placement is good)
template <int ROWS>
__device__ __forceinline__ g8_bf16x8_t g8_frag_km(const char* lds, int row0, int kk, int lane) {
  const int row = row0 + (lane & 31);
  const int cc = kk * 2 + (lane >> 5);
  return *reinterpret_cast<const g8_bf16x8_t*>(lds + row * 128 + ((cc ^ ((row >> 1) & 7)) * 16));
}

template <int N> __device__ __forceinline__ void g8_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PIPE (round 5): the k-loop software-pipelined ACROSS k-tiles -- ONE barrier and no exposed LDS latency per k-tile (see the loop)
template <bool AK, bool BK, int BN, bool BATCHED, bool TWO, bool PIPE>
__global__ __launch_bounds__(G8_NT, (G8_NW == 8 || G8_ROWS == 128) ? 2 : 1)
void gemm8_kernel(const G8 p) {
  constexpr int BM = G8_ROWS, WGN = 2, WGM = G8_NW / WGN, TM = BM / (WGM * 32), TN = BN / 64;
  using GA = G8Geom<AK, BM>;
  using GB = G8Geom<BK, BN>;
  constexpr int STAGE = GA::BYTES + GB::BYTES;
  constexpr int SP = TN * 32 + 4;                           // epilogue staging pitch (floats)
  constexpr int STG_BYTES = G8_NW * 32 * SP * 4;
  constexpr int SMEM = 2 * STAGE > STG_BYTES ? 2 * STAGE : STG_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[SMEM];   // ONE LDS object (a second one de-pipelines the DMA waits)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  // work list (split, m-group, n-tile, m-tile in group), cut into 8 contiguous runs: workgroup L runs on XCD L & 7 as its
  // (L >> 3)-th arrival, so the ~32 workgroups resident on an XCD are neighbouring list entries: `gm` m-panels of A against
  // 32 / gm column tiles of one split.  Per k-tile they fetch gm A-chunks + 32/gm B-chunks into that XCD's private L2 for 32
  // workgroups' worth of demand (gm = 4: 8 x less than it serves).
  const unsigned ntile = (unsigned)p.tiles_m * p.tiles_n;
  const unsigned total = ntile * gridDim.y;
  const unsigned L = blockIdx.x + gridDim.x * blockIdx.y;
  const unsigned qd = total >> 3, rd = total & 7, x = L & 7, y = L >> 3;
  const unsigned pp = (x < rd ? x * (qd + 1) : rd * (qd + 1) + (x - rd) * qd) + y;
  const int zs = __builtin_amdgcn_readfirstlane((int)(pp / ntile));
  const int t = (int)(pp - (unsigned)zs * ntile);
  const int gsz = p.gm * p.tiles_n;                          // entries of a full m-group
  const int grp = t / gsz, rem = t - grp * gsz;
  int gw = p.tiles_m - grp * p.gm; gw = gw < p.gm ? gw : p.gm;   // the last group may be narrower
  const int tn = __builtin_amdgcn_readfirstlane(rem / gw);
  const int tm = __builtin_amdgcn_readfirstlane(grp * p.gm + (rem - tn * gw));
  const int m0 = tm * BM, n0 = tn * BN;
  const int kt_begin = zs * p.kt_per_split;
  int kt_end = kt_begin + p.kt_per_split;
  if (kt_end > p.kt_total) kt_end = p.kt_total;
  const int nk = kt_end - kt_begin;

  g8_f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto issue = [&](int kt, int buf) {
    char* la = smem + buf * STAGE;
    g8_glds<AK, BM, false, TWO>(la, p.A, p.lda, p.a_kbs, 0, 0, 0u, m0, kt * 64, p.K, p.kinv, wave, lane);
    g8_glds<BK, BN, BATCHED, TWO>(la + GA::BYTES, p.B, p.ldb, p.b_kbs, p.b_bs, p.Nsub, p.ninv, n0, kt * 64, p.K, p.kinv, wave, lane);
  };
  constexpr int NDMA = GA::NI + GB::NI;                      // DMA instructions per wave and k-tile

  constexpr bool ASMRD = !(AK && BK);                        // any transpose-read operand: hand-ordered fragment reads
  const unsigned smem_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
  // per-lane fragment addresses (buffer 0, k-step 0) of this wave's row blocks; K-major images add a k-step term kx[kk]
  unsigned baseA[TM], baseB[TN], kx[4];
#pragma unroll
  for (int i = 0; i < TM; ++i) baseA[i] = smem_lds + G8FragAddr<AK, BM>::base((wm * TM + i) * 32, lane);
#pragma unroll
  for (int j = 0; j < TN; ++j) baseB[j] = smem_lds + GA::BYTES + G8FragAddr<BK, BN>::base((wn * TN + j) * 32, lane);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) kx[kk] = (unsigned)((((kk * 2 + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) * 16));

  if constexpr (PIPE) {
    // One k-tile = four 16-deep k-steps; fragment registers alternate between two sets.  Steady state of iteration `it` (tile `it` in
    // buffer it & 1, its k-step 0 fragments already in set 0):
    //     rd(1) | mma(0) | rd(2) | mma(1) | rd(3) | mma(2) | -- every read of tile `it` has returned --
    //     s_waitcnt vmcnt(0) (this wave's pieces of tile it + 1: the only DMA group in flight) ; s_barrier
    //         => tile it + 1 is visible to every wave AND every wave is done reading tile `it`: one barrier serves both hazards
    //     DMA of tile it + 2 into buffer it & 1 ; rd(0) of tile it + 1 | mma(3)
    // i.e. one barrier per k-tile instead of two, and the first fragment read of a tile flies under the last MFMA group of the tile
    // before it instead of sitting exposed behind the barrier.
    G8F<AK> fa[2][TM];
    G8F<BK> fb[2][TN];
    auto rdp = [&](auto kkc, unsigned boff) {
      constexpr int KK = decltype(kkc)::value;
      constexpr int S = KK & 1;
      constexpr int PA = BM * 2, PB = BN * 2;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (AK) g8_rd128<0>(baseA[i] + boff + kx[KK], fa[S][i]);
        else g8_rdtr<KK * 16 * PA, KK * 16 * PA + 4 * PA>(baseA[i] + boff, fa[S][i]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (BK) g8_rd128<0>(baseB[j] + boff + kx[KK], fb[S][j]);
        else g8_rdtr<KK * 16 * PB, KK * 16 * PB + 4 * PB>(baseB[j] + boff, fb[S][j]);
      }
    };
    auto landedp = [&](int S) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < TM; ++i) g8_tie(fa[S][i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) g8_tie(fb[S][j]);
    };
    const bool prio = p.dbg & 16;                              // experiment: s_setprio 1 around every MFMA group
    auto mmap = [&](int S) {
      if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g8_val(fa[S][i]), g8_val(fb[S][j]), acc[i][j], 0, 0, 0);
      if (prio) __builtin_amdgcn_s_setprio(0);
    };
    // (round 6, p.stag) The two waves of a SIMD are waves w and w + NW / 2.  In lock-step both issue their 7 LDS-DMA instructions of the
    // tile after next right behind the barrier (~100 cycles each beside reads and MFMAs) while the SIMD's matrix pipe has nothing to do.
    // With `stag` the upper half of the waves issues its share a quarter of a k-tile LATER (tile it + 1 into the buffer the barrier at
    // the end of iteration it - 1 freed, behind the first MFMA group of iteration it; still covered by the vmcnt(0) in front of this
    // iteration's barrier), so one wave of every SIMD multiplies while the other one issues.
    const bool late = p.stag && wave >= G8_NW / 2;           // (half a k-tile later instead: measured, less -- the late half then waits for its own data)
    if (nk > 0) {
      issue(kt_begin, 0);
      if (nk > 1) { issue(kt_begin + 1, 1); g8_wait_vm<NDMA>(); } else g8_wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      rdp(std::integral_constant<int, 0>{}, 0u);
      landedp(0);
      for (int it = 0; it < nk; ++it) {
        const unsigned boff = (unsigned)((it & 1) * STAGE), bnext = (unsigned)(((it + 1) & 1) * STAGE);
        rdp(std::integral_constant<int, 1>{}, boff);
        __builtin_amdgcn_sched_barrier(0);
        mmap(0);
        __builtin_amdgcn_sched_barrier(0);
        landedp(1);
        if (late && it >= 1 && it + 1 < nk) issue(kt_begin + it + 1, (it + 1) & 1);
        rdp(std::integral_constant<int, 2>{}, boff);
        __builtin_amdgcn_sched_barrier(0);
        mmap(1);
        __builtin_amdgcn_sched_barrier(0);
        landedp(0);
        rdp(std::integral_constant<int, 3>{}, boff);
        __builtin_amdgcn_sched_barrier(0);
        mmap(0);
        __builtin_amdgcn_sched_barrier(0);
        landedp(1);                                            // last read of tile `it` has returned
        if (it + 1 < nk) {
          g8_wait_vm<0>();                                     // tile it + 1 (the only group in flight) has landed
          __builtin_amdgcn_s_barrier();
          if (!late && it + 2 < nk) issue(kt_begin + it + 2, it & 1);
          rdp(std::integral_constant<int, 0>{}, bnext);
        }
        __builtin_amdgcn_sched_barrier(0);
        mmap(1);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < nk) landedp(0);
      }
    }
  } else {
  const bool dbg_nodma = p.dbg & 1, dbg_nord = p.dbg & 2, dbg_nomma = p.dbg & 4, dbg_early = p.dbg & 8;
  if (nk > 0) issue(kt_begin, 0);
  if (nk > 1) issue(kt_begin + 1, 1);
  for (int it = 0; it < nk; ++it) {
    // tile `it` has landed once this wave's older DMA group is done (the younger one, tile it+1, stays in flight) AND
    // every other wave says the same of its part: wait, then barrier, then read
    if (it + 1 < nk) g8_wait_vm<NDMA>(); else g8_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if constexpr (!ASMRD) {
      const char* la = smem + (it & 1) * STAGE;
      const char* lb = la + GA::BYTES;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        g8_bf16x8_t af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = g8_frag_km<BM>(la, (wm * TM + i) * 32, kk, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = g8_frag_km<BN>(lb, (wn * TN + j) * 32, kk, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      // every fragment read of this buffer has returned (the MFMAs consumed them); once all waves are here it may be refilled
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (it + 2 < nk && !dbg_nodma) issue(kt_begin + it + 2, it & 1);
    } else {
      const unsigned boff = (unsigned)((it & 1) * STAGE);
      G8F<AK> fa[2][TM];
      G8F<BK> fb[2][TN];
      // k-step KK of this tile -> register set KK & 1
      auto rd = [&](auto kkc) {
        if (dbg_nord) return;
        constexpr int KK = decltype(kkc)::value;
        constexpr int S = KK & 1;
        constexpr int PA = BM * 2, PB = BN * 2;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if constexpr (AK) g8_rd128<0>(baseA[i] + boff + kx[KK], fa[S][i]);
          else g8_rdtr<KK * 16 * PA, KK * 16 * PA + 4 * PA>(baseA[i] + boff, fa[S][i]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (BK) g8_rd128<0>(baseB[j] + boff + kx[KK], fb[S][j]);
          else g8_rdtr<KK * 16 * PB, KK * 16 * PB + 4 * PB>(baseB[j] + boff, fb[S][j]);
        }
      };
      auto landed = [&](int S) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < TM; ++i) g8_tie(fa[S][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) g8_tie(fb[S][j]);
      };
      auto mma = [&](int S) {
        if (dbg_nomma) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g8_val(fa[S][i]), g8_val(fb[S][j]), acc[i][j], 0, 0, 0);
      };
      // (sched_barrier on both sides of every MFMA group: left alone, hipcc hoists the NEXT group's lgkmcnt wait to just
      //  behind the first MFMA, so the wave sat out the LDS latency with five MFMAs still to issue)
      if (dbg_early && it + 2 < nk) issue(kt_begin + it + 2, it & 1);     // (timing experiment: overwrites the tile being read)
      rd(std::integral_constant<int, 0>{});
      landed(0);
      rd(std::integral_constant<int, 1>{});                  // the reads of k-step 1 fly under the MFMAs of k-step 0
      __builtin_amdgcn_sched_barrier(0);
      mma(0);
      __builtin_amdgcn_sched_barrier(0);
      landed(1);
      rd(std::integral_constant<int, 2>{});
      __builtin_amdgcn_sched_barrier(0);
      mma(1);
      __builtin_amdgcn_sched_barrier(0);
      landed(0);
      rd(std::integral_constant<int, 3>{});
      __builtin_amdgcn_sched_barrier(0);
      mma(0);
      __builtin_amdgcn_sched_barrier(0);
      landed(1);                                             // last read of this buffer has returned: it may be refilled
      __builtin_amdgcn_s_barrier();
      if (it + 2 < nk && !dbg_nodma && !dbg_early) issue(kt_begin + it + 2, it & 1);     // (address arithmetic + DMA issue interleave with the last MFMAs)
      mma(1);
    }
  }

  }
  // ---- epilogue.  accumulator element r of tile (i, j): row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), col = lane & 31
  const int ncol0 = n0 + wn * TN * 32;
  if (p.atomic || nk <= 0) {
    if (nk <= 0) return;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = ncol0 + j * 32 + (lane & 31);
      long cb = n;
      if (BATCHED) { const int bb = (int)__umulhi((unsigned)n, p.ninv); cb = (long)bb * p.dbs + (n - bb * p.Nsub); }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          unsafeAtomicAdd(reinterpret_cast<float*>(p.D) + (long)m * p.ldd + cb, acc[i][j][r]);
        }
    }
    return;
  }
  // rows leave through LDS as 16-byte stores (an accumulator holds a COLUMN per lane)
  constexpr int CPR = TN * 4;                                 // 8-column chunks per staged row
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * SP);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    float r1m[16];
    if (p.r1_m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) r1m[r] = p.r1_m[m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
    }
    __syncthreads();                                          // operand buffers / previous block fully consumed
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      int n = ncol0 + j * 32 + (lane & 31);
      if (BATCHED) n -= (int)__umulhi((unsigned)n, p.ninv) * p.Nsub;
      const float bn = p.bias_n ? p.bias_n[n] : 0.f;
      const float r1n = p.r1_n ? p.r1_n[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = acc[i][j][r] + bn;
        if (p.r1_m) v += r1m[r] * r1n;
        stg[row * SP + j * 32 + (lane & 31)] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int itc = 0; itc < (32 * CPR) / 64; ++itc) {
      const int c = itc * 64 + lane;
      const int row = c / CPR, cc = c % CPR;
      const int m = m0 + (wm * TM + i) * 32 + row;
      const int n = ncol0 + cc * 8;
      long cb = n;
      if (BATCHED) { const int bb = (int)__umulhi((unsigned)n, p.ninv); cb = (long)bb * p.dbs + (n - bb * p.Nsub); }
      float v[8];
      const float4 a = *reinterpret_cast<const float4*>(stg + row * SP + cc * 8);
      const float4 b = *reinterpret_cast<const float4*>(stg + row * SP + cc * 8 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      const long od = (long)m * p.ldd + cb;
      if (p.R) {                                              // (pair backward: d f = dX(other call) + this product)
        float rv[8];
        ldv<DT_BF16, 8>(p.R, od, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      if (p.ddt == DT_F32) {
        stv<DT_F32, 4>(p.D, od, *reinterpret_cast<const float(*)[4]>(v));
        stv<DT_F32, 4>(p.D, od + 4, *reinterpret_cast<const float(*)[4]>(v + 4));
      } else {
        stv<DT_BF16, 8>(p.D, od, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// the one variant hipcc cannot fit into 256 VGPRs (MN-major A, K-major B, 256 columns, two-level contraction: spills into the
// k-loop); no caller of the adapter schedule has that shape -- gemm8_try leaves it to the tiled engine
template <bool AK, bool BK, int BN> constexpr bool G8_TWO_OK = !(!AK && BK && BN == 256);

template <bool AK, bool BK, int BN, bool PIPE>
static void g8_launch_lay(const G8& k, bool batched, bool two, dim3 grid, hipStream_t s) {
  if (batched)  hipLaunchKernelGGL((gemm8_kernel<AK, BK, BN, true, false, PIPE>), grid, dim3(G8_NT), 0, s, k);
  else if (two) {
    if constexpr (G8_TWO_OK<AK, BK, BN>) hipLaunchKernelGGL((gemm8_kernel<AK, BK, BN, false, true, PIPE>), grid, dim3(G8_NT), 0, s, k);
  }
  else          hipLaunchKernelGGL((gemm8_kernel<AK, BK, BN, false, false, PIPE>), grid, dim3(G8_NT), 0, s, k);
}
template <int BN, bool PIPE>
static void g8_launch_p(const G8& k, int ak, int bk, bool batched, bool two, dim3 grid, hipStream_t s) {
  if (ak && bk)       g8_launch_lay<true, true, BN, PIPE>(k, batched, two, grid, s);
  else if (ak && !bk) g8_launch_lay<true, false, BN, PIPE>(k, batched, two, grid, s);
  else if (!ak && bk) g8_launch_lay<false, true, BN, PIPE>(k, batched, two, grid, s);
  else                g8_launch_lay<false, false, BN, PIPE>(k, batched, two, grid, s);
}
// "g8pipe" (dgsct_test_tune / DGSCT_G8PIPE): 1 = the cross-tile pipelined k-loop (default), 0 = the round-3 loop (two barriers per k-tile)
static std::atomic<int> g_g8pipe{getenv("DGSCT_G8PIPE") ? atoi(getenv("DGSCT_G8PIPE")) : 1};
static std::atomic<int> g_g8stag{getenv("DGSCT_G8STAG") ? atoi(getenv("DGSCT_G8STAG")) : 1};      // 1: the upper wave half issues behind the first MFMA group of the next iteration (default), 0: lock-step
int gemm8_stag_mode(int set) { const int old = g_g8stag.load(); if (set >= 0) g_g8stag.store(set ? 1 : 0); return old; }
int gemm8_pipe_mode(int set) { const int old = g_g8pipe.load(); if (set >= 0) g_g8pipe.store(set ? 1 : 0); return old; }
template <int BN>
static void g8_launch(const G8& k, int ak, int bk, bool batched, bool two, dim3 grid, hipStream_t s) {
  if (g_g8pipe.load(std::memory_order_relaxed)) g8_launch_p<BN, true>(k, ak, bk, batched, two, grid, s);
  else g8_launch_p<BN, false>(k, ak, bk, batched, two, grid, s);
}

static inline bool g8_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline double g8_eff(long wgs) { return (double)wgs / (double)(((wgs + 255) / 256) * 256); }

int gemm8_mode(int set) {
  static std::atomic<int> mode{getenv("DGSCT_GEMM8") ? atoi(getenv("DGSCT_GEMM8")) : 1};
  const int old = mode.load(std::memory_order_relaxed);
  if (set >= 0) mode.store(set, std::memory_order_relaxed);
  return old;
}

// "g8wg" (dgsct_test_tune / DGSCT_G8WG; 0 = off): late-stage weight gradients (few 256 x 256 output tiles, a 23 040 .. 40 960-row
// contraction) on this kernel with only as many split-K slabs as give ~`target` workgroups: a FEW CUs for longer instead of every CU
// for a short burst that ends in 16-40-way atomics.  Round 4 measured the opposite choice (60 splits: 36.7 -> 61 us, kept off).  With
// the aux join deferred (round 5) only the CU-time and the L2 traffic of these products count, not their latency: in-process A/B
// (tools/call_overlap.py AB=g8wg=..): 32 -> -0.4 .. -0.7 ms per step, 64 -> -0.25, 96 -> +0.15, 16 -> noise.  Default 32.
static std::atomic<int> g_g8wg{getenv("DGSCT_G8WG") ? atoi(getenv("DGSCT_G8WG")) : 32};
int gemm8_wg_target(int set) { const int old = g_g8wg.load(); if (set >= 0) g_g8wg.store(set); return old; }

bool gemm8_try(const Ctx& ctx, const Gemm& g) {
  // DGSCT_GEMM8=0 switches the kernel off (A/B runs against the tiled engine); =2 also takes shapes below the size gates
  const int mode = gemm8_mode(-1);
  if (!mode || ctx.mode != DT_BF16) return false;
  if (g.R && (g.atomic || g.R2 || g.ddt != DT_BF16 || g.rdt != DT_BF16 || g.ldr != g.ldd || g.rbs != g.dbs || g.beta != 1.f || !g8_al16(g.R))) return false;
  if (g.act != ACT_NONE || g.mask || g.R2 || g.bias_m || g.alpha_ptr || g.alpha != 1.f || g.sm_scale || g.sm_dot) return false;
  if (g.m_mod > 0 || g.bias_n_bs != 0) return false;
  if ((g.r1_m == nullptr) != (g.r1_n == nullptr)) return false;
  const long kflat = (long)g.K * g.KB;
  if (kflat % 64 || kflat < 256 || g.K % 8) return false;
  // Short contractions (round 5): the late stages' [rows, C] x [C, C] products on 128 x 128 tiles pull every operand panel through L2
  // 3-4 times; 256 x 192 / 256 x 256 tiles halve that and hold <= 180-640 CUs' worth of tiles in one round.  Measured alone / in the
  // step (tools/gemm8_gate_ab.py, tools/call_overlap.py AB=gemm8=2): 23 040 x 384 x 512 26.6 -> 21.3 us, 92 160 x 192 x 256 37 -> 27 us, the
  // batched 1024 x 192 x 576 x 160 frames 69.5 -> 54.2 us, step -0.35 ms; 40 960 x 384 x 384 loses 6 % (two column tiles of a 6-tile k-loop)
  // and split-K weight gradients keep their 1024 gate.
  if (mode < 2 && kflat < 1024 && (g.atomic || !(kflat >= 512 || g.N * (long)(g.batch > 1 && g.A.bs == 0 ? g.batch : 1) <= 256))) return false;
  if (g.M % G8_ROWS || g.M < G8_ROWS) return false;
  if (!g8_al16(g.A.p) || !g8_al16(g.B.p) || !g8_al16(g.D) || g.A.ld % 8 || g.B.ld % 8 || g.A.kbs % 8 || g.B.kbs % 8 || g.B.bs % 8) return false;
  if (!g.A.kmajor && g.M % 8) return false;
  if (g.KB > 1 && (unsigned long long)g.K * g.KB * g.K >= 0x100000000ULL) return false;
  const bool batched = g.batch > 1;
  const bool two = g.KB > 1;
  long Ntot = g.N;
  if (batched) {
    if (g.A.bs != 0 || g.B.bs == 0 || two || g.atomic || g.N % 8) return false;
    Ntot = (long)g.N * g.batch;
    if (Ntot * (long)g.N >= 0x100000000LL) return false;        // frame split by multiply-high
  } else if (g.B.bs != 0 && g.batch > 1) return false;
  if (g.atomic && (g.ddt != DT_F32)) return false;
  if (!g.atomic && (g.ldd % 8 || g.dbs % 8)) return false;
  if ((g.r1_m || g.bias_n) && g.atomic) return false;
  if (!g.B.kmajor && Ntot % 8) return false;
  // column tile: the one that fills whole rounds of 256 workgroups better (ties: the wider tile)
  const int tiles_m = g.M / G8_ROWS;
  int BN = 0;
  double best = -1;
  for (int bn : {256, 192}) {
    if (Ntot % bn) continue;
    const long tiles = (long)tiles_m * (Ntot / bn);
    double e = g.atomic ? 1.0 : g8_eff(tiles);                  // split-K fills the rounds itself
    if (bn == 192) e *= 0.97;
    if (g.atomic && g.sole_writer && !batched)                  // unsplit weight gradient (below): the most tiles that fit one round
      e = (tiles >= 96 && tiles <= 256) ? 2.0 + (double)tiles / 256.0 : e;
    if (e > best) { best = e; BN = bn; }
  }
  if (!BN) return false;
  if (two && !g.A.kmajor && g.B.kmajor && BN == 256) return false;      // (G8_TWO_OK)
  const int tiles_n = (int)(Ntot / BN);
  const long tiles = (long)tiles_m * tiles_n;
  const int kt_total = (int)(kflat / 64);
  int splitk = 1;
  bool plain = false;                                           // atomic request served as ONE unsplit pass with plain stores
  if (g.atomic && g.sole_writer && !batched && tiles >= 96 && tiles <= 256 && kt_total >= 64 && g.ldd % 8 == 0) {
    // The remap weight gradient dWn (144 / 192 tiles of a 15 360-deep contraction): one workgroup per tile on as many CUs, the whole
    // contraction in one k-loop, plain fp32 rows out.  Split 7 ways to fill 256 CUs it paid seven 256 x 256 slabs of fp32 atomics
    // per tile (66 M per launch) and four prologue / epilogue rounds: 420-465 us on every CU; unsplit it holds 144-192 CUs for
    // 240 k-tiles and leaves the rest to the kernels of the other streams (stage 0 is throughput-bound: CU-time is what counts).
    plain = true;
  } else if (g.atomic) {

    // Few output tiles = many splits of a handful of k-tiles each: prologue / epilogue bound and 60-way atomics per address.
    // Measured (tools/gemm8_ab_all.sh): 512 x 512 x 23 040 36.7 -> 61 us, 1024 x 1024 x 5 760 35.9 -> 59 us -- the tiled engine keeps them.
    const int wgt = g_g8wg.load(std::memory_order_relaxed);
    if (wgt > 0 && tiles < 96 && !two && kt_total >= 48) {
      splitk = (int)((wgt + tiles - 1) / tiles);
      if (splitk > kt_total / 8) splitk = kt_total / 8;
      if (splitk < 1) splitk = 1;
      goto split_done;
    }
    if (mode < 2 && tiles < 96) return false;
    // dWn with both operands K-major (2304 x 4096 x (160 x 96)): 434 vs 443 us alone, 455 vs 515 us inside the step (seven 256 x 256
    // fp32 atomic slabs per tile against the tiled engine's two) -- stays on the tiled engine; the K-major x MN-major one gains 12 %
    if (mode < 2 && two && g.A.kmajor && g.B.kmajor) return false;      // (split: seven atomic slabs against the tiled engine's two)
    // enough splits for >= ~2 full rounds, each walking >= 6 k-tiles; among those the best-filled last round
    int smax = kt_total / 6; if (smax < 1) smax = 1; if (smax > 64) smax = 64;
    double be = -1;
    for (int s = 1; s <= smax; ++s) {
      const long w = tiles * s;
      if (w < 192 && s < smax) continue;
      if (w > 1536 && s > 1) break;
      const double e = g8_eff(w) - 0.01 * s;                    // mild preference for fewer splits (atomic traffic)
      if (e > be) { be = e; splitk = s; }
    }
  } else if (mode < 2 && tiles < 160) {
    return false;                                               // too few tiles to fill the chip without split-K
  }
  if (mode < 2 && !plain && tiles * splitk < 128) return false;
split_done:
  const int kt_per_split = (kt_total + splitk - 1) / splitk;
  splitk = (kt_total + kt_per_split - 1) / kt_per_split;

  G8 k;
  k.M = g.M; k.Ntot = (int)Ntot; k.Nsub = g.N;
  k.ninv = batched ? (unsigned)((0x100000000ULL + (unsigned long long)g.N - 1) / (unsigned long long)g.N) : 0u;
  k.K = g.K; k.kflat = (int)kflat;
  k.kinv = two ? (unsigned)((0x100000000ULL + (unsigned long long)g.K - 1) / (unsigned long long)g.K) : 0u;
  k.tiles_m = tiles_m; k.tiles_n = tiles_n; k.kt_total = kt_total; k.kt_per_split = kt_per_split;
  k.A = (const char*)g.A.p; k.lda = g.A.ld; k.a_kbs = g.A.kbs;
  k.B = (const char*)g.B.p; k.ldb = g.B.ld; k.b_bs = g.B.bs; k.b_kbs = g.B.kbs;
  k.D = (char*)g.D; k.ddt = g.ddt; k.ldd = g.ldd; k.dbs = g.dbs; k.atomic = plain ? 0 : g.atomic;
  k.r1_m = g.r1_m; k.r1_n = g.r1_n; k.bias_n = g.bias_n;
  k.R = (const char*)g.R;
  static const int gm_env = getenv("DGSCT_GEMM8_GM") ? atoi(getenv("DGSCT_GEMM8_GM")) : 4;
  k.gm = gm_env < 1 ? 1 : (gm_env > tiles_m ? tiles_m : gm_env);
  static const int dbg_env = getenv("DGSCT_GEMM8_DBG") ? atoi(getenv("DGSCT_GEMM8_DBG")) : 0;
  k.dbg = dbg_env;
  k.stag = g_g8stag.load(std::memory_order_relaxed);
  dim3 grid((unsigned)tiles, splitk, 1);
  hipStream_t s = (hipStream_t)ctx.stream;
  GemmProfShape shp{g.M, g.N, g.K, g.KB, g.batch, splitk, BN == 256 ? 8 : 9, g.A.kmajor, g.B.kmajor, g.atomic, !g.atomic, 0.0};
  shp.bytes = ((double)g.M * kflat + (double)Ntot * kflat) * 2 + (double)g.M * Ntot * (g.ddt == DT_F32 ? 4 : 2);
  void* rec = gemm_prof_begin(s, 2.0 * g.M * (double)Ntot * (double)kflat, shp);
  if (BN == 256) g8_launch<256>(k, g.A.kmajor, g.B.kmajor, batched, two, grid, s);
  else g8_launch<192>(k, g.A.kmajor, g.B.kmajor, batched, two, grid, s);
  gemm_prof_end(rec, s);
  return true;
}

}  // namespace dgsct
