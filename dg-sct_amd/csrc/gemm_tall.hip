// Weight gradients over the token rows (gfx950, bf16):  D[b][m][n] += sum_r A[r][b*a_bs + m] * B[r][b*b_bs + n]
//
// Both operands are [rows][width] activations / cotangents (MN-major in the GEMM engine's terms: the contraction runs ACROSS rows),
// the output is a small weight-shaped matrix (6 x 48 ... 128 x 128) and the contraction is 92 160 ... 655 360 rows deep: the five
// products per adapter call that the backward releases on the aux stream at stages 0-1 (dWu, dWd, dWv2, dWv1, dWc / dWn's sibling).
// On the tiled engine they ran as 250-320-way split-K launches of 128-row-padded tiles whose operand staging spends most of its load
// slots on clamped columns: 1.1-2 TB/s for what is a pure stream of two [rows, <= 128] tensors.
//
// Here a workgroup walks a contiguous range of 64-row blocks.  A block of BOTH operands is one contiguous byte range each (whole
// rows: the column slab of a group is picked when the fragment is read), fetched with 16-byte loads one block ahead, written to
// LDS as is, and read back MN-major (ds_read_b64_tr_b16) as the two operands of v_mfma_f32_32x32x16_bf16 with the ROWS as the
// contraction.  The (group, m-tile, n-tile) tiles of the output are dealt round-robin to the four waves (<= 4 tiles = 64
// accumulator registers each); at the end every wave adds its tiles to D with fp32 atomics.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include "prims.h"
#include "device_util.h"
#include "mma_tile.h"
#include "err.h"
#include "gemm_int.h"

namespace dgsct {

namespace {
typedef unsigned gt_u32x4 __attribute__((ext_vector_type(4)));
constexpr int GT_RB = 64;                                    // rows per block
constexpr int GT_MAXT = 4;                                   // tiles per wave
constexpr int GT_MAXW = 512;                                 // widest operand row (elements): 64 x 512 x 2 B = 64 KB per operand image

struct GT {
  const unsigned short* A; const unsigned short* B; float* D;
  long lda, ldb, a_bs, b_bs, dbs, ldd;
  int M, N, batch, mt, nt, ntiles;
  long nblk; int bpw;                                        // 64-row blocks in all / per workgroup
  int na, nb;                                                // 16-byte chunks of one block of A / B
};

template <int NCH>                                           // 16-byte chunks per thread and operand block (upper bound)
__global__ __launch_bounds__(256) void gemm_tall_k(const GT p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* imgA = smem;
  char* imgB = smem + (size_t)p.na * 16 + 64;                // (+64: fragments of a padded tile may read a little past the block)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long blk0 = (long)blockIdx.x * p.bpw;
  long blk1 = blk0 + p.bpw; if (blk1 > p.nblk) blk1 = p.nblk;
  if (blk0 >= blk1) return;
  const int pa = (int)p.lda * 2, pb = (int)p.ldb * 2;
  // this wave's tiles: t = wave, wave + 4, ...  ->  (group, m-tile, n-tile)
  // (a group's slab may start on any even column -- 6 at ds = 12, g = 2 -- but a transpose read wants 4-column alignment: the tile
  //  starts at the aligned column below and its first `sh` rows / columns are skipped when it is written out)
  int colA[GT_MAXT], colB[GT_MAXT], shA[GT_MAXT], shB[GT_MAXT];
#pragma unroll
  for (int i = 0; i < GT_MAXT; ++i) {
    int t = wave + 4 * i; t = t < p.ntiles ? t : 0;
    const int b = t / (p.mt * p.nt), r = t - b * p.mt * p.nt, m = r / p.nt, n = r - m * p.nt;
    const int ca = (int)(b * p.a_bs) + 32 * m, cb = (int)(b * p.b_bs) + 32 * n;
    colA[i] = ca & ~3; shA[i] = ca & 3;
    colB[i] = cb & ~3; shB[i] = cb & 3;
  }
  mt_f32x16 acc[GT_MAXT];
#pragma unroll
  for (int i = 0; i < GT_MAXT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  gt_u32x4 ra[NCH], rb[NCH];
  auto fetch = [&](long blk) {                               // unconditional loads from clamped chunk indices
    const gt_u32x4* sa = reinterpret_cast<const gt_u32x4*>(p.A + blk * GT_RB * p.lda);
    const gt_u32x4* sb = reinterpret_cast<const gt_u32x4*>(p.B + blk * GT_RB * p.ldb);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = i * 256 + tid;
      ra[i] = sa[c < p.na ? c : p.na - 1];
      rb[i] = sb[c < p.nb ? c : p.nb - 1];
    }
  };
  fetch(blk0);
  for (long blk = blk0; blk < blk1; ++blk) {
    __syncthreads();                                         // the previous block's fragments have been read
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = i * 256 + tid;
      if (c < p.na) reinterpret_cast<gt_u32x4*>(imgA)[c] = ra[i];
      if (c < p.nb) reinterpret_cast<gt_u32x4*>(imgB)[c] = rb[i];
    }
    __syncthreads();
    fetch(blk + 1 < blk1 ? blk + 1 : blk);
#pragma unroll
    for (int i = 0; i < GT_MAXT; ++i) {
      if (wave + 4 * i < p.ntiles) {                         // (wave-uniform)
#pragma unroll
        for (int kk = 0; kk < GT_RB / 16; ++kk)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mt_frag_mn(imgA, pa, colA[i], kk, lane), mt_frag_mn(imgB, pb, colB[i], kk, lane),
                                                           acc[i], 0, 0, 0);
      }
    }
  }
  // accumulator element r: m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), n = lane & 31
#pragma unroll
  for (int i = 0; i < GT_MAXT; ++i) {
    const int t = wave + 4 * i;
    if (t < p.ntiles) {
      const int b = t / (p.mt * p.nt), r0 = t - b * p.mt * p.nt, mi = r0 / p.nt, ni = r0 - mi * p.nt;
      const int nl = (lane & 31) - shB[i], n = 32 * ni + nl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) - shA[i], m = 32 * mi + ml;
        if (ml >= 0 && nl >= 0 && m < p.M && n < p.N) unsafeAtomicAdd(p.D + b * p.dbs + (long)m * p.ldd + n, acc[i][r]);
      }
    }
  }
}

std::atomic<int> g_tall{-1};
}  // namespace

int gemm_tall_mode(int set) {
  if (g_tall.load(std::memory_order_relaxed) < 0) g_tall.store(getenv("DGSCT_NO_GEMM_TALL") ? 0 : 1, std::memory_order_relaxed);
  const int old = g_tall.load(std::memory_order_relaxed);
  if (set >= 0) g_tall.store(set > 2 ? 1 : set, std::memory_order_relaxed);
  return old;
}

bool gemm_tall_try(const Ctx& ctx, const Gemm& g0) {
  if (!gemm_tall_mode(-1) || ctx.mode != DT_BF16) return false;
  Gemm g = g0;
  // a two-level contraction over (frame, row) whose frames lie back to back in BOTH operands is one flat stream of KB * K rows
  // (dWc of the audio direction: sum_b dT2[b]^T Y[b])
  if (g.KB > 1 && !g.A.kmajor && !g.B.kmajor && g.A.kbs == (long)g.K * g.A.ld && g.B.kbs == (long)g.K * g.B.ld &&
      (long)g.K * g.KB < 0x7fffffffL) { g.K *= g.KB; g.KB = 1; }
  if (g.A.kmajor || g.B.kmajor || g.KB != 1 || !g.atomic || g.ddt != DT_F32) return false;
  if (g.act != ACT_NONE || g.mask || g.R || g.R2 || g.bias_m || g.bias_n || g.r1_m || g.r1_n || g.alpha_ptr || g.alpha != 1.f || g.sm_scale || g.sm_dot) return false;
  // tall only; whole 64-row blocks.  Default gate = the stage-0 depths (368 640 / 655 360 rows): in the step the stage-1 products
  // (92 160 / 163 840 rows) measured +30..+47 us per pair with this kernel although each is as fast or faster alone
  // (tools/call_overlap.py AB=gemmtall); "gemmtall" = 2 takes everything from 16 384 rows (tests)
  const long kmin = gemm_tall_mode(-1) >= 2 ? 16384 : 262144;
  if (g.K < kmin || g.K % GT_RB) return false;
  if (g.A.ld > GT_MAXW || g.B.ld > GT_MAXW) return false;
  if (g.A.ld % 4 || g.B.ld % 4) return false;                // 8-byte aligned transpose reads
  if ((reinterpret_cast<uintptr_t>(g.A.p) & 15) || (reinterpret_cast<uintptr_t>(g.B.p) & 15)) return false;
  if ((g.A.ld * GT_RB * 2) % 16 || (g.B.ld * GT_RB * 2) % 16) return false;
  if (g.A.bs % 2 || g.B.bs % 2) return false;
  const int wsa = g.batch > 1 && g.A.bs % 4 ? 2 : 0, wsb = g.batch > 1 && g.B.bs % 4 ? 2 : 0;     // worst shift of a slab start
  // The kernel streams WHOLE 64-row x ld blocks starting at A.p / B.p, so both pointers must sit at COLUMN 0 of a row (ADVICE r4:
  // a base + column_offset operand reads column_offset elements past the tensor in its last block).  That cannot be seen from the
  // pointer; what can be is checked -- the column slabs must lie inside a row -- and the contract is stated in gemm_int.h: every
  // caller in plan.cpp passes whole tensors (slabs selected through `bs`, never through the base pointer).
  if ((long)(g.batch - 1) * g.A.bs + g.M > g.A.ld || (long)(g.batch - 1) * g.B.bs + g.N > g.B.ld) return false;
  if ((wsa && g.M > 30) || (wsb && g.N > 30)) return false;   // (a shifted slab must fit ONE tile: the stage-0 bottleneck widths do)
  const int mt = (g.M + 31) / 32, nt = (g.N + 31) / 32;
  const int ntiles = g.batch * mt * nt;
  if (ntiles > 4 * GT_MAXT) return false;
  GT a;
  a.A = (const unsigned short*)g.A.p; a.B = (const unsigned short*)g.B.p; a.D = (float*)g.D;
  a.lda = g.A.ld; a.ldb = g.B.ld; a.a_bs = g.A.bs; a.b_bs = g.B.bs; a.dbs = g.dbs; a.ldd = g.ldd;
  a.M = g.M; a.N = g.N; a.batch = g.batch; a.mt = mt; a.nt = nt; a.ntiles = ntiles;
  a.nblk = g.K / GT_RB;
  a.na = (int)(g.A.ld * GT_RB * 2 / 16); a.nb = (int)(g.B.ld * GT_RB * 2 / 16);
  const int nch = ((a.na > a.nb ? a.na : a.nb) + 255) / 256;
  if (nch > 8) return false;
  const size_t shmem = (size_t)a.na * 16 + 64 + (size_t)a.nb * 16 + 64;
  // workgroups: as many as are resident, fewer for big outputs (every workgroup adds its whole output with atomics), >= 4 blocks each
  long nwg = 512;
  if ((long)g.M * g.N * g.batch >= 8192) nwg = 256;
  if (nwg > a.nblk / 4) nwg = a.nblk / 4;
  if (nwg < 1) nwg = 1;
  a.bpw = (int)((a.nblk + nwg - 1) / nwg);
  nwg = (a.nblk + a.bpw - 1) / a.bpw;
  hipStream_t s = (hipStream_t)ctx.stream;
  GemmProfShape shp{g.M, g.N, g.K, 1, g.batch, (int)nwg, 11, 0, 0, 1, 0, 0.0};
  shp.bytes = ((double)g.A.ld + (double)g.B.ld) * g.K * 2;
  void* rec = gemm_prof_begin(s, 2.0 * g.M * (double)g.N * (double)g.K * g.batch, shp);
  auto launch = [&](auto kern) {
    if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), shmem, s, a);
  };
  if (nch <= 2) launch(gemm_tall_k<2>);
  else if (nch <= 4) launch(gemm_tall_k<4>);
  else launch(gemm_tall_k<8>);
  gemm_prof_end(rec, s);
  return true;
}

}  // namespace dgsct
