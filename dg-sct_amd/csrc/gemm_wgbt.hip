// The gate MLPs' weight gradients in ONE launch (gfx950, bf16):  D_p[m][n] = sum_b A_p[b][m] * B_p[b][n]  for up to 4 problems p
//
// Four of an adapter call's weight gradients contract over the FRAMES only (b < BT = 160): d fc_affine_v_c_att.weight = dpre^T . q,
// d fc_affine_bottleneck.weight = dq^T . m1, d fc_affine_audio_1 / _2.weight = dpa^T . aE -- [C or C/2] x [C or C/2] outputs, 0.01-0.3 GFLOP
// each.  On the tiled engine they were four launches of 2-3 workgroups (one 128-wide tile row of a 160-deep contraction), 17-31 us each on
// the weight-gradient stream: latency, not work.  Here one launch covers the 64 x 64 output tiles of all four ("grouped
// non-atomic weight-gradient launch", for the group it fits: the deep ones over the token rows are gemm_tall / gemm8 launches of their own).
//
// A workgroup owns one 64 x 64 tile: both operands are [BT][width] matrices (MN-major: the contraction runs across rows), staged 64 rows
// at a time as 16-byte chunks into LDS and read back transposed (ds_read_b64_tr_b16) as the operands of v_mfma_f32_32x32x16_bf16; wave w
// computes the 32 x 32 quadrant (w >> 1, w & 1).  Plain fp32 stores (each output has one writer).
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
constexpr int WB_PITCH = 64 * 2 + 64;                       // 192 B rows: the four k-rows of a transpose read fall on the four quarters of the bank line

struct WgBtK {
  WgBtJob j[WGBT_MAX];
  int tile0[WGBT_MAX + 1];                                  // first tile of each problem in the grid
  int tn[WGBT_MAX];                                         // column tiles of each problem
  int n;
};

__global__ __launch_bounds__(256) void wgrad_bt_k(const WgBtK t) {
  __shared__ __attribute__((aligned(16))) char sA[64 * WB_PITCH];
  __shared__ __attribute__((aligned(16))) char sB[64 * WB_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < WGBT_MAX; ++i)
    if (i < t.n && (int)blockIdx.x >= t.tile0[i]) pi = i;
  const WgBtJob& p = t.j[pi];
  const int tile = blockIdx.x - t.tile0[pi];
  const int m0 = (tile / t.tn[pi]) * 64, n0 = (tile % t.tn[pi]) * 64;
  mt_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const unsigned short* A = reinterpret_cast<const unsigned short*>(p.A);
  const unsigned short* B = reinterpret_cast<const unsigned short*>(p.B);
  for (int k0 = 0; k0 < p.K; k0 += 64) {
    __syncthreads();                                        // the previous block's fragments have been read
#pragma unroll
    for (int i = 0; i < 2; ++i) {                           // 64 rows x 8 chunks of 8 columns per operand, 2 per thread
      const int c = i * 256 + tid, r = c >> 3, cc = (c & 7) * 8;
      uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
      if (k0 + r < p.K) {
        if (m0 + cc < p.M) va = *reinterpret_cast<const uint4*>(A + (long)(k0 + r) * p.lda + m0 + cc);     // (widths are multiples of 8: whole chunks)
        if (n0 + cc < p.N) vb = *reinterpret_cast<const uint4*>(B + (long)(k0 + r) * p.ldb + n0 + cc);
      }
      *reinterpret_cast<uint4*>(sA + r * WB_PITCH + cc * 2) = va;
      *reinterpret_cast<uint4*>(sB + r * WB_PITCH + cc * 2) = vb;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mt_frag_mn(sA, WB_PITCH, 32 * (wave >> 1), kk, lane), mt_frag_mn(sB, WB_PITCH, 32 * (wave & 1), kk, lane),
                                                    acc, 0, 0, 0);
  }
  // accumulator element r: m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), n = lane & 31
  const int n = n0 + 32 * (wave & 1) + (lane & 31);
  if (n < p.N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + 32 * (wave >> 1) + mt_row(r, lane);
      if (m < p.M) p.D[(long)m * p.ldd + n] = acc[r];
    }
  }
}

std::atomic<int> g_wgbt{-1};
}  // namespace

int wgrad_bt_mode(int set) {
  if (g_wgbt.load(std::memory_order_relaxed) < 0) g_wgbt.store(getenv("DGSCT_NO_WGBT") ? 0 : 1, std::memory_order_relaxed);
  const int old = g_wgbt.load(std::memory_order_relaxed);
  if (set >= 0) g_wgbt.store(set ? 1 : 0, std::memory_order_relaxed);
  return old;
}

bool wgrad_bt_supported(const Ctx& ctx, const WgBtJob* jobs, int n) {
  if (!wgrad_bt_mode(-1) || ctx.mode != DT_BF16 || n < 1 || n > WGBT_MAX) return false;
  for (int i = 0; i < n; ++i) {
    const WgBtJob& j = jobs[i];
    if (!j.A || !j.B || !j.D || j.M < 1 || j.N < 1 || j.K < 1) return false;
    if (j.M % 8 || j.N % 8 || j.lda % 8 || j.ldb % 8) return false;                  // whole 16-byte chunks
    if ((reinterpret_cast<uintptr_t>(j.A) & 15) || (reinterpret_cast<uintptr_t>(j.B) & 15)) return false;
  }
  return true;
}

void wgrad_bt(const Ctx& ctx, const WgBtJob* jobs, int n) {
  WgBtK t;
  t.n = n;
  int tiles = 0;
  for (int i = 0; i < WGBT_MAX; ++i) {
    t.tile0[i] = tiles;
    t.tn[i] = 1;
    if (i < n) {
      t.j[i] = jobs[i];
      t.tn[i] = (jobs[i].N + 63) / 64;
      tiles += ((jobs[i].M + 63) / 64) * t.tn[i];
    } else {
      t.j[i] = jobs[0];
    }
  }
  t.tile0[WGBT_MAX] = tiles;
  double flops = 0, bytes = 0;
  for (int i = 0; i < n; ++i) {
    flops += 2.0 * jobs[i].M * (double)jobs[i].N * (double)jobs[i].K;
    bytes += 2.0 * jobs[i].K * ((double)jobs[i].M + (double)jobs[i].N) + 4.0 * jobs[i].M * (double)jobs[i].N;
  }
  // (logged with the GEMM family -- bench.py's roofline block counts its FLOPs and its time -- under the first problem's shape, cfg 12)
  GemmProfShape shp{jobs[0].M, jobs[0].N, jobs[0].K, 1, n, 1, 12, 0, 0, 0, 0, bytes};
  void* rec = gemm_prof_begin(ctx.stream, flops, shp);
  hipLaunchKernelGGL(wgrad_bt_k, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)ctx.stream, t);
  gemm_prof_end(rec, ctx.stream);
}

}  // namespace dgsct
