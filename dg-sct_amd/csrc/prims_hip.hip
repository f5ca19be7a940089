// Non-GEMM kernels of the DG-SCT adapter path for gfx950 (wave64, 256-thread workgroups).
//
// Two access patterns cover everything:
//  * "row" kernels (LayerNorm-like): a power-of-two group of GS <= 64 lanes owns one token row of C
//    channels, each lane holds NV 16-byte vectors in registers; row statistics are wave-shuffle
//    reductions (no LDS); per-channel sums across rows (parameter gradients, BatchNorm sums) stay in
//    registers across the row loop, are combined in LDS (ds_add_f32) and leave the workgroup as one
//    fp32 atomic per channel.
//  * "column-strip" kernels (per-channel affine / reductions over tokens): a thread owns VE consecutive
//    channels and walks rows; loads are 16 B per lane and coalesced along the channel axis.
// All of them are HBM-bound; they read and write each big tensor exactly once.
#include <cstdlib>
#include <hip/hip_runtime.h>
#include "prims.h"
#include "device_util.h"
#include "err.h"

namespace dgsct {

#define STREAM(ctx) ((hipStream_t)(ctx).stream)

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline long cdiv(long a, long b) { return (a + b - 1) / b; }

// Cross-stream ordering with thread-local rings of timing-less events.  A wait captures the record that preceded it, so
// re-recording an event is well defined -- but the rings are long (an event is reused ~60 adapter calls later, when its
// wait has long been submitted) and separate per direction (a fork event is never reused as a join event), so the
// ordering does not depend on how lazily the runtime resolves a pending wait.
static hipEvent_t next_event(int dir) {
  // one ring per (thread, device, direction): an event belongs to the device that was current when it was created, and a
  // forward thread may serve several GPUs (nn.DataParallel replicas, AVS/AVQA call sites)
  constexpr int RING = 1024, MAXDEV = 16;
  struct Ring { hipEvent_t ev[RING]; int n = 0, made = 0; };
  static thread_local Ring* rings[MAXDEV][2] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) { set_error("dgsct: cannot resolve the current HIP device for a stream fork/join"); return nullptr; }
  Ring*& r = rings[dev][dir];
  if (!r) r = new Ring();
  if (r->made < RING) {
    hipError_t e = hipEventCreateWithFlags(&r->ev[r->made], hipEventDisableTiming);
    if (e != hipSuccess) { set_error("hipEventCreateWithFlags: %s", hipGetErrorString(e)); return nullptr; }
    return r->ev[r->made++];
  }
  hipEvent_t e = r->ev[r->n];
  r->n = (r->n + 1) % RING;
  return e;
}
static void order_after(hipStream_t waiter, hipStream_t producer, int dir) {
  hipEvent_t e = next_event(dir);
  if (!e) return;
  hipError_t r = hipEventRecord(e, producer);
  if (r == hipSuccess) r = hipStreamWaitEvent(waiter, e, 0);
  if (r != hipSuccess) set_error("dgsct: cross-stream ordering failed (%s) -- are both streams on the current device?", hipGetErrorString(r));
}
// Launch errors are sticky per thread: one check at the end of an entry point reports the first failed launch of the call
// (a wrong current device, an invalid configuration) instead of returning 0 with uninitialised outputs.
// A pending (non-sticky) HIP error may belong to ANOTHER user of the runtime on this thread (a PyTorch launch whose status has not
// been read yet): it is remembered at entry and NOT consumed while this call raises nothing.  At exit:
//   * no error, or the very error that was pending at entry and no launch of ours in between could tell them apart -> see below;
//   * a DIFFERENT code than the one pending at entry: ours, consumed and reported;
//   * the SAME code as at entry: hipPeekAtLastError cannot say whether one of this call's own launches failed with that
//     code as well (launch errors overwrite each other, they do not queue), and returning 0 with unwritten outputs is the worse
//     mistake -- so it is reported as this call's error too (the message names the ambiguity) and consumed, which also ends the
//     masking for later calls.  (HIP-version note: this relies on hipPeekAtLastError / hipGetLastError returning the LAST non-sticky
//     error of the calling thread, which is what ROCm 6.x / 7.x implement; HIP releases before 5.6 kept a per-device value.)
static thread_local hipError_t g_pre_err = hipSuccess;
void check_async(const char* where) {
  const hipError_t e = hipPeekAtLastError();
  if (e == hipSuccess) return;
  (void)hipGetLastError();
  if (e == g_pre_err)
    set_error("%s: HIP error '%s' was already pending on this thread when the call started (raised by another user of the HIP runtime) "
              "and cannot be told from a failed launch of this call: the outputs are not to be trusted", where, hipGetErrorString(e));
  else
    set_error("%s: HIP error '%s' (is the current device the one that owns the buffers and streams?)", where, hipGetErrorString(e));
  g_pre_err = hipSuccess;
}
void clear_async() { g_pre_err = hipPeekAtLastError(); }
void* stream_create(int priority_class) {
  int least = 0, greatest = 0;                       // numerically: the greatest priority is the SMALLER number
  hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
  if (e != hipSuccess) { set_error("hipDeviceGetStreamPriorityRange: %s", hipGetErrorString(e)); return nullptr; }
  int prio = priority_class < 0 ? greatest : (priority_class > 0 ? least : (least + greatest) / 2);
  hipStream_t s = nullptr;
  e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio);
  if (e != hipSuccess) { set_error("hipStreamCreateWithPriority(%d): %s", prio, hipGetErrorString(e)); return nullptr; }
  return s;
}

void stream_destroy(void* stream) {
  if (stream) (void)hipStreamDestroy(static_cast<hipStream_t>(stream));
}

void stream_fork(const Ctx& ctx) {
  if (!ctx.aux) return;
  order_after((hipStream_t)ctx.aux, (hipStream_t)ctx.stream, 0);
}
void stream_join(const Ctx& ctx) {
  if (!ctx.aux) return;
  order_after((hipStream_t)ctx.stream, (hipStream_t)ctx.aux, 1);
}

void event_record(const Ctx& ctx, void* ev) {
  if (ev && hipEventRecord(static_cast<hipEvent_t>(ev), STREAM(ctx)) != hipSuccess) set_error("hipEventRecord on the caller's event failed");
}
void event_wait(const Ctx& ctx, void* ev) {
  if (ev && hipStreamWaitEvent(STREAM(ctx), static_cast<hipEvent_t>(ev), 0) != hipSuccess) set_error("hipStreamWaitEvent on the caller's event failed");
}

void part_reduce_run(void* stream, const PartJob& j) {
  if (!j.n) return;
  PartTable t = j.t;
  part_reduce(stream, j.part, t, j.n);
  if (j.n2) { PartTable t2 = j.t2; part_reduce(stream, j.part2, t2, j.n2); }
}

void zero(const Ctx& ctx, void* p, size_t bytes) {
  if (bytes) (void)hipMemsetAsync(p, 0, bytes, STREAM(ctx));
}
// two buffers in one launch (backward: the accumulate-into scratch block and the flat gradient buffer; two hipMemsetAsync
// are two fill kernels of 5-8 us each on the dependency chain).  Pointers 16-byte aligned, sizes multiples of 4.
__global__ __launch_bounds__(256) void zero2_k(uint4* a, long na16, unsigned* atail, int nat, uint4* b, long nb16, unsigned* btail, int nbt) {
  const long i0 = (long)blockIdx.x * 256 + threadIdx.x, step = (long)gridDim.x * 256;
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (long i = i0; i < na16; i += step) a[i] = z;
  for (long i = i0; i < nb16; i += step) b[i] = z;
  if (blockIdx.x == 0) {
    if ((int)threadIdx.x < nat) atail[threadIdx.x] = 0;
    if ((int)threadIdx.x < nbt) btail[threadIdx.x] = 0;
  }
}
void zero2(const Ctx& ctx, void* a, size_t abytes, void* b, size_t bbytes) {
  if (((uintptr_t)a | (uintptr_t)b) & 15 || (abytes | bbytes) & 3) { zero(ctx, a, abytes); zero(ctx, b, bbytes); return; }
  const long na = (long)(abytes / 16), nb = (long)(bbytes / 16);
  long blocks = (na + nb + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(zero2_k, dim3((int)blocks), dim3(256), 0, STREAM(ctx), (uint4*)a, na, (unsigned*)((char*)a + na * 16),
                     (int)((abytes - na * 16) / 4), (uint4*)b, nb, (unsigned*)((char*)b + nb * 16), (int)((bbytes - nb * 16) / 4));
}

// ================================================================================================
// row-kernel geometry
// ================================================================================================
struct RowGeom { int gs, nv, rpp, rpc, chunks; };
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
// min_iters: a workgroup keeps per-channel constants in registers and (reduction kernels) flushes per-channel sums with
// one LDS pass + one atomic per channel; it must stream enough rows to amortise that (late stages: 36-144 rows per frame).
// floor_to_cap: target_wgs is the resident capacity (wg_capacity) -> never exceed one round
static RowGeom row_geom(int C, int VE, int N, int B, long target_wgs = 2048, int min_iters = 1, bool floor_to_cap = false) {
  RowGeom g;
  int nvec = C / VE;
  g.gs = 1;
  while (g.gs < nvec && g.gs < 64) g.gs <<= 1;
  g.nv = (nvec + g.gs - 1) / g.gs;
  g.rpp = 256 / g.gs;
  long want = floor_to_cap ? target_wgs / B : cdiv(target_wgs, B);
  long maxc = cdiv(N, (long)g.rpp * min_iters);
  long chunks = want < 1 ? 1 : (want > maxc ? maxc : want);
  g.rpc = (int)(cdiv(cdiv(N, chunks), g.rpp) * g.rpp);
  g.chunks = (int)cdiv(N, g.rpc);
  return g;
}
// dispatch on (mode, vector width): bf16 -> 8 (C%8==0) or 4; f32 -> 4.  C % 4 == 0 is required.
// The register arrays of a row kernel are sized by the template NV: pick the smallest instantiation that
// holds the row (NV = vectors per lane) -- an oversized one costs occupancy (C=128 bf16 needs NV=1, not 3).
#define ROW_L_(KERNEL, DT_, VE_, NV_, GRID, SHMEM, ...) \
  hipLaunchKernelGGL((KERNEL<DT_, VE_, NV_>), GRID, dim3(256), SHMEM, STREAM(ctx), __VA_ARGS__)
#define ROW_DISPATCH_SH(ctx, C, NV, KERNEL, GRID, SHMEM, ...)                                              \
  do {                                                                                                      \
    if ((C) % 4 != 0 || (C) > 1536) { set_error("row kernel: C=%d must be a multiple of 4 and <= 1536", (int)(C)); break; } \
    if ((ctx).mode == DT_BF16) {                                                                            \
      if ((C) % 8 == 0) {                                                                                   \
        if ((NV) <= 1) ROW_L_(KERNEL, DT_BF16, 8, 1, GRID, SHMEM, __VA_ARGS__);                             \
        else if ((NV) <= 2) ROW_L_(KERNEL, DT_BF16, 8, 2, GRID, SHMEM, __VA_ARGS__);                        \
        else ROW_L_(KERNEL, DT_BF16, 8, 3, GRID, SHMEM, __VA_ARGS__);                                       \
      } else ROW_L_(KERNEL, DT_BF16, 4, 6, GRID, SHMEM, __VA_ARGS__);                                       \
    } else {                                                                                                \
      if ((NV) <= 1) ROW_L_(KERNEL, DT_F32, 4, 1, GRID, SHMEM, __VA_ARGS__);                                \
      else if ((NV) <= 2) ROW_L_(KERNEL, DT_F32, 4, 2, GRID, SHMEM, __VA_ARGS__);                           \
      else if ((NV) <= 3) ROW_L_(KERNEL, DT_F32, 4, 3, GRID, SHMEM, __VA_ARGS__);                           \
      else ROW_L_(KERNEL, DT_F32, 4, 6, GRID, SHMEM, __VA_ARGS__);                                          \
    }                                                                                                       \
  } while (0)
#define ROW_DISPATCH(ctx, C, NV, KERNEL, GRID, ...) ROW_DISPATCH_SH(ctx, C, NV, KERNEL, GRID, 0, __VA_ARGS__)
// resident capacity of the instantiation ROW_DISPATCH_SH would launch
#define ROW_FN_(KERNEL, DT_, VE_, NV_) reinterpret_cast<const void*>(&KERNEL<DT_, VE_, NV_>)
#define ROW_CAPACITY(OUT, ctx, C, NV, KERNEL, SHMEM)                                                        \
  do {                                                                                                      \
    const void* fn_;                                                                                        \
    if ((ctx).mode == DT_BF16) {                                                                            \
      if ((C) % 8 == 0) fn_ = (NV) <= 1 ? ROW_FN_(KERNEL, DT_BF16, 8, 1) : ((NV) <= 2 ? ROW_FN_(KERNEL, DT_BF16, 8, 2) \
                                                                                     : ROW_FN_(KERNEL, DT_BF16, 8, 3)); \
      else fn_ = ROW_FN_(KERNEL, DT_BF16, 4, 6);                                                            \
    } else {                                                                                                \
      fn_ = (NV) <= 1 ? ROW_FN_(KERNEL, DT_F32, 4, 1) : ((NV) <= 2 ? ROW_FN_(KERNEL, DT_F32, 4, 2)          \
                      : ((NV) <= 3 ? ROW_FN_(KERNEL, DT_F32, 4, 3) : ROW_FN_(KERNEL, DT_F32, 4, 6)));       \
    }                                                                                                       \
    OUT = wg_capacity(fn_, SHMEM);                                                                          \
  } while (0)
static inline int row_ve(const Ctx& ctx, int C) { return ctx.mode == DT_BF16 ? (C % 8 == 0 ? 8 : 4) : 4; }

// flush per-lane channel accumulators: combine across the row-groups of the workgroup through LDS, then one global
// atomic per channel -- or, with `part`, one plain store per channel of this workgroup's partial sums (finished by
// part_reduce_k).  lds: (256 / gs) * C floats.  The combine is store / barrier / column-sum per quantity: LDS float
// atomics (the first version) cost 31 of the 52 us of tail_bwd at C = 512 (8192 ds atomics per workgroup, 4-way
// contended).
template <int NQ, int VE, int MAXNV>
__device__ __forceinline__ void flush_cols(float (&acc)[NQ][MAXNV][VE], float* lds, int C, int gs, int nv, int gl,
                                           float* const (&dst)[NQ], float* part = nullptr) {
  const int sub = threadIdx.x / gs, rpp = 256 / gs;
  float* p = part ? part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (NQ * C) : nullptr;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    __syncthreads();
#pragma unroll
    for (int v = 0; v < MAXNV; ++v) {
      const int col = (v * gs + gl) * VE;
      if (v < nv && col < C) {
#pragma unroll
        for (int e = 0; e < VE; e += 4)
          *reinterpret_cast<float4*>(&lds[sub * C + col + e]) = make_float4(acc[q][v][e], acc[q][v][e + 1], acc[q][v][e + 2], acc[q][v][e + 3]);
      }
    }
    __syncthreads();
    if (p || dst[q]) {
      for (int i = threadIdx.x; i < C; i += 256) {
        float s = 0.f;
        for (int r = 0; r < rpp; ++r) s += lds[r * C + i];
        if (p) p[q * C + i] = s;
        else unsafeAtomicAdd(dst[q] + i, s);
      }
    }
  }
  __syncthreads();
}

long row_part_floats(int B, int C) { return ((long)1024 + B) * 4 * C; }

// ================================================================================================
// modulation + ln_before                                   (reference net_trans.py:611-627)
// ================================================================================================
template <int DT, int VE, int MAXNV>
__global__ __launch_bounds__(256) void modln_fwd_k(const void* X1, const float* ch, const float* sg, const float* tg,
                                                   float alpha, float beta, float gamma, const float* lnw,
                                                   const float* lnb, float eps, int N, int C, int gs, int nv, int rpc,
                                                   void* X3, float* mu, float* rstd) {
  const int b = blockIdx.y, gl = threadIdx.x & (gs - 1), sub = threadIdx.x / gs, rpp = 256 / gs;
  const int n_end = imin_d(N, (blockIdx.x + 1) * rpc);
  float cm[MAXNV][VE], w[MAXNV][VE], bb[MAXNV][VE];
  const float tgv = tg ? gamma * tg[b] : 0.f;
#pragma unroll
  for (int v = 0; v < MAXNV; ++v) {
    const int col = (v * gs + gl) * VE;
    if (v < nv && col < C) {
      float t[VE];
      ldf<VE>(ch + (long)b * C, col, t);
#pragma unroll
      for (int e = 0; e < VE; ++e) cm[v][e] = alpha * t[e] + 1.f - alpha + tgv;
      if (lnw) { ldf<VE>(lnw, col, w[v]); ldf<VE>(lnb, col, bb[v]); }
    }
  }
  for (int n = blockIdx.x * rpc + sub; n < n_end; n += rpp) {
    const long row = (long)b * N + n;
    const float sgv = beta * sg[row];
    float x[MAXNV][VE];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < MAXNV; ++v) {
      const int col = (v * gs + gl) * VE;
      if (v < nv && col < C) {
        ldv<DT, VE>(X1, row * C + col, x[v]);
#pragma unroll
        for (int e = 0; e < VE; ++e) { x[v][e] *= (cm[v][e] + sgv); s += x[v][e]; }
      }
    }
    if (lnw) {
      const float mean = group_sum(s, gs) / C;
      float q = 0.f;
#pragma unroll
      for (int v = 0; v < MAXNV; ++v) {
        const int col = (v * gs + gl) * VE;
        if (v < nv && col < C) {
#pragma unroll
          for (int e = 0; e < VE; ++e) { const float d = x[v][e] - mean; q += d * d; }
        }
      }
      const float rs = rsqrtf(group_sum(q, gs) / C + eps);
#pragma unroll
      for (int v = 0; v < MAXNV; ++v) {
        const int col = (v * gs + gl) * VE;
        if (v < nv && col < C) {
#pragma unroll
          for (int e = 0; e < VE; ++e) x[v][e] = (x[v][e] - mean) * rs * w[v][e] + bb[v][e];
        }
      }
      if (gl == 0) { mu[row] = mean; rstd[row] = rs; }
    }
#pragma unroll
    for (int v = 0; v < MAXNV; ++v) {
      const int col = (v * gs + gl) * VE;
      if (v < nv && col < C) stv<DT, VE>(X3, row * C + col, x[v]);
    }
  }
}

void modln_fwd(const Ctx& ctx, const void* X1, const float* ch, const float* sg, const float* tg, float alpha, float beta,
               float gamma, const float* lnw, const float* lnb, float eps, int B, int N, int C, void* X3, float* mu,
               float* rstd) {
  RowGeom g = row_geom(C, row_ve(ctx, C), N, B);
  {                                                   // one full round of resident workgroups (2080 on 2048 slots = 2 rounds)
    int cap = 2048;
    ROW_CAPACITY(cap, ctx, C, g.nv, modln_fwd_k, 0);
    g = row_geom(C, row_ve(ctx, C), N, B, cap, 1, true);
  }
  ROW_DISPATCH(ctx, C, g.nv, modln_fwd_k, dim3(g.chunks, B), X1, ch, sg, tg, alpha, beta, gamma, lnw, lnb, eps, N, C, g.gs, g.nv,
               g.rpc, X3, mu, rstd);
}

template <int DT, int VE, int MAXNV>
__global__ __launch_bounds__(256) void modln_bwd_k(const void* dX3, const void* X1, const float* ch, const float* sg,
                                                   const float* tg, float alpha, float beta, float gamma,
                                                   const float* lnw, const float* mu, const float* rstd, int N, int C,
                                                   int gs, int nv, int rpc, void* dX1, float* dlnw, float* dlnb,
                                                   float* dch, float* dsg, float* dtg, float* part) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.y, gl = threadIdx.x & (gs - 1), sub = threadIdx.x / gs, rpp = 256 / gs;
  const int n_end = imin_d(N, (blockIdx.x + 1) * rpc);
  float cm[MAXNV][VE], w[MAXNV][VE];
  float acc[3][MAXNV][VE];   // 0: dlnw, 1: dlnb, 2: sum_n dX2*X1 (-> dch)
  const float tgv = tg ? gamma * tg[b] : 0.f;
#pragma unroll
  for (int v = 0; v < MAXNV; ++v) {
    const int col = (v * gs + gl) * VE;
#pragma unroll
    for (int e = 0; e < VE; ++e) { acc[0][v][e] = 0.f; acc[1][v][e] = 0.f; acc[2][v][e] = 0.f; }
    if (v < nv && col < C) {
      float t[VE];
      ldf<VE>(ch + (long)b * C, col, t);
#pragma unroll
      for (int e = 0; e < VE; ++e) cm[v][e] = alpha * t[e] + 1.f - alpha + tgv;
      if (lnw) ldf<VE>(lnw, col, w[v]);
    }
  }
  float tsum = 0.f;
  // per-row scalars travel with the row prefetch (see tail_bwd_k); two prefetch slots, as there
  auto fetch = [&](long row, uint4 (&pg)[MAXNV], uint4 (&px)[MAXNV], float& psg, float& pmean, float& prs) {
    psg = sg[row];
    if (lnw) { pmean = mu[row]; prs = rstd[row]; }
#pragma unroll
    for (int v = 0; v < MAXNV; ++v) {
      const int col = (v * gs + gl) * VE;
      if (v < nv && col < C) { pg[v] = ldraw<DT, VE>(dX3, row * C + col); px[v] = ldraw<DT, VE>(X1, row * C + col); }
    }
  };
  auto body = [&](long row, const uint4 (&pg)[MAXNV], const uint4 (&px)[MAXNV], float psg, float mean, float rs) {
    const float sgv = beta * psg;
    float g[MAXNV][VE], x1[MAXNV][VE], xh[MAXNV][VE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < MAXNV; ++v) {
      const int col = (v * gs + gl) * VE;
      if (v < nv && col < C) {
        unpack<DT, VE>(pg[v], g[v]);
        unpack<DT, VE>(px[v], x1[v]);
        if (lnw) {
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            xh[v][e] = (x1[v][e] * (cm[v][e] + sgv) - mean) * rs;
            acc[0][v][e] += g[v][e] * xh[v][e];
            acc[1][v][e] += g[v][e];
            g[v][e] *= w[v][e];
            s1 += g[v][e];
            s2 += g[v][e] * xh[v][e];
          }
        }
      }
    }
    if (lnw) { group_sum2(s1, s2, gs); s1 /= C; s2 /= C; }
    float rsum = 0.f;
#pragma unroll
    for (int v = 0; v < MAXNV; ++v) {
      const int col = (v * gs + gl) * VE;
      if (v < nv && col < C) {
        float o[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          const float dx2 = lnw ? rs * (g[v][e] - s1 - xh[v][e] * s2) : g[v][e];
          const float dm = dx2 * x1[v][e];
          acc[2][v][e] += dm;
          rsum += dm;
          o[e] = dx2 * (cm[v][e] + sgv);
        }
        stv<DT, VE>(dX1, row * C + col, o);
      }
    }
    rsum = group_sum(rsum, gs);
    if (gl == 0) { dsg[row] = beta * rsum; tsum += rsum; }
  };
  {
    uint4 pgA[MAXNV], pxA[MAXNV], pgB[MAXNV], pxB[MAXNV], cg[MAXNV], cx[MAXNV];
    float sA = 0.f, mA = 0.f, rA = 1.f, sB = 0.f, mB = 0.f, rB = 1.f;
    const long base = (long)b * N;
    int n = blockIdx.x * rpc + sub;
    if (n < n_end) fetch(base + n, pgA, pxA, sA, mA, rA);
    if (n + rpp < n_end) fetch(base + n + rpp, pgB, pxB, sB, mB, rB);
    for (; n < n_end; n += 2 * rpp) {
      float cs = sA, cm_ = mA, cr = rA;
#pragma unroll
      for (int v = 0; v < MAXNV; ++v) { cg[v] = pgA[v]; cx[v] = pxA[v]; }
      if (n + 2 * rpp < n_end) fetch(base + n + 2 * rpp, pgA, pxA, sA, mA, rA);
      body(base + n, cg, cx, cs, cm_, cr);
      if (n + rpp < n_end) {
        cs = sB; cm_ = mB; cr = rB;
#pragma unroll
        for (int v = 0; v < MAXNV; ++v) { cg[v] = pgB[v]; cx[v] = pxB[v]; }
        if (n + 3 * rpp < n_end) fetch(base + n + 3 * rpp, pgB, pxB, sB, mB, rB);
        body(base + n + rpp, cg, cx, cs, cm_, cr);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < MAXNV; ++v)
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[2][v][e] *= alpha;
  float* const dst[3] = {dlnw, dlnb, dch + (long)b * C};
  flush_cols<3, VE, MAXNV>(acc, lds, C, gs, nv, gl, dst, part);
  if (dtg) {
    const float t = block_sum(tsum, lds);
    if (threadIdx.x == 0) unsafeAtomicAdd(dtg + b, gamma * t);
  }
}

void modln_bwd(const Ctx& ctx, const void* dX3, const void* X1, const float* ch, const float* sg, const float* tg,
               float alpha, float beta, float gamma, const float* lnw, const float* mu, const float* rstd, int B, int N,
               int C, void* dX1, float* dlnw, float* dlnb, float* dch, float* dsg, float* dtg, float* part, long part_floats) {
  static const int mi = env_int("DGSCT_ROW_MIN_ITERS", 8);
  static const int use_cap = env_int("DGSCT_ROW_CAP", 1);
  RowGeom g = row_geom(C, row_ve(ctx, C), N, B, 1024, mi);
  const size_t sh = (size_t)(256 / g.gs) * C * sizeof(float);      // flush_cols: one row of C floats per row-group
  if (use_cap) {
    int cap = 1024;
    ROW_CAPACITY(cap, ctx, C, g.nv, modln_bwd_k, sh);
    g = row_geom(C, row_ve(ctx, C), N, B, cap, mi, true);
  }
  static const int use_part = env_int("DGSCT_ROW_PART", 1);
  if (!use_part || (long)g.chunks * B * 3 * C > part_floats) part = nullptr;
  ROW_DISPATCH_SH(ctx, C, g.nv, modln_bwd_k, dim3(g.chunks, B), sh, dX3, X1, ch, sg, tg, alpha, beta, gamma, lnw, mu, rstd, N, C,
                  g.gs, g.nv, g.rpc, dX1, dlnw, dlnb, dch, dsg, tg ? dtg : nullptr, part);
  if (part) {
    PartTable t; t.NQ = 3; t.C = C;
    int n = 0;
    if (dlnw) t.d[n++] = PartDesc{0, 1, g.chunks * B, 1, dlnw, 0, 1.f};
    if (dlnb) t.d[n++] = PartDesc{1, 1, g.chunks * B, 1, dlnb, 0, 1.f};
    if (dch) t.d[n++] = PartDesc{2, B, g.chunks, 1, dch, C, 1.f};
    part_reduce(ctx.stream, part, t, n);
  }
}

// ================================================================================================
// tail: BN2 affine -> ln_post / gate                        (reference net_trans.py:643,668-671)
// ================================================================================================
template <int DT, int VE, int MAXNV>
__global__ __launch_bounds__(256) void tail_fwd_k(const void* Op, const float* sc2, const float* sh2, const float* lnw,
                                                  const float* lnb, const float* gate, int gate_first, float eps,
                                                  long rows, int C, int gs, int nv, int rpc, void* out, float* mu,
                                                  float* rstd, const void* res, const BnFin fin, int has_fin) {
  const int gl = threadIdx.x & (gs - 1), sub = threadIdx.x / gs, rpp = 256 / gs;
  const long r_end = lmin_d(rows, (long)(blockIdx.x + 1) * rpc);
  const float gv = gate ? *gate : 1.f;
  float sc[MAXNV][VE], sh[MAXNV][VE], w[MAXNV][VE], bb[MAXNV][VE];
#pragma unroll
  for (int v = 0; v < MAXNV; ++v) {
    const int col = (v * gs + gl) * VE;
    if (v < nv && col < C) {
      if (has_fin) {                                   // BN2 finalised here (BnFin, prims.h): sc2 / sh2 are its output vectors
        bn_fin_vec<VE>(fin, C, col, blockIdx.x == 0 && sub == 0, sc[v], sh[v]);
      } else if (sc2) { ldf<VE>(sc2, col, sc[v]); ldf<VE>(sh2, col, sh[v]); }
      if (lnw) { ldf<VE>(lnw, col, w[v]); ldf<VE>(lnb, col, bb[v]); }
    }
  }
  for (long row = (long)blockIdx.x * rpc + sub; row < r_end; row += rpp) {
    float x[MAXNV][VE];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < MAXNV; ++v) {
      const int col = (v * gs + gl) * VE;
      if (v < nv && col < C) {
        ldv<DT, VE>(Op, row * C + col, x[v]);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          if (sc2) x[v][e] = x[v][e] * sc[v][e] + sh[v][e];
          if (gate_first) x[v][e] *= gv;
          s += x[v][e];
        }
      }
    }
    if (lnw) {
      const float mean = group_sum(s, gs) / C;
      float q = 0.f;
#pragma unroll
      for (int v = 0; v < MAXNV; ++v) {
        const int col = (v * gs + gl) * VE;
        if (v < nv && col < C) {
#pragma unroll
          for (int e = 0; e < VE; ++e) { const float d = x[v][e] - mean; q += d * d; }
        }
      }
      const float rs = rsqrtf(group_sum(q, gs) / C + eps);
#pragma unroll
      for (int v = 0; v < MAXNV; ++v) {
        const int col = (v * gs + gl) * VE;
        if (v < nv && col < C) {
#pragma unroll
          for (int e = 0; e < VE; ++e) x[v][e] = (x[v][e] - mean) * rs * w[v][e] + bb[v][e];
        }
      }
      if (gl == 0) { mu[row] = mean; rstd[row] = rs; }
    }
#pragma unroll
    for (int v = 0; v < MAXNV; ++v) {
      const int col = (v * gs + gl) * VE;
      if (v < nv && col < C) {
        if (!gate_first) {
#pragma unroll
          for (int e = 0; e < VE; ++e) x[v][e] *= gv;
        }
        if (res) {
          float rr[VE];
          ldv<DT, VE>(res, row * C + col, rr);
#pragma unroll
          for (int e = 0; e < VE; ++e) x[v][e] += rr[e];
        }
        stv<DT, VE>(out, row * C + col, x[v]);
      }
    }
  }
}

void tail_fwd(const Ctx& ctx, const void* Op, const float* sc2, const float* sh2, const float* lnw, const float* lnb,
              const float* gate, int gate_first, float eps, long rows, int C, void* out, float* mu, float* rstd,
              const void* residual, const BnFin* fin2) {
  BnFin fin{};
  int has_fin = 0;
  if (fin2 && (!bnfold_mode(-1) || rows < 1)) {
    bn_finalize(ctx, fin2->acc, fin2->rows, C, fin2->w, fin2->b, fin2->run_mean, fin2->run_var, fin2->momentum, fin2->eps,
                fin2->training, fin2->mean, fin2->rstd, fin2->sc, fin2->sh);
  } else if (fin2) {
    fin = *fin2; has_fin = 1;
  }
  RowGeom g = row_geom(C, row_ve(ctx, C), (int)rows, 1);
  {
    int cap = 2048;
    ROW_CAPACITY(cap, ctx, C, g.nv, tail_fwd_k, 0);
    g = row_geom(C, row_ve(ctx, C), (int)rows, 1, cap, 1, true);
  }
  ROW_DISPATCH(ctx, C, g.nv, tail_fwd_k, dim3(g.chunks), Op, sc2, sh2, lnw, lnb, gate, gate_first, eps, rows, C, g.gs, g.nv, g.rpc,
               out, mu, rstd, residual, fin, has_fin);
}

// AVE = the flag set of the AVE / AVVP / AVQA-visual / pretrain flavours (BN2 on, ln_post on, gate present, LN before
// gate) as compile-time constants: with runtime flags the per-element body compiled to ~70 scalar branches per row
// (hipcc does not unswitch a loop this size) in a kernel whose row iteration is latency-bound.
template <int DT, int VE, int MAXNV, bool AVE>
__device__ __forceinline__ void tail_bwd_body(const void* dOut, const void* Op, const float* sc2_, const float* sh2,
                                              const float* mean2, const float* rstd2, const float* lnw_,
                                              const float* lnb, const float* gate_, int gate_first_, const float* mu,
                                              const float* rstd, long rows, int C, int gs, int nv, int rpc, void* dO,
                                              float* dlnw, float* dlnb, float* dgate, float* bnsums, float* part, float eps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const bool sc2 = AVE || sc2_ != nullptr, lnw = AVE || lnw_ != nullptr, gate = AVE || gate_ != nullptr;
  const bool gate_first = AVE ? false : gate_first_ != 0;
  const int gl = threadIdx.x & (gs - 1), sub = threadIdx.x / gs, rpp = 256 / gs;
  const long r_end = lmin_d(rows, (long)(blockIdx.x + 1) * rpc);
  const float gv = gate ? *gate_ : 1.f;
  // d gate of out = LN(gate * O): LayerNorm is scale invariant, so sum_c dG_c O_c cancels down to its eps term; summed
  // element by element in fp32 the result is rounding noise (+-3e-4 at 2 M elements, run to run).  For gate != 0 the row
  // sum has the closed form  C eps rstd^2 mean_c(dy w xhat) / gate  (from sum_c dG = 0 and mean(xhat^2) = 1 - eps rstd^2).
  const bool gate_closed = gate_first && gate && lnw && gv != 0.f;
  const float gate_k = gate_closed ? (float)C * eps / gv : 0.f;
  float sc[MAXNV][VE], sh[MAXNV][VE], w[MAXNV][VE], bb[MAXNV][VE];
  float acc[4][MAXNV][VE];   // 0: dlnw 1: dlnb 2: sum dO 3: sum dO*Op (turned into sum dO*xh2 at the end)
#pragma unroll
  for (int v = 0; v < MAXNV; ++v) {
    const int col = (v * gs + gl) * VE;
#pragma unroll
    for (int e = 0; e < VE; ++e) { acc[0][v][e] = acc[1][v][e] = acc[2][v][e] = acc[3][v][e] = 0.f; }
    if (v < nv && col < C) {
      if (sc2) { ldf<VE>(sc2_, col, sc[v]); ldf<VE>(sh2, col, sh[v]); }
      if (lnw) { ldf<VE>(lnw_, col, w[v]); ldf<VE>(lnb, col, bb[v]); }
    }
  }
  float gsum = 0.f;
  // software pipeline: the next row's two 16-byte loads are in flight while this row is reduced and stored
  // The row statistics are fetched WITH the row: vmcnt retires in order, so a scalar load issued after the prefetch of
  // the next row would force a wait for that prefetch as well (hipcc emitted s_waitcnt vmcnt(0) right behind it).
  // Two prefetch slots: the loads of rows i+1 and i+2 are in flight while row i is reduced and stored (one slot kept
  // the kernel at ~1.9 TB/s: 32 bytes per lane in flight at 3 waves per SIMD).
  auto fetch = [&](long row, uint4 (&pg)[MAXNV], uint4 (&pop)[MAXNV], float& pmean, float& prs) {
    if (lnw) { pmean = mu[row]; prs = rstd[row]; }
#pragma unroll
    for (int v = 0; v < MAXNV; ++v) {
      const int col = (v * gs + gl) * VE;
      if (v < nv && col < C) { pg[v] = ldraw<DT, VE>(dOut, row * C + col); pop[v] = ldraw<DT, VE>(Op, row * C + col); }
    }
  };
  auto body = [&](long row, const uint4 (&pg)[MAXNV], const uint4 (&pop)[MAXNV], float mean, float rs) {
    float g[MAXNV][VE], o[MAXNV][VE], xh[MAXNV][VE], op[MAXNV][VE];
#pragma unroll
    for (int v = 0; v < MAXNV; ++v) {
      const int col = (v * gs + gl) * VE;
      if (v < nv && col < C) { unpack<DT, VE>(pg[v], g[v]); unpack<DT, VE>(pop[v], op[v]); }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < MAXNV; ++v) {
      const int col = (v * gs + gl) * VE;
      if (v < nv && col < C) {
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          o[v][e] = sc2 ? op[v][e] * sc[v][e] + sh[v][e] : op[v][e];
          if (gate_first) {
            // G = gate*O ; out = LN(G)
            if (lnw) {
              xh[v][e] = (o[v][e] * gv - mean) * rs;
              acc[0][v][e] += g[v][e] * xh[v][e];
              acc[1][v][e] += g[v][e];
              g[v][e] *= w[v][e];
              s1 += g[v][e];
              s2 += g[v][e] * xh[v][e];
            }
          } else {
            // L = LN(O) ; out = gate*L
            float L;
            if (lnw) { xh[v][e] = (o[v][e] - mean) * rs; L = xh[v][e] * w[v][e] + bb[v][e]; }
            else L = o[v][e];
            if (gate) gsum += g[v][e] * L;
            g[v][e] *= gv;                      // dL
            if (lnw) {
              acc[0][v][e] += g[v][e] * xh[v][e];
              acc[1][v][e] += g[v][e];
              g[v][e] *= w[v][e];
              s1 += g[v][e];
              s2 += g[v][e] * xh[v][e];
            }
          }
        }
      }
    }
    if (lnw) { group_sum2(s1, s2, gs); s1 /= C; s2 /= C; }
    if (gate_closed && gl == 0) gsum += gate_k * rs * rs * s2;
#pragma unroll
    for (int v = 0; v < MAXNV; ++v) {
      const int col = (v * gs + gl) * VE;
      if (v < nv && col < C) {
        float d[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          float t = lnw ? rs * (g[v][e] - s1 - xh[v][e] * s2) : g[v][e];   // dG (gate_first) or dO
          if (gate_first) {
            if (gate && !gate_closed) gsum += t * o[v][e];
            t *= gv;
          }
          d[e] = t;
          if (sc2) {
            acc[2][v][e] += t;
            acc[3][v][e] += t * op[v][e];
          }
        }
        stv<DT, VE>(dO, row * C + col, d);
      }
    }
    };
  {
    uint4 pgA[MAXNV], popA[MAXNV], pgB[MAXNV], popB[MAXNV], cg[MAXNV], cop[MAXNV];
    float mA = 0.f, rA = 1.f, mB = 0.f, rB = 1.f;
    long row = (long)blockIdx.x * rpc + sub;
    if (row < r_end) fetch(row, pgA, popA, mA, rA);
    if (row + rpp < r_end) fetch(row + rpp, pgB, popB, mB, rB);
    for (; row < r_end; row += 2 * rpp) {
      float cm = mA, cr = rA;
#pragma unroll
      for (int v = 0; v < MAXNV; ++v) { cg[v] = pgA[v]; cop[v] = popA[v]; }
      if (row + 2 * rpp < r_end) fetch(row + 2 * rpp, pgA, popA, mA, rA);
      body(row, cg, cop, cm, cr);
      if (row + rpp < r_end) {
        cm = mB; cr = rB;
#pragma unroll
        for (int v = 0; v < MAXNV; ++v) { cg[v] = pgB[v]; cop[v] = popB[v]; }
        if (row + 3 * rpp < r_end) fetch(row + 3 * rpp, pgB, popB, mB, rB);
        body(row + rpp, cg, cop, cm, cr);
      }
    }
  }
  if (sc2) {       // sum dO*xh2 = rstd2 * (sum dO*Op - mean2 * sum dO): keeps mean2/rstd2 out of the row loop's registers
#pragma unroll
    for (int v = 0; v < MAXNV; ++v) {
      const int col = (v * gs + gl) * VE;
      if (v < nv && col < C) {
        float m2[VE], r2[VE];
        ldf<VE>(mean2, col, m2); ldf<VE>(rstd2, col, r2);
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[3][v][e] = r2[e] * (acc[3][v][e] - m2[e] * acc[2][v][e]);
      }
    }
  }
  float* const dst[4] = {dlnw, dlnb, bnsums, bnsums ? bnsums + C : nullptr};
  flush_cols<4, VE, MAXNV>(acc, lds, C, gs, nv, gl, dst, part);
  if (gate) {
    const float t = block_sum(gsum, lds);
    if (threadIdx.x == 0) unsafeAtomicAdd(dgate, t);
  }
}
#define TAIL_BWD_ARGS_                                                                                                  \
  const void *dOut, const void *Op, const float *sc2, const float *sh2, const float *mean2, const float *rstd2,         \
      const float *lnw, const float *lnb, const float *gate, int gate_first, const float *mu, const float *rstd,       \
      long rows, int C, int gs, int nv, int rpc, void *dO, float *dlnw, float *dlnb, float *dgate, float *bnsums, float *part, float eps
#define TAIL_BWD_PASS_                                                                                                  \
  dOut, Op, sc2, sh2, mean2, rstd2, lnw, lnb, gate, gate_first, mu, rstd, rows, C, gs, nv, rpc, dO, dlnw, dlnb, dgate, bnsums, part, eps
template <int DT, int VE, int MAXNV>
__global__ __launch_bounds__(256, MAXNV == 1 ? 3 : 1) void tail_bwd_k(TAIL_BWD_ARGS_) {
  tail_bwd_body<DT, VE, MAXNV, false>(TAIL_BWD_PASS_);
}
template <int DT, int VE, int MAXNV>
__global__ __launch_bounds__(256, MAXNV == 1 ? 3 : 1) void tail_bwd_ave_k(TAIL_BWD_ARGS_) {
  tail_bwd_body<DT, VE, MAXNV, true>(TAIL_BWD_PASS_);
}


void tail_bwd(const Ctx& ctx, const void* dOut, const void* Op, const float* sc2, const float* sh2, const float* mean2,
              const float* rstd2, const float* lnw, const float* lnb, const float* gate, int gate_first, const float* mu,
              const float* rstd, long rows, int C, void* dO, float* dlnw, float* dlnb, float* dgate, float* bnsums,
              float eps, float* part, long part_floats) {
  static const int mi = env_int("DGSCT_ROW_MIN_ITERS", 8);
  static const int use_cap = env_int("DGSCT_ROW_CAP", 1);
  RowGeom g = row_geom(C, row_ve(ctx, C), (int)rows, 1, 1024, mi);
  const size_t sh = (size_t)(256 / g.gs) * C * sizeof(float);      // flush_cols: one row of C floats per row-group
  if (use_cap) {
    int cap = 1024;
    ROW_CAPACITY(cap, ctx, C, g.nv, tail_bwd_k, sh);
    g = row_geom(C, row_ve(ctx, C), (int)rows, 1, cap, mi, true);
  }
  static const int use_part = env_int("DGSCT_ROW_PART", 1);
  if (!use_part || (long)g.chunks * 4 * C > part_floats) part = nullptr;
  if (sc2 && lnw && gate && !gate_first)
    ROW_DISPATCH_SH(ctx, C, g.nv, tail_bwd_ave_k, dim3(g.chunks), sh, dOut, Op, sc2, sh2, mean2, rstd2, lnw, lnb, gate, gate_first,
                    mu, rstd, rows, C, g.gs, g.nv, g.rpc, dO, dlnw, dlnb, dgate, bnsums, part, eps);
  else
    ROW_DISPATCH_SH(ctx, C, g.nv, tail_bwd_k, dim3(g.chunks), sh, dOut, Op, sc2, sh2, mean2, rstd2, lnw, lnb, gate, gate_first, mu,
                    rstd, rows, C, g.gs, g.nv, g.rpc, dO, dlnw, dlnb, dgate, bnsums, part, eps);
  if (part) {
    PartTable t; t.NQ = 4; t.C = C;
    int n = 0;
    if (dlnw) t.d[n++] = PartDesc{0, 1, g.chunks, 1, dlnw, 0, 1.f};
    if (dlnb) t.d[n++] = PartDesc{1, 1, g.chunks, 1, dlnb, 0, 1.f};
    if (bnsums) { t.d[n++] = PartDesc{2, 1, g.chunks, 1, bnsums, 0, 1.f}; t.d[n++] = PartDesc{3, 1, g.chunks, 1, bnsums + C, 0, 1.f}; }
    part_reduce(ctx.stream, part, t, n);
  }
}

// ================================================================================================
// rowdot: out[b][n] = sum_c x[b][n][c] * w[b][c] * w2[c] + bias
// ================================================================================================
template <int DT, int VE, int MAXNV>
__global__ __launch_bounds__(256) void rowdot_k(const void* x, long ld, long bs, int N, int C, const void* w, int wdt,
                                                long w_bs, const float* w2, const float* bias, int gs, int nv, int rpc,
                                                float* out) {
  const int b = blockIdx.y, gl = threadIdx.x & (gs - 1), sub = threadIdx.x / gs, rpp = 256 / gs;
  const int n_end = imin_d(N, (blockIdx.x + 1) * rpc);
  float ww[MAXNV][VE];
#pragma unroll
  for (int v = 0; v < MAXNV; ++v) {
    const int col = (v * gs + gl) * VE;
#pragma unroll
    for (int e = 0; e < VE; ++e) ww[v][e] = 0.f;              // columns past C weigh nothing: the row loop loads them clamped
    if (v < nv && col < C) {
      ldv_rt<VE>(w, wdt, (long)b * w_bs + col, ww[v]);
      if (w2) {
        float t2[VE];
        ldv<DT_F32, VE>(w2, col, t2);
#pragma unroll
        for (int e = 0; e < VE; ++e) ww[v][e] *= t2[e];
      }
    }
  }
  const float bv = bias ? *bias : 0.f;
  for (int n = blockIdx.x * rpc + sub; n < n_end; n += rpp) {
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < MAXNV; ++v) {
      const int col = (v * gs + gl) * VE;
      {                                                       // no condition at all: vectors past nv / C load column 0 and weigh 0
        float t[VE];
        ldv<DT, VE>(x, (long)b * bs + (long)n * ld + ((v < nv && col < C) ? col : 0), t);
#pragma unroll
        for (int e = 0; e < VE; ++e) s += t[e] * ww[v][e];
      }
    }
    s = group_sum(s, gs);
    if (gl == 0) out[(long)b * N + n] = s + bv;
  }
}

void rowdot_batched(const Ctx& ctx, const void* x, long ld, long bs, int B, int N, int C, const void* w, int wdt, long w_bs,
                    const float* w2, const float* bias, float* out) {
  int ve = row_ve(ctx, C);
  if (ld % ve != 0 || bs % ve != 0) { set_error("rowdot_batched: unaligned ld/bs"); return; }
  RowGeom g = row_geom(C, ve, N, B);
  {
    int cap = 2048;
    ROW_CAPACITY(cap, ctx, C, g.nv, rowdot_k, 0);
    g = row_geom(C, ve, N, B, cap, 1, true);
  }
  ROW_DISPATCH(ctx, C, g.nv, rowdot_k, dim3(g.chunks, B), x, ld, bs, N, C, w, wdt, w_bs, w2, bias, g.gs, g.nv, g.rpc, out);
}

// ================================================================================================
// rowdot_colsum: the two bias-gradient reductions of the remap backward in ONE pass over dYp (plan.cpp B1)
//   out_row[n] += sum_b sum_c x[b][n][c] * w[c]            (d conv_adapter.bias;  optional)
//   out_col[c] += sum_b sum_n roww[n] * x[b][n][c]         (d rowsum(Wc) / d fc.bias of the bicubic flavour;  optional)
// They were rowdot_batched -> sum_batch and colsum_batched: two full reads of the [R, C] cotangent plus a third launch at
// the very end of every adapter backward (214 us of a 6 ms stage-0 adapter call).  A lane group owns token row n for FPG
// frames in turn, so the row dot is summed over those frames in registers before its single atomic (one per (n, frame
// group) instead of one per (n, frame): atomics on one address serialise at the memory side), and the per-channel sums
// stay in registers for the whole walk.  FPG frames' loads are issued together (independent addresses).
// ================================================================================================
template <int DT, int VE, int MAXNV>
__global__ __launch_bounds__(256) void rowdot_colsum_k(const void* x, long ld, long bs, int B, int N, int C, const float* w,
                                                       const float* roww, int gs, int nv, int rpc, int fpg, float* out_row,
                                                       float* out_col, float* part) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int gl = threadIdx.x & (gs - 1), sub = threadIdx.x / gs, rpp = 256 / gs;
  const int n_end = imin_d(N, (blockIdx.x + 1) * rpc);
  const int b0 = blockIdx.y * fpg, b1 = imin_d(B, b0 + fpg);
  float ww[MAXNV][VE], acc[1][MAXNV][VE];
#pragma unroll
  for (int v = 0; v < MAXNV; ++v) {
    const int col = (v * gs + gl) * VE;
#pragma unroll
    for (int e = 0; e < VE; ++e) { ww[v][e] = 0.f; acc[0][v][e] = 0.f; }
    if (w && v < nv && col < C) ldv<DT_F32, VE>(w, col, ww[v]);
  }
  constexpr int FU = 4;                                       // frames in flight per row
  for (int n = blockIdx.x * rpc + sub; n < n_end; n += rpp) {
    const float rw = roww ? roww[n] : 1.f;
    float s = 0.f;
    for (int bb = b0; bb < b1; bb += FU) {
      float t[FU][MAXNV][VE];
#pragma unroll
      for (int u = 0; u < FU; ++u) {
        const int bc = bb + u < b1 ? bb + u : b1 - 1;         // unconditional, clamped loads; masked below
#pragma unroll
        for (int v = 0; v < MAXNV; ++v) {
          const int col = (v * gs + gl) * VE;
          ldv<DT, VE>(x, (long)bc * bs + (long)n * ld + ((v < nv && col < C) ? col : 0), t[u][v]);
        }
      }
#pragma unroll
      for (int u = 0; u < FU; ++u) {
        const float m = bb + u < b1 ? 1.f : 0.f;
#pragma unroll
        for (int v = 0; v < MAXNV; ++v) {
          const int col = (v * gs + gl) * VE;
          const float mv = (v < nv && col < C) ? m : 0.f;
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            s += mv * t[u][v][e] * ww[v][e];
            acc[0][v][e] += mv * rw * t[u][v][e];
          }
        }
      }
    }
    if (out_row) {
      s = group_sum(s, gs);
      if (gl == 0) unsafeAtomicAdd(out_row + n, s);
    }
  }
  float* const dst[1] = {out_col};
  if (out_col) flush_cols<1, VE, MAXNV>(acc, lds, C, gs, nv, gl, dst, part);
}

// Frame-contiguous variant (round 4): a workgroup walks consecutive rows of ONE frame (the rows of a frame are contiguous in memory;
// the kernel above strides FPG frames 0.8 MB apart per row and ran at 0.84 TB/s), FU rows of a lane group in flight.  The row dots
// leave as plain stores into row_part[b][n] and are summed over the frames by sum_batch; the column sums as above.
template <int DT, int VE, int MAXNV>
__global__ __launch_bounds__(256) void rowdot_colsum_fr_k(const void* x, long ld, long bs, int N, int C, const float* w,
                                                          const float* roww, int gs, int nv, int rpc, float* row_part,
                                                          float* out_col, float* part) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int gl = threadIdx.x & (gs - 1), sub = threadIdx.x / gs, rpp = 256 / gs;
  const int b = blockIdx.y;
  const int n_end = imin_d(N, (blockIdx.x + 1) * rpc);
  float ww[MAXNV][VE], acc[1][MAXNV][VE];
#pragma unroll
  for (int v = 0; v < MAXNV; ++v) {
    const int col = (v * gs + gl) * VE;
#pragma unroll
    for (int e = 0; e < VE; ++e) { ww[v][e] = 0.f; acc[0][v][e] = 0.f; }
    if (w && v < nv && col < C) ldv<DT_F32, VE>(w, col, ww[v]);
  }
  constexpr int FU = 4;
  const char* xb = reinterpret_cast<const char*>(x) + (long)b * bs * El<DT>::ES;
  for (int n = blockIdx.x * rpc + sub; n < n_end; n += rpp * FU) {
    uint4 raw[FU][MAXNV];
    float rw[FU];
#pragma unroll
    for (int u = 0; u < FU; ++u) {
      const int nc = n + u * rpp < n_end ? n + u * rpp : n_end - 1;      // unconditional, clamped loads; masked below
      rw[u] = roww ? roww[nc] : 1.f;
#pragma unroll
      for (int v = 0; v < MAXNV; ++v) {
        const int col = (v * gs + gl) * VE;
        raw[u][v] = ldraw<DT, VE>(xb, (long)nc * ld + ((v < nv && col < C) ? col : 0));
      }
    }
#pragma unroll
    for (int u = 0; u < FU; ++u) {
      const bool ok = n + u * rpp < n_end;
      float s = 0.f;
#pragma unroll
      for (int v = 0; v < MAXNV; ++v) {
        const int col = (v * gs + gl) * VE;
        const float mv = (ok && v < nv && col < C) ? 1.f : 0.f;
        float t[VE];
        unpack<DT, VE>(raw[u][v], t);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          s += mv * t[e] * ww[v][e];
          acc[0][v][e] += mv * rw[u] * t[e];
        }
      }
      if (row_part) {
        s = group_sum(s, gs);
        if (gl == 0 && ok) row_part[(long)b * N + n + u * rpp] = s;
      }
    }
  }
  float* const dst[1] = {out_col};
  if (out_col) flush_cols<1, VE, MAXNV>(acc, lds, C, gs, nv, gl, dst, part);
}

void rowdot_colsum_frames(const Ctx& ctx, const void* x, long ld, long bs, int B, int N, int C, const float* w, const float* roww,
                          float* out_row, float* out_col, float* row_part, float* part, long part_floats) {
  int ve = row_ve(ctx, C);
  if (ld % ve != 0 || bs % ve != 0) { set_error("rowdot_colsum: unaligned ld/bs"); return; }
  if (!out_row && !out_col) return;
  RowGeom g = row_geom(C, ve, N, B);
  const size_t sh = (size_t)(256 / g.gs) * C * sizeof(float);
  {
    int cap = 2048;
    ROW_CAPACITY(cap, ctx, C, g.nv, rowdot_colsum_fr_k, sh);
    g = row_geom(C, ve, N, B, cap, 4, true);
  }
  static const int use_part = env_int("DGSCT_ROW_PART", 1);
  if (!use_part || !out_col || (long)g.chunks * B * C > part_floats) part = nullptr;
  ROW_DISPATCH_SH(ctx, C, g.nv, rowdot_colsum_fr_k, dim3(g.chunks, B), sh, x, ld, bs, N, C, w, roww, g.gs, g.nv, g.rpc,
                  out_row ? row_part : nullptr, out_col, part);
  if (out_row) sum_batch(ctx, row_part, N, B, N, out_row, 1.f, 1);
  if (part) {
    PartTable t; t.NQ = 1; t.C = C;
    t.d[0] = PartDesc{0, 1, g.chunks * B, 1, out_col, 0, 1.f};
    part_reduce(ctx.stream, part, t, 1);
  }
}

void rowdot_colsum(const Ctx& ctx, const void* x, long ld, long bs, int B, int N, int C, const float* w, const float* roww,
                   float* out_row, float* out_col, float* part, long part_floats) {
  int ve = row_ve(ctx, C);
  if (ld % ve != 0 || bs % ve != 0) { set_error("rowdot_colsum: unaligned ld/bs"); return; }
  if (!out_row && !out_col) return;
  const int fpg = B >= 8 ? 8 : B;                              // frames per lane group: 8 x fewer atomics per out_row address
  const int groups = (B + fpg - 1) / fpg;
  RowGeom g = row_geom(C, ve, N, groups);
  const size_t sh = (size_t)(256 / g.gs) * C * sizeof(float);  // flush_cols: one row of C floats per row-group
  {
    int cap = 1024;
    ROW_CAPACITY(cap, ctx, C, g.nv, rowdot_colsum_k, sh);
    g = row_geom(C, ve, N, groups, cap, 1, true);
  }
  static const int use_part = env_int("DGSCT_ROW_PART", 1);
  if (!use_part || !out_col || (long)g.chunks * groups * C > part_floats) part = nullptr;
  ROW_DISPATCH_SH(ctx, C, g.nv, rowdot_colsum_k, dim3(g.chunks, groups), sh, x, ld, bs, B, N, C, w, roww, g.gs, g.nv, g.rpc, fpg,
                  out_row, out_col, part);
  if (part) {
    PartTable t; t.NQ = 1; t.C = C;
    t.d[0] = PartDesc{0, 1, g.chunks * groups, 1, out_col, 0, 1.f};
    part_reduce(ctx.stream, part, t, 1);
  }
}

// ================================================================================================
// softmax over rows (fp32 logits in, E/fp32 probabilities out)
// ================================================================================================
// short rows (L <= 64): a group of GS lanes per row
__global__ __launch_bounds__(256) void softmax_short_k(const float* in, long ld_in, void* out, int odt, long ld_out, long rows,
                                                       int L, int gs, int pre_tanh) {
  const int gl = threadIdx.x & (gs - 1), sub = threadIdx.x / gs, rpp = 256 / gs;
  for (long r = (long)blockIdx.x * rpp + sub; r < rows; r += (long)gridDim.x * rpp) {
    float x = -INFINITY;
    if (gl < L) { x = in[r * ld_in + gl]; if (pre_tanh) x = tanhf(x); }
    const float m = group_max(x, gs);
    const float e = gl < L ? __expf(x - m) : 0.f;
    const float s = group_sum(e, gs);
    for (int c = gl; c < ld_out; c += gs) ste_rt(out, odt, r * ld_out + c, c < L ? e / s : 0.f);
  }
}
// medium rows (64 < L <= 256; the latent-token axis of num_tokens > 32, attn_wide.cpp): one wavefront per row, four elements per lane
__global__ __launch_bounds__(256) void softmax_wave_k(const float* in, long ld_in, void* out, int odt, long ld_out, long rows, int L,
                                                      int pre_tanh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (long r = (long)blockIdx.x * 4 + w; r < rows; r += (long)gridDim.x * 4) {
    float x[4], m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      x[i] = -INFINITY;
      if (c < L) { x[i] = in[r * ld_in + c]; if (pre_tanh) x[i] = tanhf(x[i]); }
      m = fmaxf(m, x[i]);
    }
    m = group_max(m, 64);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[i] = lane + 64 * i < L ? __expf(x[i] - m) : 0.f; s += x[i]; }
    s = group_sum(s, 64);
    const float inv = 1.f / s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      if (c < ld_out) ste_rt(out, odt, r * ld_out + c, x[i] * inv);
    }
  }
}
// long rows: one workgroup per row, three passes (logits are L2-resident: just written by the GEMM)
__global__ __launch_bounds__(256) void softmax_long_k(const float* in, long ld_in, void* out, int odt, long ld_out, int L,
                                                      int pre_tanh) {
  __shared__ float red[4];
  const long r = blockIdx.x;
  const float* x = in + r * ld_in;
  float m = -INFINITY;
  for (int c = threadIdx.x; c < L; c += 256) { float v = x[c]; if (pre_tanh) v = tanhf(v); m = fmaxf(m, v); }
  m = block_max(m, red);
  float s = 0.f;
  for (int c = threadIdx.x; c < L; c += 256) { float v = x[c]; if (pre_tanh) v = tanhf(v); s += __expf(v - m); }
  s = block_sum(s, red);
  const float inv = 1.f / s;
  for (int c = threadIdx.x; c < ld_out; c += 256) {
    float o = 0.f;
    if (c < L) { float v = x[c]; if (pre_tanh) v = tanhf(v); o = __expf(v - m) * inv; }
    ste_rt(out, odt, r * ld_out + c, o);
  }
}
// Row in registers: one float4 pass over the logits (the scalar version read the row three times through L2 with 4-byte
// loads and wrote 2-byte elements behind a run-time dtype branch), 8-byte bf16 / 16-byte fp32 stores.  L <= 1024 * NCH.
template <int NCH, bool OBF>
__global__ __launch_bounds__(256) void softmax_long_vec_k(const float* in, long ld_in, void* out, long ld_out, int L, int pre_tanh) {
  __shared__ float red[4];
  const long r = blockIdx.x;
  const float4* x = reinterpret_cast<const float4*>(in + r * ld_in);
  float4 v[NCH];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c4 = threadIdx.x + i * 256;
    v[i] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (c4 * 4 < L) {
      v[i] = x[c4];
      if (pre_tanh) { v[i].x = tanhf(v[i].x); v[i].y = tanhf(v[i].y); v[i].z = tanhf(v[i].z); v[i].w = tanhf(v[i].w); }
    }
    m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
  }
  m = block_max(m, red);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    v[i].x = __expf(v[i].x - m); v[i].y = __expf(v[i].y - m); v[i].z = __expf(v[i].z - m); v[i].w = __expf(v[i].w - m);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  s = block_sum(s, red);
  const float inv = 1.f / s;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c4 = threadIdx.x + i * 256;
    if (c4 * 4 >= ld_out) continue;
    const float4 o = make_float4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);   // exp(-inf) = 0 beyond L
    if (OBF) {
      uint2 w;
      w.x = f2bf2(o.x, o.y);
      w.y = f2bf2(o.z, o.w);
      reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(out) + r * ld_out)[c4] = w;
    } else {
      reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + r * ld_out)[c4] = o;
    }
  }
}
// backward, bf16 P and output: out = s * P * (dP - sum P*dP)
template <int NCH>
__global__ __launch_bounds__(256) void softmax_bwd_long_vec_k(const void* P, long ldp, const float* dP, long lddp, void* out, long ldo,
                                                              int L, const float* scale_ptr, float* dot_accum) {
  __shared__ float red[4];
  const long r = blockIdx.x;
  const float sc = scale_ptr ? *scale_ptr : 1.f;
  const uint2* pp = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(P) + r * ldp);
  const float4* gp = reinterpret_cast<const float4*>(dP + r * lddp);
  float4 p[NCH], g[NCH];
  float pd = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c4 = threadIdx.x + i * 256;
    p[i] = make_float4(0.f, 0.f, 0.f, 0.f); g[i] = p[i];
    if (c4 * 4 < L) {
      const uint2 w = pp[c4];
      g[i] = gp[c4];
      p[i] = make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                         __uint_as_float(w.y & 0xffff0000u));
    }
    pd += (p[i].x * g[i].x + p[i].y * g[i].y) + (p[i].z * g[i].z + p[i].w * g[i].w);
  }
  pd = block_sum(pd, red);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c4 = threadIdx.x + i * 256;
    if (c4 * 4 >= ldo) continue;
    uint2 w;
    w.x = f2bf2(sc * p[i].x * (g[i].x - pd), sc * p[i].y * (g[i].y - pd));
    w.y = f2bf2(sc * p[i].z * (g[i].z - pd), sc * p[i].w * (g[i].w - pd));
    reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(out) + r * ldo)[c4] = w;
  }
  if (dot_accum && threadIdx.x == 0) unsafeAtomicAdd(dot_accum, pd);
}
static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline bool al8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }
static inline int pow2ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }
static inline int flat_grid(long nvec) { long g = cdiv(nvec, 256); return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

void softmax_rows(const Ctx& ctx, const float* in, long ld_in, void* out, int odt, long ld_out, long rows, int L, int pre_tanh) {
  if (rows <= 0) return;
  if (L <= 64) {
    const int gs = imax(pow2ceil(L), 1);
    const long nb = cdiv(rows, 256 / gs);
    hipLaunchKernelGGL(softmax_short_k, dim3((int)(nb > 8192 ? 8192 : nb)), dim3(256), 0, STREAM(ctx), in, ld_in, out, odt, ld_out,
                       rows, L, gs, pre_tanh);
  } else if (L <= 256 && ld_out <= 256) {
    const long nb = cdiv(rows, 4);
    hipLaunchKernelGGL(softmax_wave_k, dim3((int)(nb > 16384 ? 16384 : nb)), dim3(256), 0, STREAM(ctx), in, ld_in, out, odt, ld_out,
                       rows, L, pre_tanh);
  } else if (L % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0 && ld_out <= 4096 && al16(in) && al16(out)) {
    const int nch = (int)cdiv(ld_out, 1024);
#define SMV_(N_)                                                                                                        \
  do {                                                                                                                  \
    if (odt == DT_BF16) hipLaunchKernelGGL((softmax_long_vec_k<N_, true>), dim3((int)rows), dim3(256), 0, STREAM(ctx), in, ld_in, out, ld_out, L, pre_tanh); \
    else hipLaunchKernelGGL((softmax_long_vec_k<N_, false>), dim3((int)rows), dim3(256), 0, STREAM(ctx), in, ld_in, out, ld_out, L, pre_tanh); \
  } while (0)
    if (nch <= 1) SMV_(1); else if (nch == 2) SMV_(2); else if (nch == 3) SMV_(3); else SMV_(4);
#undef SMV_
  } else {
    hipLaunchKernelGGL(softmax_long_k, dim3((int)rows), dim3(256), 0, STREAM(ctx), in, ld_in, out, odt, ld_out, L, pre_tanh);
  }
}

__global__ __launch_bounds__(256) void softmax_bwd_short_k(const void* P, int pdt, long ldp, const float* dP, long lddp, void* out,
                                                           int odt, long ldo, long rows, int L, int gs,
                                                           const float* scale_ptr, float* dot_accum) {
  __shared__ float red[4];
  const int gl = threadIdx.x & (gs - 1), sub = threadIdx.x / gs, rpp = 256 / gs;
  const float sc = scale_ptr ? *scale_ptr : 1.f;
  float dacc = 0.f;
  for (long r = (long)blockIdx.x * rpp + sub; r < rows; r += (long)gridDim.x * rpp) {
    float p = 0.f, g = 0.f;
    if (gl < L) { p = lde_rt(P, pdt, r * ldp + gl); g = dP[r * lddp + gl]; }
    const float pd = group_sum(p * g, gs);
    if (gl == 0) dacc += pd;
    for (int c = gl; c < ldo; c += gs) ste_rt(out, odt, r * ldo + c, c < L ? sc * p * (g - pd) : 0.f);
  }
  if (dot_accum) {
    const float t = block_sum(dacc, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(dot_accum, t);
  }
}
// medium rows (64 < L <= 256): one wavefront per row; the dot accumulator gets ONE atomic per workgroup (a row each on one address --
// 368 640 rows at stage 0 -- serialised for 12 ms)
__global__ __launch_bounds__(256) void softmax_bwd_wave_k(const void* P, int pdt, long ldp, const float* dP, long lddp, void* out, int odt,
                                                          long ldo, long rows, int L, const float* scale_ptr, float* dot_accum) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float sc = scale_ptr ? *scale_ptr : 1.f;
  float dacc = 0.f;
  for (long r = (long)blockIdx.x * 4 + w; r < rows; r += (long)gridDim.x * 4) {
    float p[4], g[4], pd = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      p[i] = 0.f; g[i] = 0.f;
      if (c < L) { p[i] = lde_rt(P, pdt, r * ldp + c); g[i] = dP[r * lddp + c]; }
      pd += p[i] * g[i];
    }
    pd = group_sum(pd, 64);
    if (lane == 0) dacc += pd;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      if (c < ldo) ste_rt(out, odt, r * ldo + c, sc * p[i] * (g[i] - pd));
    }
  }
  if (dot_accum) {
    const float t = block_sum(dacc, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(dot_accum, t);
  }
}
__global__ __launch_bounds__(256) void softmax_bwd_long_k(const void* P, int pdt, long ldp, const float* dP, long lddp, void* out,
                                                          int odt, long ldo, int L, const float* scale_ptr, float* dot_accum) {
  __shared__ float red[4];
  const long r = blockIdx.x;
  const float sc = scale_ptr ? *scale_ptr : 1.f;
  float pd = 0.f;
  for (int c = threadIdx.x; c < L; c += 256) pd += lde_rt(P, pdt, r * ldp + c) * dP[r * lddp + c];
  pd = block_sum(pd, red);
  for (int c = threadIdx.x; c < ldo; c += 256) {
    float o = 0.f;
    if (c < L) o = sc * lde_rt(P, pdt, r * ldp + c) * (dP[r * lddp + c] - pd);
    ste_rt(out, odt, r * ldo + c, o);
  }
  if (dot_accum && threadIdx.x == 0) unsafeAtomicAdd(dot_accum, pd);
}

void softmax_bwd_rows(const Ctx& ctx, const void* P, long ldp, const float* dP, long lddp, void* out, int odt, long ldo,
                      long rows, int L, const float* scale_ptr, float* dot_accum) {
  if (rows <= 0) return;
  const int pdt = ctx.mode;
  if (L <= 64) {
    const int gs = imax(pow2ceil(L), 1);
    const long nb = cdiv(rows, 256 / gs);
    hipLaunchKernelGGL(softmax_bwd_short_k, dim3((int)(nb > 2048 ? 2048 : nb)), dim3(256), 0, STREAM(ctx), P, pdt, ldp, dP, lddp,
                       out, odt, ldo, rows, L, gs, scale_ptr, dot_accum);
  } else if (L <= 256 && ldo <= 256) {
    const long nb = cdiv(rows, 4);
    hipLaunchKernelGGL(softmax_bwd_wave_k, dim3((int)(nb > 4096 ? 4096 : nb)), dim3(256), 0, STREAM(ctx), P, pdt, ldp, dP, lddp, out, odt,
                       ldo, rows, L, scale_ptr, dot_accum);
  } else if (pdt == DT_BF16 && odt == DT_BF16 && L % 4 == 0 && ldp % 4 == 0 && lddp % 4 == 0 && ldo % 4 == 0 && ldo <= 4096 &&
             al8(P) && al16(dP) && al8(out)) {
    const int nch = (int)cdiv(ldo, 1024);
#define SMB_(N_) hipLaunchKernelGGL((softmax_bwd_long_vec_k<N_>), dim3((int)rows), dim3(256), 0, STREAM(ctx), P, ldp, dP, lddp, out, ldo, L, scale_ptr, dot_accum)
    if (nch <= 1) SMB_(1); else if (nch == 2) SMB_(2); else if (nch == 3) SMB_(3); else SMB_(4);
#undef SMB_
  } else {
    hipLaunchKernelGGL(softmax_bwd_long_k, dim3((int)rows), dim3(256), 0, STREAM(ctx), P, pdt, ldp, dP, lddp, out, odt, ldo, L,
                       scale_ptr, dot_accum);
  }
}

// ================================================================================================
// spatial gate                                                 (reference net_trans.py:604-608)
// ================================================================================================
__global__ __launch_bounds__(256) void spatial_fwd_k(const float* sl, int N, float* sg, float* map, float* map2) {
  __shared__ float red[4];
  const long o = (long)blockIdx.x * N;
  float m = -INFINITY;
  for (int n = threadIdx.x; n < N; n += 256) m = fmaxf(m, tanhf(sl[o + n]));
  m = block_max(m, red);
  float s = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) s += __expf(tanhf(sl[o + n]) - m);
  s = block_sum(s, red);
  const float inv = 1.f / s;
  for (int n = threadIdx.x; n < N; n += 256) {
    const float v = sl[o + n];
    sg[o + n] = sigmoidf_(v);
    const float mv = __expf(tanhf(v) - m) * inv;
    map[o + n] = mv;
    if (map2) map2[o + n] = mv;
  }
}
void spatial_fwd(const Ctx& ctx, const float* sl, int B, int N, float* sg, float* map, float* map2) {
  hipLaunchKernelGGL(spatial_fwd_k, dim3(B), dim3(256), 0, STREAM(ctx), sl, N, sg, map, map2);
}
__global__ __launch_bounds__(256) void spatial_bwd_k(const float* sl, const float* sg, const float* map, const float* dsg,
                                                     const float* dMap, int N, float* dsl, float* dbs) {
  __shared__ float red[4];
  const long o = (long)blockIdx.x * N;
  float pd = 0.f;
  if (dMap) {
    for (int n = threadIdx.x; n < N; n += 256) pd += map[o + n] * dMap[o + n];
    pd = block_sum(pd, red);
  }
  float acc = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) {
    const float s = sg[o + n];
    float d = dsg[o + n] * s * (1.f - s);
    if (dMap) {
      const float t = tanhf(sl[o + n]);
      d += map[o + n] * (dMap[o + n] - pd) * (1.f - t * t);
    }
    dsl[o + n] = d;
    acc += d;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) unsafeAtomicAdd(dbs, acc);
}
void spatial_bwd(const Ctx& ctx, const float* sl, const float* sg, const float* map, const float* dsg, const float* dMap,
                 int B, int N, float* dsl, float* dbs) {
  hipLaunchKernelGGL(spatial_bwd_k, dim3(B), dim3(256), 0, STREAM(ctx), sl, sg, map, dsg, dMap, N, dsl, dbs);
}

// ================================================================================================
// small helpers
// ================================================================================================
__device__ __forceinline__ void ew_eval(const EwCall& p, long i) {
  float r;
  switch (p.op) {
    case EW_MUL: r = lde_rt(p.a.p, p.a.dt, i) * lde_rt(p.b.p, p.b.dt, i); break;
    case EW_MUL_MASK: r = lde_rt(p.c.p, p.c.dt, i) > 0.f ? lde_rt(p.a.p, p.a.dt, i) * lde_rt(p.b.p, p.b.dt, i) : 0.f; break;
    case EW_SIGMOID_BWD: { const float y = lde_rt(p.b.p, p.b.dt, i); r = lde_rt(p.a.p, p.a.dt, i) * y * (1.f - y); break; }
    case EW_SCALE: r = p.s * lde_rt(p.a.p, p.a.dt, i); break;
    case EW_ADD_BCAST: r = lde_rt(p.a.p, p.a.dt, i) + p.s * lde_rt(p.b.p, p.b.dt, i / p.div); break;
    case EW_MULB_MASK: r = lde_rt(p.c.p, p.c.dt, i) > 0.f ? lde_rt(p.a.p, p.a.dt, i) * lde_rt(p.b.p, p.b.dt, i % p.div) : 0.f; break;
    case EW_RND_MUL: {
      float t = p.s * lde_rt(p.a.p, p.a.dt, i);
      if (p.c.dt == DT_BF16) t = bf2f(f2bf(t));
      r = t * lde_rt(p.b.p, p.b.dt, i);
      break;
    }
    case EW_MUL3B: r = lde_rt(p.a.p, p.a.dt, i) * lde_rt(p.b.p, p.b.dt, i) * lde_rt(p.c.p, p.c.dt, i % p.div); break;
    case EW_OUTER_ACC: r = lde_rt(p.o, p.odt, i) + lde_rt(p.a.p, p.a.dt, i / p.div) * lde_rt(p.b.p, p.b.dt, i % p.div); break;
    default: r = lde_rt(p.a.p, p.a.dt, i); break;
  }
  ste_rt(p.o, p.odt, i, r);
}
__global__ void ew_k(const EwCall p) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < p.n) ew_eval(p, i);
}
// two independent element-wise ops in one launch (the gate-MLP backward has two pairs with shared inputs)
__global__ void ew2_k(const EwCall p, const EwCall q, int first_q) {
  if ((int)blockIdx.x < first_q) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p.n) ew_eval(p, i);
  } else {
    const long i = (long)((int)blockIdx.x - first_q) * blockDim.x + threadIdx.x;
    if (i < q.n) ew_eval(q, i);
  }
}
void ew(const Ctx& ctx, int op, void* o, int odt, EwArg a, EwArg b, EwArg c, long n, float s, long div) {
  if (n <= 0) return;
  const EwCall p{op, o, odt, a, b, c, n, s, div < 1 ? 1 : div};
  hipLaunchKernelGGL(ew_k, dim3((int)cdiv(n, 256)), dim3(256), 0, STREAM(ctx), p);
}
void ew2(const Ctx& ctx, EwCall p, EwCall q) {
  if (p.n <= 0 || q.n <= 0) { if (p.n > 0) ew(ctx, p.op, p.o, p.odt, p.a, p.b, p.c, p.n, p.s, p.div); if (q.n > 0) ew(ctx, q.op, q.o, q.odt, q.a, q.b, q.c, q.n, q.s, q.div); return; }
  if (p.div < 1) p.div = 1;
  if (q.div < 1) q.div = 1;
  const int fq = (int)cdiv(p.n, 256);
  hipLaunchKernelGGL(ew2_k, dim3(fq + (int)cdiv(q.n, 256)), dim3(256), 0, STREAM(ctx), p, q, fq);
}

__global__ __launch_bounds__(64) void temporal_fwd_k(const float* a, const float* wt, const float* bt, int C, float* tg) {
  const int b = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 64) s += a[(long)b * C + c] * wt[c];
  s = group_sum(s, 64);
  if (threadIdx.x == 0) tg[b] = sigmoidf_(s + bt[0]);
}
void temporal_fwd(const Ctx& ctx, const float* a, const float* wt, const float* bt, int B, int C, float* tg) {
  hipLaunchKernelGGL(temporal_fwd_k, dim3(B), dim3(64), 0, STREAM(ctx), a, wt, bt, C, tg);
}

__global__ void cvt_k(const float* in, void* out, int odt, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) ste_rt(out, odt, i, in[i]);
}
void cvt(const Ctx& ctx, const float* in, void* out, int odt, long n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(cvt_k, dim3(flat_grid(n)), dim3(256), 0, STREAM(ctx), in, out, odt, n);
}

struct CvtTable { CvtSeg seg[CVT_MAX_SEG]; long first_block[CVT_MAX_SEG + 1]; int nseg; };
__global__ __launch_bounds__(256) void cvt_multi_k(const CvtTable t) {
  __shared__ float tile[64][65];
  int s = 0;
  while (s + 1 < t.nseg && (long)blockIdx.x >= t.first_block[s + 1]) ++s;
  const CvtSeg sg = t.seg[s];
  const long blk = (long)blockIdx.x - t.first_block[s];
  if (sg.tr_cols > 0) {
    // dst[j][r] = src[r][j] in 64 x 64 tiles through LDS: rows of src are read, rows of dst are written (a thread-per-
    // destination-element version read with a stride of tr_cols floats: 3.8 GB fetched per step for 0.3 GB of weights)
    const long cols = sg.tr_cols, rows = sg.n / cols, tcols = (cols + 63) / 64;
    const long r0 = (blk / tcols) * 64, c0 = (blk % tcols) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    {   // 16 loads in flight (clamped, unconditional: inside the bounds test each was a serialised round trip)
      float v[16];
      const long cc = c0 + tx < cols ? c0 + tx : cols - 1;
#pragma unroll
      for (int i = 0; i < 16; ++i) { const long rr = r0 + ty + 4 * i < rows ? r0 + ty + 4 * i : rows - 1; v[i] = sg.src[rr * cols + cc]; }
#pragma unroll
      for (int i = 0; i < 16; ++i) tile[ty + 4 * i][tx] = v[i];
    }
    __syncthreads();
    for (int k = ty; k < 64; k += 4)
      if (c0 + k < cols && r0 + tx < rows) ste_rt(sg.dst, sg.odt, (c0 + k) * rows + r0 + tx, tile[tx][k]);
    return;
  }
  const long base = blk * 2048;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { const long i = base + k * 256 + threadIdx.x; v[k] = sg.src[i < sg.n ? i : sg.n - 1]; }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const long i = base + k * 256 + threadIdx.x;
    if (i < sg.n) ste_rt(sg.dst, sg.odt, i, v[k]);
  }
}
void cvt_multi(const Ctx& ctx, const CvtSeg* segs, int nseg) {
  if (nseg <= 0) return;
  if (nseg > CVT_MAX_SEG) { set_error("cvt_multi: too many segments"); return; }
  CvtTable t;
  long blocks = 0;
  t.nseg = nseg;
  for (int s = 0; s < nseg; ++s) {
    t.seg[s] = segs[s]; t.first_block[s] = blocks;
    if (segs[s].tr_cols > 0) blocks += cdiv(segs[s].n / segs[s].tr_cols, 64) * cdiv(segs[s].tr_cols, 64);
    else blocks += cdiv(segs[s].n, 2048);
  }
  t.first_block[nseg] = blocks;
  if (blocks == 0) return;
  hipLaunchKernelGGL(cvt_multi_k, dim3((int)blocks), dim3(256), 0, STREAM(ctx), t);
}

// hi = bf16(src), lo = bf16(src - hi): the fp32 latent tokens as the operand pair of the two-launch logit products (attn_wide.cpp)
__global__ __launch_bounds__(256) void split_hilo_k(const float* __restrict__ src, long n, unsigned short* __restrict__ hi,
                                                    unsigned short* __restrict__ lo) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = src[i];
    const unsigned short h = f2bf(v);
    hi[i] = h;
    lo[i] = f2bf(v - __uint_as_float((unsigned)h << 16));
  }
}
void split_hilo(const Ctx& ctx, const float* src, long n, void* hi, void* lo) {
  if (n <= 0) return;
  hipLaunchKernelGGL(split_hilo_k, dim3(flat_grid(n)), dim3(256), 0, STREAM(ctx), src, n, (unsigned short*)hi, (unsigned short*)lo);
}

struct ColsumTable { ColsumSeg seg[COLSUM_MAX_SEG]; int first_block[COLSUM_MAX_SEG + 1]; int nseg; };
// thread -> one column (coalesced row reads), 4 row slices per workgroup combined in LDS
__global__ __launch_bounds__(256) void colsum_multi_k(const ColsumTable t) {
  __shared__ float red[4][64];
  int s = 0;
  while (s + 1 < t.nseg && (int)blockIdx.x >= t.first_block[s + 1]) ++s;
  const ColsumSeg sg = t.seg[s];
  const int c = ((int)blockIdx.x - t.first_block[s]) * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  // rows in slabs of 64 per blockIdx.y: 16 rows per thread as 4 independent loads per trip (256-row slabs with two
  // accumulators were one 32-deep dependent chain per thread: 14.5 us for the 5 bias gradients of an adapter)
  const int r_lo = (int)blockIdx.y * 64, r_hi = r_lo + 64 < sg.rows ? r_lo + 64 : sg.rows;
  if (r_lo >= sg.rows) return;                       // (uniform per workgroup)
  float a0 = 0.f, a1 = 0.f;
  if (c < sg.C) {
    if (sg.dt == DT_F32) {
      const float* x = reinterpret_cast<const float*>(sg.x) + c;
      int r = r_lo + q;
      for (; r + 12 < r_hi; r += 16) {
        const float v0 = x[(long)r * sg.C], v1 = x[(long)(r + 4) * sg.C], v2 = x[(long)(r + 8) * sg.C], v3 = x[(long)(r + 12) * sg.C];
        a0 += v0 + v2; a1 += v1 + v3;
      }
      for (; r < r_hi; r += 4) a0 += x[(long)r * sg.C];
    } else {
      int r = r_lo + q;
      for (; r + 4 < r_hi; r += 8) { a0 += lde_rt(sg.x, sg.dt, (long)r * sg.C + c); a1 += lde_rt(sg.x, sg.dt, (long)(r + 4) * sg.C + c); }
      if (r < r_hi) a0 += lde_rt(sg.x, sg.dt, (long)r * sg.C + c);
    }
  }
  red[q][threadIdx.x & 63] = a0 + a1;
  __syncthreads();
  if (q == 0 && c < sg.C)     // atomic: several segments (and the row slabs of one) may accumulate into the same output
    unsafeAtomicAdd(sg.out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
void colsum_multi(const Ctx& ctx, const ColsumSeg* segs, int nseg) {
  if (nseg <= 0) return;
  if (nseg > COLSUM_MAX_SEG) { set_error("colsum_multi: too many segments"); return; }
  ColsumTable t;
  int blocks = 0;
  t.nseg = nseg;
  for (int s = 0; s < nseg; ++s) { t.seg[s] = segs[s]; t.first_block[s] = blocks; blocks += (segs[s].C + 63) / 64; }
  t.first_block[nseg] = blocks;
  if (blocks == 0) return;
  int maxrows = 1;
  for (int s = 0; s < nseg; ++s) maxrows = segs[s].rows > maxrows ? segs[s].rows : maxrows;
  hipLaunchKernelGGL(colsum_multi_k, dim3(blocks, (maxrows + 63) / 64), dim3(256), 0, STREAM(ctx), t);
}

__global__ __launch_bounds__(64) void rowsum_f32_k(const float* W, int C, float* out) {
  const int r = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 64) s += W[(long)r * C + c];
  s = group_sum(s, 64);
  if (threadIdx.x == 0) out[r] = s;
}
void rowsum_f32(const Ctx& ctx, const float* W, int R, int C, float* out) {
  hipLaunchKernelGGL(rowsum_f32_k, dim3(R), dim3(64), 0, STREAM(ctx), W, C, out);
}

}  // namespace dgsct
