// Primitive operations of the DG-SCT adapter path.
//
// plan.cpp (the kernel schedule of one adapter forward / backward) is written against this
// interface only.  The product library links it with prims_hip.hip + gemm.hip (hand-written gfx950
// kernels).  tests/emu/prims_host.cpp implements the same interface with plain host loops so that
// the schedule itself (offsets, strides, operand roles) can be checked against the oracle in the
// CPU-only container; that object is TEST INFRASTRUCTURE and is never linked into libdgsct.so.
//
// Conventions: "E" is the storage/MFMA-operand dtype of the call (ctx.mode: F32 or BF16), "F" is
// fp32.  All reductions and transcendental math are fp32.  Every pointer is a device pointer (host
// pointer in the emulation), nothing is allocated or freed here, every launch goes to ctx.stream.
#pragma once
#include <cstddef>
#include <cstdint>

namespace dgsct {

enum DType : int { DT_F32 = 0, DT_BF16 = 1 };
inline size_t dt_size(int dt) { return dt == DT_F32 ? 4 : 2; }

struct Ctx {
  void* stream;   // hipStream_t
  int mode;       // DType of E
  void* aux = nullptr;   // optional second hipStream_t for work off the critical path (weight gradients)
  struct PartJob* late = nullptr;   // when set: a primitive whose second-stage reduction feeds PARAMETER gradients only records it here
                                    // instead of launching it (the plan releases it on the aux stream with the weight gradients)
};
// second stage of the per-channel partial sums (device_util.h: part_reduce_k): dst[j][c] += scale * sum_{k<K} part[j*K + k][q][c]
struct PartDesc { int q, J, K, S; float* dst; long dst_stride; float scale; };
struct PartTable { PartDesc d[4]; int NQ, C; };
struct PartJob { const float* part = nullptr; PartTable t; int n = 0; const float* part2 = nullptr; PartTable t2; int n2 = 0; };   // (t2: rows of another width)
void part_reduce_run(void* stream, const PartJob&);
// diagnostics: event pair around every adapter call on its stream (dgsct_test_tune "callprof": 1 on, 0 off, 2 dump to $DGSCT_CALL_PROF)
int call_prof_mode(int set);
void* call_prof_begin(void* stream, int kind, int N, int C);
void call_prof_end(void* rec);
void call_prof_dump(const char* path);
// aux waits for everything enqueued on stream so far / stream waits for everything enqueued on aux so far.
// No-ops when ctx.aux is null.
// a stream of its own priority class (-1 high, 0 normal, +1 low): the HIP runtime keeps one hardware-queue pool per
// priority, so streams of different classes never share (and serialise on) a hardware queue
void* stream_create(int priority_class);
void stream_destroy(void* stream);
void stream_fork(const Ctx&);
void stream_join(const Ctx&);
void event_record(const Ctx&, void* ev);    // hipEventRecord(ev, ctx.stream) -- the caller's hipEvent_t (pair backward, plan.cpp)
void event_wait(const Ctx&, void* ev);      // hipStreamWaitEvent(ctx.stream, ev)
// report (through set_error) the first failed kernel launch / runtime call of this thread since the last check
void check_async(const char* where);
// drop a stale sticky runtime error of this thread (left by another HIP user) before a call starts
void clear_async();

// One GEMM operand X[r][k] (r = M- or N-index, k = contraction index), element type E.
//   kmajor = 1 : &X[r][k] = p + r*ld + k     (k contiguous)
//   kmajor = 0 : &X[r][k] = p + k*ld + r     (r contiguous; staged to LDS and transpose-read)
// Batched by bs (0 = shared by all batches).  Two-level contraction: k = kb*K + ki, kb < KB adds kb*kbs.
struct MatOp {
  const void* p = nullptr;
  long ld = 0;
  int kmajor = 1;
  long bs = 0;
  long kbs = 0;
};

// ACT_SOFTMAX / ACT_SOFTMAX_BWD: fused epilogues of the X <- latent-token attention (net_trans.py:583-589).  The GEMM
// computes logits^T (M = latent tokens <= 32, N = tokens) and the epilogue works down each COLUMN and writes the
// result TRANSPOSED, D[b][n*ldd + m]:  softmax_m(acc)   /   s * P * (acc - sum_m P*acc)  with P[b][n*ldmask + m] = `mask`
// (dtype E), s = *sm_scale (or 1), and *sm_dot += sum_{m,n} P*acc (the gate_av gradient).
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2, ACT_SOFTMAX = 3, ACT_SOFTMAX_BWD = 4 };

// D[b][m][n] = epi( sum_{kb<KB} sum_{k<K} A[b][m][(kb,k)] * B[b][n][(kb,k)] )
//   v  = alpha * (alpha_ptr ? *alpha_ptr : 1) * acc
//        + bias_m[m % m_mod] + bias_n[b*bias_n_bs + n] + r1_m[m % m_mod] * r1_n[n]
//   v  = act(v);  if (mask) v *= (mask[b][m][n] > 0);  if (R) v += beta * R[b][m][n]
//   atomic ? atomicAdd(D, v) (D fp32, pre-zeroed, used with splitk > 1) : D = v
struct Gemm {
  int M = 0, N = 0, K = 0, KB = 1, batch = 1, splitk = 1, atomic = 0;
  int sole_writer = 0;       // with atomic: D is pre-zeroed and nothing else accumulates into it, so an UNSPLIT product may store it plainly
  MatOp A, B;
  void* D = nullptr; int ddt = DT_F32; long ldd = 0, dbs = 0;
  float alpha = 1.f; const float* alpha_ptr = nullptr;
  const float* bias_m = nullptr; const float* bias_n = nullptr; long bias_n_bs = 0; int m_mod = 0;
  const float* r1_m = nullptr; const float* r1_n = nullptr;
  int act = ACT_NONE;
  const void* R = nullptr; int rdt = DT_F32; long ldr = 0, rbs = 0; float beta = 1.f;
  const void* R2 = nullptr;                                  // second residual, same dtype / ld / batch stride as R, weight 1
  const void* mask = nullptr; long ldmask = 0, maskbs = 0;   // dtype E
  const float* sm_scale = nullptr; float* sm_dot = nullptr;  // ACT_SOFTMAX_BWD only
};

void gemm(const Ctx&, const Gemm&);

// ---- producer / consumer fusion hooks of the tiled engine (gemm_fx.hip; round 5) ----------------------------------------------------
// A GEMM of the adapter's BACKWARD chain with the elementwise launch in front of it folded into the staging of its A operand and the
// one behind it into its epilogue (bf16, both operands K-major, E output).  `rpf` = token rows per frame: row m belongs to frame m / rpf.
//   a_pro = APRO_MASKSCALE (relu_bwd_scale, net_trans.py:594 / 602 autograd):
//       A'[m][k] = A[m][k] > 0 ? E( (a_rs ? a_rs[m] : 1) * a_scale * a_cs[frame][k] * (a_cs2 ? a_cs2[k] : 1) ) : 0
//   a_pro = APRO_BNBWD (bn_bwd_apply, net_trans.py:636 / 643 autograd; batch = conv groups, channel c = b * K + k, x = a2 laid out like A):
//       gg = bn_relu && !(x * sc[c] + sh[c] > 0) ? 0 : A;   A' = E( k1 gg - k2 - x k3 ),  k1 = sc,  k3 = sc rstd sums[C + c] / rows,
//       k2 = sc sums[c] / rows - mean k3   (eval mode: k2 = k3 = 0)
//   a_store (optional): A' is written once, laid out like A (may BE A when every A element feeds one output tile only: N <= the tile width)
//   epi = EPI_XCBWD (xc_bwd, net_trans.py:598 autograd):   v = E(acc);  D = R + v * (1 + e_cs[frame][n]);  e_acc[frame][n] += sum_m v * e_x[m][n]
//   (e_x: E [M][ldd] like D;  e_cs / e_acc: fp32 [frames][e_ld])
//   epi = EPI_COLSTATS (bn_stats of the output, net_trans.py:636 / 643):  v = E(act(acc + bias));  D = v;  channel c = b * N + n (b: batch = group):
//       e_acc[c] += sum_m v,  e_acc2[c] += sum_m v^2     (the BatchNorm sums with shift 0: bn_stats' accumulators [shift | sum | sum of squares])
//   epi = EPI_COLSUM (colsum_batched_pos of the output, net_trans.py:594-595):  v = E(act(acc + bias));  D = v;
//       e_acc[frame][n] += e_scale * sum_m v,  e_acc2[frame][n] += #{m: v > 0}
enum APro : int { APRO_NONE = 0, APRO_MASKSCALE = 1, APRO_BNBWD = 2 };
enum FxEpi : int { EPI_NONE = 0, EPI_XCBWD = 1, EPI_COLSTATS = 2, EPI_COLSUM = 3 };
struct GemmFx {
  int a_pro = APRO_NONE, epi = EPI_NONE, rpf = 0;
  const float* a_rs = nullptr; const void* a_cs = nullptr; int a_cs_dt = DT_F32; long a_cs_ld = 0; const float* a_cs2 = nullptr; float a_scale = 1.f;
  const void* a2 = nullptr; const float* bn_mean = nullptr; const float* bn_rstd = nullptr; const float* bn_sc = nullptr; const float* bn_sh = nullptr;
  const float* bn_sums = nullptr; long bn_rows = 0; int bn_C = 0, bn_relu = 0, bn_training = 1;
  void* a_store = nullptr;
  const float* e_cs = nullptr; const void* e_x = nullptr; float* e_acc = nullptr; long e_ld = 0;
  float* e_acc2 = nullptr; float e_scale = 1.f;
};
// plain D = A' B^T (+ bias_n, + R) in E; g carries the operands / output as for gemm()
bool gemm_fx_supported(const Ctx&, const Gemm&, const GemmFx&);
void gemm_fx(const Ctx&, const Gemm&, const GemmFx&);
int gemmfx_mode(int set);          // dgsct_test_tune "gemmfx": bit mask of the fused call sites (1 dZ, 2 dX3, 4 dXc, 8 dX1; 15 default, 0 = the separate launches); set < 0: query
// A [BT, C] gate-MLP product with its neighbouring elementwise launches folded in (gemm_skinny.hip; bf16, M <= 256, both operands
// K-major):   D = epi( act( sum_k A'[m][k] B[n][k] + sum_k2 A2[m][k2] B2[n][k2] + bias_n ) masked by (mask > 0) )
//   a_mode 0: A' = A (bf16)          1: A' = bf16(A (bf16) * a_mul (fp32))          2: A' = bf16(A (fp32) * s (1 - s)), s = a_mul (fp32)
//   a_store (E, may be null): A' written once;   epi 0: D (ddt) = v      1: D (E) = v * e_mul (fp32) * (e_q (E) > 0), D2 (fp32) = v * e_q
struct SkFuse {
  int M = 0, N = 0, K = 0;
  int a_mode = 0; const void* A = nullptr; long lda = 0; const float* a_mul = nullptr; long ld_mul = 0; void* a_store = nullptr; long ld_store = 0;
  const void* B = nullptr; long ldb = 0; int b_kmajor = 1;      // (the HIP kernel takes K-major operands only; the flags serve the host emulation)
  int K2 = 0; const void* A2 = nullptr; long lda2 = 0; const void* B2 = nullptr; long ldb2 = 0; int b2_kmajor = 1;
  const float* bias_n = nullptr; int act = ACT_NONE; const void* mask = nullptr; long ldmask = 0;
  int epi = 0; void* D = nullptr; int ddt = DT_F32; long ldd = 0;
  const float* e_mul = nullptr; long ld_emul = 0; const void* e_q = nullptr; long ld_eq = 0; float* D2 = nullptr; long ldd2 = 0;
};
int skfuse_mode(int set);         // dgsct_test_tune "skfuse"
bool skinny_fused_supported(const Ctx&, int M, int N, int K, int K2);
void skinny_fused(const Ctx&, const SkFuse&);
// bench-only: time every GEMM launch of this thread with HIP events (see gemm.hip)
void gemm_prof_enable(int on);
void gemm_prof_collect(long* launches, double* total_ms, double* total_flops);
void zero(const Ctx&, void* p, size_t bytes);
void zero2(const Ctx&, void* a, size_t abytes, void* b, size_t bbytes);      // both in one launch

// out[r][0..L) = softmax(pre_tanh ? tanh(in[r][.]) : in[r][.]); out[r][L..ld_out) = 0.  out dtype odt.
void softmax_rows(const Ctx&, const float* in, long ld_in, void* out, int odt, long ld_out, long rows, int L, int pre_tanh);
// out = P * (s*dP - sum_j P_j*s*dP_j), s = scale_ptr ? *scale_ptr : 1; if dot_accum: *dot_accum += sum P*dP (unscaled).
// P is E; out dtype odt; out[r][L..ldo) = 0.
void softmax_bwd_rows(const Ctx&, const void* P, long ldp, const float* dP, long lddp, void* out, int odt, long ldo,
                      long rows, int L, const float* scale_ptr, float* dot_accum);

// out[b][c] += scale * sum_n roww[b*roww_bs + n] * x[b][n][c]     (roww == null -> 1).  x is E [B][N][ld]; out pre-zeroed.
void colsum_batched(const Ctx&, const void* x, long ld, long bs, int B, int N, int C, const float* roww, long roww_bs,
                    float scale, float* out, long out_bs);
// colsum_batched + the same sum over the POSITIVE entries only: out_pos[b][c] += sum_n roww * (x[b][n][c] > 0)   (pre-zeroed)
// (with roww == null: the number of positive entries -- what a ReLU layer's bias gradient needs once its cotangent is a per-frame
//  vector: sum_n (x > 0) * v[b][c] = v[b][c] * count)
void colsum_batched_pos(const Ctx&, const void* x, long ld, long bs, int B, int N, int C, const float* roww, long roww_bs,
                        float scale, float* out, long out_bs, float* out_pos, long out_pos_bs);
// out[b][n] = sum_c x[b][n][c] * w[b*w_bs + c] * (w2 ? w2[c] : 1) + (bias ? *bias : 0).   w dtype wdt.
void rowdot_batched(const Ctx&, const void* x, long ld, long bs, int B, int N, int C, const void* w, int wdt, long w_bs,
                    const float* w2, const float* bias, float* out);
// Both reductions of one [B][N][C] tensor (E) in a single pass; outputs fp32, PRE-ZEROED, either may be null:
//   out_row[n] += sum_{b,c} x[b][n][c] * w[c]          out_col[c] += sum_{b,n} roww[n] * x[b][n][c]    (roww null -> 1)
// part / part_floats: optional scratch (>= row_part_floats(B, C)) for the per-workgroup partial column sums.
void rowdot_colsum(const Ctx&, const void* x, long ld, long bs, int B, int N, int C, const float* w, const float* roww,
                   float* out_row, float* out_col, float* part = nullptr, long part_floats = 0);
// Same reductions, frames walked one at a time (contiguous rows); row_part: B*N floats of scratch for the per-frame row dots.
void rowdot_colsum_frames(const Ctx&, const void* x, long ld, long bs, int B, int N, int C, const float* w, const float* roww,
                          float* out_row, float* out_col, float* row_part, float* part = nullptr, long part_floats = 0);
// out[i] = (accumulate ? out[i] : 0) + scale * sum_b in[b*bs + i]
void sum_batch(const Ctx&, const float* in, long bs, int B, long n, float* out, float scale, int accumulate);
// y[b][n][c] = x[b][n][c] * (add + colw[b][c])         x,y are E (may alias)
void scale_cols(const Ctx&, const void* x, void* y, int B, int N, int C, const float* colw, float add);
// y[b][n][c] = roww[b][n] * colw[b][c]                       y is E (the backward of a map-weighted token pooling)
void outer_rows(const Ctx&, const float* roww, const float* colw, int B, int N, int C, void* y);
// y[b][n][c] = (x[b][n][c] > 0) * (roww ? roww[b][n] : 1) * colw[b][c] * (colw2 ? colw2[c] : 1) * scale     x,y E (may alias)
// If colsum_out: colsum_out[c] += sum_{b,n} y[b][n][c]  (the bias gradient of the layer whose pre-activation x masks).
void relu_bwd_scale(const Ctx&, const void* x, void* y, int B, int N, int C, const float* roww, const void* colw, int cdt,
                    const float* colw2, float scale, float* colsum_out, float* part = nullptr, long part_floats = 0);
// dX1[b][n][c] += dXc[b][n][c] * (1 + ch[b][c]);  dch[b][c] += sum_n dXc * X1        (all big tensors E, dch pre-initialised)
void xc_bwd(const Ctx&, const void* dXc, const void* X1, void* dX1, int B, int N, int C, const float* ch, float* dch);

// sg = sigmoid(sl); map = softmax_N(tanh(sl))                [B][N] fp32
void spatial_fwd(const Ctx&, const float* sl, int B, int N, float* sg, float* map, float* map2 = nullptr);   // map2: second copy (the returned map)
// dsl = dsg*sg*(1-sg) + [dMap] map*(dMap - sum map*dMap)*(1 - tanh(sl)^2);  *dbs += sum dsl
void spatial_bwd(const Ctx&, const float* sl, const float* sg, const float* map, const float* dsg, const float* dMap,
                 int B, int N, float* dsl, float* dbs);

// X2 = X1 * (alpha*ch[b][c] + beta*sg[b][n] + gamma*tg[b] + 1 - alpha); X3 = lnw ? LN(X2) : X2; mu/rstd [B*N] (if lnw)
void modln_fwd(const Ctx&, const void* X1, const float* ch, const float* sg, const float* tg, float alpha, float beta,
               float gamma, const float* lnw, const float* lnb, float eps, int B, int N, int C, void* X3, float* mu, float* rstd);
// inverse of the above: dX1 = dX2 * mod (written, not accumulated); dlnw/dlnb[c] += ...; dch[b][c] += alpha*sum_n dX2*X1;
// dsg[b][n] = beta * sum_c dX2*X1; dtg[b] += gamma * sum_{n,c} dX2*X1 (if tg)
void modln_bwd(const Ctx&, const void* dX3, const void* X1, const float* ch, const float* sg, const float* tg, float alpha,
               float beta, float gamma, const float* lnw, const float* mu, const float* rstd, int B, int N, int C,
               void* dX1, float* dlnw, float* dlnb, float* dch, float* dsg, float* dtg, float* part = nullptr,
               long part_floats = 0);
// part / part_floats (modln_bwd, tail_bwd): optional scratch, >= row_part_floats(B, C).  When given, the per-channel sums
// leave each workgroup as plain stores of its partial sums and a small second kernel adds them up, instead of one global
// atomic per (workgroup, channel): ~1.5 M atomics on 2048 addresses cost 55 of the 62 us of tail_bwd at C = 512.
long row_part_floats(int B, int C);

// Grouped projections with a tiny per-group narrow width dg = ds/g <= 8 (early-stage bottlenecks), on the vector units:
//   narrow: y[r][gi*dg + jl] = sum_cl x[r][gi*cg + cl] * W(gi, jl, cl)      x E [rows][C]  -> y E [rows][ds]
//   wide:   y[r][gi*cg + cl] = sum_jl x[r][gi*dg + jl] * W(gi, jl, cl)      x E [rows][ds] -> y E [rows][C]
// with W(gi, jl, cl) = W[gi*sg + jl*sj + cl*sc] (fp32 master weight, either orientation), cg = C/g.
// wide + stats (3*C floats, pre-zeroed): also accumulates bn_stats(y) in the same pass.
bool gproj_supported(int mode, int C, int ds, int g);
void gproj_narrow(const Ctx&, const void* x, long rows, int C, int ds, int g, const float* W, long sg, long sj, long sc, void* y);
// dx = BatchNorm backward of dy (as bn_bwd_apply with relu = 0, has_bn = 1; written, E), y = dx (x)_g W: one pass over the wide tensors
// (dgsct_test_tune "rowfuse" = 0: the two launches)
void gproj_narrow_bnb(const Ctx&, const void* dy, const void* xv, void* dx, long rows, int C, int ds, int g, const float* W, long sg,
                      long sj, long sc, void* y, const float* mean, const float* rstd, const float* bsc, const float* bsh, const float* sums, int training);
// modln_fwd + gproj_narrow + bn_stats(y) in one pass over X1 (lnw may be null; stats null = no sums)
bool modln_gproj_supported(int mode, int C, int ds, int g);
int rowfuse_mode(int set);        // test / tuning switch of the fused row passes (dgsct_test_tune "rowfuse")
void modln_gproj(const Ctx&, const void* X1, const float* ch, const float* sg, const float* tg, float alpha, float beta, float gamma,
                 const float* lnw, const float* lnb, float eps, int B, int N, int C, int ds, int g, const float* W, long wsg, long wsj,
                 long wsc, void* X3, float* mu, float* rstd, void* y, float* stats);
void gproj_wide(const Ctx&, const void* x, long rows, int C, int ds, int g, const float* W, long sg, long sj, long sc, void* y,
                float* stats);

// ---- fused row passes of the gate / bottleneck chain (fused_gate.hip; bf16, C in {96, 128, 192, 256}, N % 32 == 0) --------------
// One pass over X1 (E [B][N][C]) instead of scale_cols -> GEMM -> rowdot -> modln_fwd -> gproj_narrow -> bn_stats:
//   sl[b][n] = relu(X1 (1 + ch_b) Wv2^T + bv2) . (aq2_b * ws) + bs;   X2 = X1 (alpha ch_b + beta sigmoid(sl) + gamma tg_b + 1 - alpha);
//   X3 = lnw ? LN(X2) : X2 (+ mu, rstd);   Zp = X3 (x)_g Wd;   stats (3*ds floats, pre-zeroed, may be null): bn_stats(Zp as stored)
// Wv2 [C/2][C], Wd [ds][C/g]: the fp32 MASTER weights.  aq2: E [B][C/2].  vq2 (optional, E [B][N][C/2]): relu(...) for callers that
// still want it materialised.
bool gate_fused_supported(int mode, int N, int C, int ds, int g);
// 0: off, 1: on (default), 2: on, and the forward still materialises vq2 for inspection (tests pin the device's ReLU decisions).
// dgsct_test_tune "gatefuse"; set < 0: query
int gatefuse_mode(int set);
void gatemod_fwd(const Ctx&, const void* X1, const float* ch, const void* aq2, const float* Wv2, const float* bv2, const float* ws,
                 const float* bs, const float* tg, float alpha, float beta, float gamma, const float* lnw, const float* lnb, float eps,
                 int B, int N, int C, int ds, int g, const float* Wd, float* sl, void* X3, float* mu, float* rstd, void* Zp,
                 float* stats, void* vq2);
// Backward of the same chain in one pass over X1 (C in {96, 128}, ds = C / 8): replaces bn_bwd_apply (BN1) -> dX3 projection ->
// modln_bwd -> spatial_bwd -> colsum (u) -> relu_bwd_scale (dvq2) -> dXc GEMM -> xc_bwd.  vq2 is recomputed from X1.
//   in : dZ (E [R][ds], cotangent of relu(bn1(Zp))), Zp, BN1 vectors (mean | rstd | sc | sh) and its backward sums (2*ds, from
//        bn_bwd_stats), saved sl / sg / map / mu / rstd, dMap (may be null)
//   out: dZ <- dZp (in place), dX1 (E, written), dvq2 (E [R][C/2]), Xc = X1 (1 + ch) (E; operand of the dWv2 GEMM),
//        dch [B][C] +=, u [B][C/2] += sum_n dsl vq2, dtg [B] += (tg != null), dlnw / dlnb [C] += (lnw != null), dbv2 [C/2] +=, *dbs +=
//   mapdot_scratch: B floats; part: gate_bwd_part_floats(B, C) floats of scratch.
bool gate_bwd_fused_supported(int mode, int N, int C, int ds, int g);
// vq1 = relu(X1 Wv1^T + bv1) without the tensor (fused_gate.hip; bf16, C in {96, 128}; dgsct_test_tune "vq1fuse"):
//   vq1sum_fwd: msum[b][c] += invN * sum_n vq1[b][n][c]          (Wv1: the prepared E copy, K-major [C][C])
//   vq1_bwd:    dvq1 = (vq1 > 0) * E(coef[b][c] * invN) (written, E), dbv1[c] += sum dvq1, dX1 += dvq1 Wv1 (in place, E)
//               part: >= 1024 * C floats of scratch; with ctx.late set the second stage of dbv1 is left to the caller
int vq1fuse_mode(int set);
bool vq1_fused_shape(int mode, int N, int C);        // (without the switch: layout decisions)
bool vq1_fused_supported(int mode, int N, int C);
void vq1sum_fwd(const Ctx&, const void* X1, const void* Wv1, const float* bv1, int B, int N, int C, float invN, float* msum,
                void* vq1 = nullptr);     // vq1: test mode ("vq1fuse" = 2), the tensor is stored as well
//               dWv1 != null: dWv1 [C][C] += dvq1^T X1 accumulated in the same pass (wpart: vq1_wpart_floats(C) floats of scratch) and
//               dvq1 is not written (may be null)
long vq1_wpart_floats(int C);
void vq1_bwd(const Ctx&, const void* X1, const void* Wv1, const float* bv1, const float* coef, int B, int N, int C, float invN,
             void* dX1, void* dvq1, float* dbv1, float* part, long part_floats, float* dWv1 = nullptr, float* wpart = nullptr);
// the same test without the tuning switch: what the buffer LAYOUT keys on (Xc is scratch, not a saved activation, for these shapes)
bool gate_bwd_fused_shape(int mode, int N, int C, int ds, int g);
long gate_bwd_part_floats(int B, int C);
void gatemod_bwd(const Ctx&, const void* X1, const float* ch, const void* aq2, const float* Wv2, const float* bv2, const float* ws,
                 const float* tg, float alpha, float beta, float gamma, const float* lnw, const float* mu, const float* rstd,
                 const float* sl, const float* sg, const float* map, const float* dMap, int B, int N, int C, int ds, int g, const float* Wd,
                 void* dZ, const void* Zp, const float* bn_mean, const float* bn_rstd, const float* bn_sc, const float* bn_sh,
                 const float* bn_sums, int has_bn, int training, void* dX1, void* dvq2, void* Xc, float* dch, float* u, float* dtg,
                 float* dlnw, float* dlnb, float* dbv2, float* dbs, float* mapdot_scratch, float* part, long part_floats);

// acc = [shift | sum(x-shift) | sum((x-shift)^2)] per column, 3*C floats, pre-zeroed.  x is E [rows][C].
void bn_stats(const Ctx&, const void* x, long rows, int C, float* acc);
// training: mean/var from acc, running stats updated (momentum, unbiased var); eval: mean/var = running.
// Writes mean, rstd, sc = w*rstd, sh = b - mean*sc  (each [C]).
void bn_finalize(const Ctx&, const float* acc, long rows, int C, const float* w, const float* b, float* run_mean,
                 float* run_var, float momentum, float eps, int training, float* mean, float* rstd, float* sc, float* sh);
// y = x*sc + sh (sc == null -> identity), optional relu.  E -> E
void affine_act(const Ctx&, const void* x, void* y, long rows, int C, const float* sc, const float* sh, int relu);
// The arguments of bn_finalize as one value: a CONSUMER of the scale / shift vectors (affine_act_bn, tail_fwd with `fin`) derives
// them from the batch sums itself (two loads, an rsqrt per channel) and its first workgroup stores mean / rstd / sc / sh and
// updates the running statistics -- one launch less on the dependency chain per BatchNorm (dgsct_test_tune "bnfold").
struct BnFin {
  const float* acc; long rows; const float* w; const float* b; float* run_mean; float* run_var; float momentum, eps; int training;
  float* mean; float* rstd; float* sc; float* sh;
};
int bnfold_mode(int set);
void affine_act_bn(const Ctx&, const void* x, void* y, long rows, int C, const BnFin& fin, int relu);
// sums[0][c] += sum dyb, sums[1][c] += sum dyb*xh with dyb = dy * (relu ? (x*sc+sh > 0) : 1), xh = (x-mean)*rstd
void bn_bwd_stats(const Ctx&, const void* dy, const void* x, long rows, int C, const float* mean, const float* rstd,
                  const float* sc, const float* sh, int relu, float* sums, float* part = nullptr, long part_floats = 0);
// dx = training ? sc*(dyb - sums0/rows - xh*sums1/rows) : sc*dyb ;  has_bn == 0: dx = dyb.   in place allowed
void bn_bwd_apply(const Ctx&, const void* dy, const void* x, void* dx, long rows, int C, const float* mean, const float* rstd,
                  const float* sc, const float* sh, const float* sums, int relu, int has_bn, int training);

// O = Op*sc2 + sh2 (sc2 null -> Op).  gate_first: G = gate*O, out = lnw ? LN(G) : G.
// else: L = lnw ? LN(O) : O, out = gate ? gate*L : L.  mu/rstd [B*N] written when lnw.
void tail_fwd(const Ctx&, const void* Op, const float* sc2, const float* sh2, const float* lnw, const float* lnb,
              const float* gate, int gate_first, float eps, long rows, int C, void* out, float* mu, float* rstd,
              const void* residual = nullptr,      // out += residual (E [rows][C]) when given
              const BnFin* fin2 = nullptr);        // BN2 finalised inside (sc2 / sh2 are then the OUTPUT vectors fin2->sc / sh)
// backward of tail_fwd: writes dO (E); accumulates dlnw, dlnb [C], *dgate, and (if bnsums) bnsums[0][c] += sum dO,
// bnsums[1][c] += sum dO * (Op - mean2)*rstd2.
void tail_bwd(const Ctx&, const void* dOut, const void* Op, const float* sc2, const float* sh2, const float* mean2,
              const float* rstd2, const float* lnw, const float* lnb, const float* gate, int gate_first, const float* mu,
              const float* rstd, long rows, int C, void* dO, float* dlnw, float* dlnb, float* dgate, float* bnsums,
              float eps, float* part = nullptr, long part_floats = 0);

// ---- fp8 (e4m3) MFMA projections (gemm_fp8.hip; dtype DGSCT_BF16_FP8) ------------------------------------------------------
// out8[i] = fp8(w[i] * 448 / max|w|), *inv_scale = max|w| / 448;  amax_scratch: 4 bytes of device scratch
void fp8_quantize(const Ctx&, const float* w, long n, void* out8, float* inv_scale, void* amax_scratch);
// D (bf16 [M][ldd]) = act(*inv_scale * sum_k fp8(A[m][k]) W8[n][k] + bias[n] + r1_m[m % m_mod] * r1_n[n]);  A bf16 [M][lda], W8 fp8 [N][K]
void gemm_fp8(const Ctx&, int M, int N, int K, const void* A, long lda, const void* W8, const float* inv_scale, const float* bias,
              int relu, void* D, long ldd, const float* r1_m = nullptr, const float* r1_n = nullptr, int m_mod = 0);

// ---- fused latent-token attention (attn.hip; reference net_trans.py:572-589 and its autograd) ------------------------
// tok fp32 [B][tk][C] = T0 + softmax_N(T0 Yp^T) Yp;  lse fp32 [B][tk] = log sum_n exp(logit);  a fp32 [B][C] = mean_N Yp
// (a must be pre-zeroed; aE: optional copy of a in E);  scratch: tokattn_scratch_floats(B, N, C) floats.  tk <= 32.
long tokattn_scratch_floats(int B, int N, int C);
// ---- the frame-deep weight gradients of the gate MLPs as one launch (gemm_wgbt.hip): D[m][n] = sum_{k < K} A[k][m] * B[k][n], A / B bf16
// [K][lda / ldb] (MN-major), D fp32 [M][ldd], overwritten; up to WGBT_MAX problems per launch
constexpr int WGBT_MAX = 4;
struct WgBtJob { const void* A; const void* B; float* D; int M, N, K; long lda, ldb, ldd; };
bool wgrad_bt_supported(const Ctx&, const WgBtJob* jobs, int n);
void wgrad_bt(const Ctx&, const WgBtJob* jobs, int n);
int wgrad_bt_mode(int set);

// tokpk (optional, bf16 mode): the latent tokens PACKED for the wave-centric fast kernels (attn2.hip): per frame
// [hi 32 x C | lo 32 x C | transposed C x 32] bf16 = tok_pack_elems(B, C) elements; consumers fall back to the generic
// kernels when it is null.
int tokattn_small8_mode(int set);   // dgsct_test_tune "tfs8": 1 (default) the 8-wave short-frame kernel, 0 the 4-wave one
long tok_pack_elems(int nb, int C);
void tok_pack(const Ctx&, const float* src, int nb, int tk, int C, void* pk, const float* other = nullptr, const float* base = nullptr,
              float* D = nullptr);       // D[b][t] = sum_c src * (other - base)   (optional)
void tokattn_fwd(const Ctx&, const void* Yp, const float* T0, int B, int N, int C, int tk, float* tok, float* lse, float* a,
                 void* aE, float* scratch, void* tokpk = nullptr, const void* T0pk = nullptr);
// X1 (E) = X + gate_av * softmax_tk(X tok^T) tok
void xattn_fwd(const Ctx&, const void* X, const float* tok, const float* gate_av, int B, int N, int C, int tk, void* X1,
               const void* tokpk = nullptr);
// backward of xattn_fwd w.r.t. X and tok given dX1 (E): dX (E) = dX1 + dS2 tok (+ R2, optional, E);
// dtok fp32 [B][tk][C] += gate_av P2^T dX1 + dS2^T X  (pre-zeroed);  *dgate += sum P2 (dX1 tok^T)  (optional)
void xattn_bwd(const Ctx&, const void* X, const void* dX1, const float* tok, const float* gate_av, int B, int N, int C, int tk,
               void* dX, const void* R2, float* dtok, float* dgate, const void* tokpk = nullptr);
// backward of tokattn_fwd given dtok: dYp (E) = P1^T dtok + dS1^T T0 + invN * da[b];  dT0b fp32 [B][tk][C] += dS1 Yp
// (pre-zeroed; the my_tokens gradient is sum_b (dtok + dT0b));  Dscratch: B*tk floats.
// T0pk: packed my_tokens (tok_pack with nb = 1; from dgsct_prepare), dtokpk: tok_pack_elems(B, C) bf16 of scratch -- both
// optional (fast path).
void tokattn_bwd(const Ctx&, const void* Yp, const float* T0, const float* tok, const float* lse, const float* dtok,
                 const float* da, float invN, int B, int N, int C, int tk, void* dYp, float* dT0b, float* Dscratch,
                 const void* T0pk = nullptr, void* dtokpk = nullptr);

// ---- num_tokens > 32 (attn_wide.cpp: plain C++ over gemm() and the row softmax kernels; the fused kernels above hold one 32-row token
// tile per frame).  L / dP: fp32 scratch, dS: E scratch, each >= wide_attn_image_elems(B, N, tk) elements; P1 [B][tk][rup8(N)] and
// P2 [B][N][rup8(tk)] (E) are SAVED by the forward.  bf16 mode: T0hi / T0lo = split_hilo(my_tokens) (from dgsct_prepare), tokhi / toklo
// E [B][tk][C] written by xattn_fwd_wide (tokhi is saved), dtokE E [B][tk][C] of scratch; fp32 mode: T0hi = my_tokens, tokhi = tok and
// T0lo = toklo = dtokE = null.  daN: B * C floats of scratch.  dtok / dT0b are WRITTEN (not accumulated).
// hi = bf16(src), lo = bf16(src - hi)
void split_hilo(const Ctx&, const float* src, long n, void* hi, void* lo);
long wide_attn_image_elems(int B, int N, int tk);
void tokattn_fwd_wide(const Ctx&, const void* Yp, const float* T0, const void* T0hi, const void* T0lo, int B, int N, int C, int tk,
                      float invN, float* tok, float* a, void* aE, float* L, void* P1);
void xattn_fwd_wide(const Ctx&, const void* X, const float* tok, const float* gate_av, int B, int N, int C, int tk, void* X1,
                    void* tokhi, void* toklo, float* L, void* P2);
void xattn_bwd_wide(const Ctx&, const void* X, const void* dX1, const void* tokhi, const float* gate_av, int B, int N, int C, int tk,
                    void* dX, const void* R2, float* dtok, float* dgate, const void* P2, float* dP, void* dS);
void tokattn_bwd_wide(const Ctx&, const void* Yp, const void* T0hi, const float* dtok, const float* da, float invN, int B, int N, int C,
                      int tk, void* dYp, float* dT0b, const void* P1, float* dP, void* dS, void* dtokE, float* daN);

// ---- TemporalAttention gate application (temporal.hip; reference net_trans.py:240-251), fp32 rows [R][D] ----------------
void temporal_gate_fwd(const Ctx&, int R, int D, float gamma, const float* akv, const float* vkv, const float* vq, const float* aq,
                       const float* wa, const float* ba, const float* wv, const float* bv, float* out_v, float* out_a, float* gate,
                       float* ga, float* gv);
void temporal_gate_bwd(const Ctx&, int R, int D, float gamma, const float* akv, const float* vkv, const float* vq, const float* aq,
                       const float* wa, const float* wv, const float* ga, const float* gv, const float* dOv, const float* dOa,
                       const float* dg, float* dakv, float* dvkv, float* dvq, float* daq, float* dwa, float* dba, float* dwv, float* dbv);

// Per-frame scalar gate on a feature block (TemporalAttention of AVVP mgn.py:155-156 / AVS PVT_AVSModel.py:572-577), dtype ctx.mode:
//   y[r][i] = x[r][i] * (1 + gamma * g[r])                       r < rows, i < inner (inner * sizeof(E) a multiple of 16)
//   dx = dy * (1 + gamma * g[r]) (optional);  dg[r] = gamma * sum_i dy[r][i] * x[r][i] (optional)
void frame_scale_fwd(const Ctx&, int rows, long inner, float gamma, const void* x, const float* g, void* y);
void frame_scale_bwd(const Ctx&, int rows, long inner, float gamma, const void* x, const float* g, const void* dy, void* dx, float* dg);

// ---- fused window attention of the frozen backbone blocks (wattn.hip; reference htsat.py:50-132) --------------
// O = softmax(scale_h q k^T + bm[w % nwm][h]) v per (frame, window, head), window partition + cyclic shift as address arithmetic on the
// [B][H*W][3][heads][hd] qkv projection of the un-partitioned map; bf16, hd in {8, 16, 24, 32}, ws*ws <= 144.  Return 0 or 2 (set_error).
int window_attn_forward(void* stream, int B, int H, int W, int ws, int shift, int heads, int hd, int nwm, const void* qkv, const float* bm,
                        const float* scale, void* out, float* lse, int cosine = 0);
int window_attn_backward(void* stream, int B, int H, int W, int ws, int shift, int heads, int hd, int nwm, const void* qkv, const float* bm,
                         const float* scale, const void* out, const float* lse, const void* dout, void* dqkv, int cosine = 0);
// cosine != 0 (Swin-V2): q and k rows are L2-normalised inside (x / max(|x|, 1e-12)); backward returns the gradient of the RAW q, k

// Small fp32/E elementwise helpers on [n]-sized vectors (n <= a few 100k).
enum EwOp : int {
  EW_MUL = 0,          // o = a*b
  EW_MUL_MASK = 1,     // o = a*b*(c>0)
  EW_SIGMOID_BWD = 2,  // o = a*b*(1-b)
  EW_SCALE = 3,        // o = s*a
  EW_ADD_BCAST = 4,    // o[i] = a[i] + s*b[i / div]            (b broadcast over the fast dim of size div)
  EW_OUTER_ACC = 5,    // o[i] += a[i / div] * b[i % div]       (rank-1 accumulate, fp32 o)
  EW_COPY = 6,         // o = a
  EW_MULB_MASK = 7,    // o[i] = a[i] * b[i % div] * (c[i] > 0)
  EW_RND_MUL = 8,      // o[i] = round_c.dt(s * a[i]) * b[i]     (the value a ReLU backward wrote, times how often it wrote it; c.p unused)
  EW_MUL3B = 9,        // o[i] = a[i] * b[i] * c[i % div]
};
struct EwArg { const void* p = nullptr; int dt = DT_F32; };
void ew(const Ctx&, int op, void* o, int odt, EwArg a, EwArg b, EwArg c, long n, float s, long div);
struct EwCall { int op; void* o; int odt; EwArg a, b, c; long n; float s; long div; };
void ew2(const Ctx&, EwCall p, EwCall q);      // two independent ops, one launch
// tg[b] = sigmoid(a[b][:] . wt + bt)
void temporal_fwd(const Ctx&, const float* a, const float* wt, const float* bt, int B, int C, float* tg);
// out[i] = (E) in[i]   i < n, for weights: fp32 master -> E copy
void cvt(const Ctx&, const float* in, void* out, int odt, long n);
// Up to CVT_MAX_SEG independent fp32 -> odt[i] copies in ONE launch (all weight casts of dgsct_prepare).
constexpr int CVT_MAX_SEG = 24;
struct CvtSeg { const float* src; void* dst; long n; int odt; long tr_cols; };   // tr_cols > 0: src is [n / tr_cols][tr_cols], dst its transpose
void cvt_multi(const Ctx&, const CvtSeg* segs, int nseg);
// Up to COLSUM_MAX_SEG independent small column sums in ONE launch: out[c] += sum_{r < rows} x[r][c] (x: rows x C, dtype dt, leading
// dimension C).  For the per-frame-vector gradients ([BT][C] matrices: bias gradients of the gate MLPs): five single-workgroup
// launches of 10-30 us each per adapter call otherwise.
constexpr int COLSUM_MAX_SEG = 8;
struct ColsumSeg { const void* x; int dt; int rows; int C; float* out; };
void colsum_multi(const Ctx&, const ColsumSeg* segs, int nseg);
// out[r] = sum_c W[r][c]  (fp32 [R][C] -> fp32 [R])
void rowsum_f32(const Ctx&, const float* W, int R, int C, float* out);

}  // namespace dgsct
