/* dgsct.h -- C ABI of libdgsct.so: the MI355X (gfx950) DG-SCT cross-modal adapter path.
 *
 * Drop-in boundary.  The reference has no FFI of its own -- the hot path is the
 * Python method `VisualAdapter.forward(x, vis_token) -> (output, spatial_att_maps)`
 * (reference DG-SCT/AVE/nets/net_trans.py:552-674; ctor :437-550) and its autograd backward.
 * These entry points are what a ctypes/cffi binding of that method binds (INTEGRATION.md shows it):
 *
 *   dgsct_query            sizes of every caller-owned buffer + gradient layout     (ctor-time)
 *   dgsct_prepare          fp32 master parameters -> MFMA-operand copies + derived bias vectors
 *   dgsct_adapter_forward  replaces VisualAdapter.forward            net_trans.py:552-674
 *   dgsct_adapter_backward replaces autograd of the same graph       (a-9 in SURVEY.md 8a)
 *
 * Rules of the boundary
 *   - plain C: pointers + sizes only, no torch types; every pointer is a DEVICE pointer owned by the
 *     caller (PyTorch's caching allocator); the library never allocates, frees or keeps device memory;
 *   - every call is asynchronous on `stream` (a hipStream_t) and re-entrant (no global mutable state),
 *     so it is safe under nn.DataParallel's per-replica threads (reference AVS/AVQA call sites; the Python
 *     mirror resolves a replica's broadcast weights itself);
 *   - every call returns 0 on success; on failure a non-zero code and dgsct_last_error() (thread
 *     local) describes it -- the Python wrapper raises RuntimeError, mirroring the reference's
 *     exception-only error behaviour (NotImplementedError for unsupported adapter kinds, :549-550);
 *   - tensors are token-major: X [BT][N][C], Y [BT][No][Co], i.e. the memory the reference's
 *     `f.permute(0,2,1).unsqueeze(-1)` views alias (net_trans.py:891-892), element type = desc.dtype.
 */
#ifndef DGSCT_H
#define DGSCT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGSCT_VERSION 100
enum { DGSCT_F32 = 0, DGSCT_BF16 = 1,
       DGSCT_BF16_FP8 = 2 };  /* bf16 storage / MFMA everywhere, plus fp8 (OCP e4m3) MFMA operands for the three big weight-stationary
                                 forward projections fc, fc_affine_video_1, fc_affine_video_2 (BASELINE.json configs[4]); buffers are
                                 laid out exactly as for DGSCT_BF16, tensors handed in and out are bf16 */
enum { DGSCT_REMAP_CONV = 0,    /* conv_adapter (token axis) + fc           net_trans.py:553-554        */
       DGSCT_REMAP_FIXED = 1 }; /* fc + fixed token operator (AVS-S4 bicubic resize, PVT_AVSModel.py:190-197);
                                   params[DGSCT_P_WN] is then the dense [N][No] operator, it gets no gradient */

/* Parameter table: fp32 master pointers in this order (reference state_dict names in comments,
 * net_trans.py:444-515).  Unused entries may be NULL. */
enum {
  DGSCT_P_GATE = 0,   /* gate [1]                         (NULL when use_gate == 0)             */
  DGSCT_P_TOKENS,     /* my_tokens [tk][C]                                                      */
  DGSCT_P_GATE_AV,    /* gate_av [1]                                                            */
  DGSCT_P_WN,         /* conv_adapter.weight [N][No]      (or the fixed remap operator)         */
  DGSCT_P_BN,         /* conv_adapter.bias [N]            (NULL for DGSCT_REMAP_FIXED)          */
  DGSCT_P_WC,         /* fc.weight [C][Co]                                                      */
  DGSCT_P_BC,         /* fc.bias [C]                                                            */
  DGSCT_P_WA1,        /* fc_affine_audio_1.weight [C][C]                                        */
  DGSCT_P_BA1,        /* fc_affine_audio_1.bias [C]                                             */
  DGSCT_P_WV1,        /* fc_affine_video_1.weight [C][C]                                        */
  DGSCT_P_BV1,        /* fc_affine_video_1.bias [C]                                             */
  DGSCT_P_WB,         /* fc_affine_bottleneck.weight [C/2][C]                                   */
  DGSCT_P_BB,         /* fc_affine_bottleneck.bias [C/2]                                        */
  DGSCT_P_WV2,        /* fc_affine_video_2.weight [C/2][C]                                      */
  DGSCT_P_BV2,        /* fc_affine_video_2.bias [C/2]                                           */
  DGSCT_P_WA2,        /* fc_affine_audio_2.weight [C/2][C]                                      */
  DGSCT_P_BA2,        /* fc_affine_audio_2.bias [C/2]                                           */
  DGSCT_P_WS,         /* fc_affine_v_s_att.weight [1][C/2]                                      */
  DGSCT_P_BS,         /* fc_affine_v_s_att.bias [1]                                             */
  DGSCT_P_WCATT,      /* fc_affine_v_c_att.weight [C][C/2]                                      */
  DGSCT_P_BCATT,      /* fc_affine_v_c_att.bias [C]                                             */
  DGSCT_P_WD,         /* down_sampler.weight [C/r][C/g]                                         */
  DGSCT_P_WU,         /* up_sampler.weight [C][C/r/g]                                           */
  DGSCT_P_BN1_W, DGSCT_P_BN1_B, DGSCT_P_BN1_RM, DGSCT_P_BN1_RV,   /* bn1.{weight,bias,running_mean,running_var} [C/r] */
  DGSCT_P_BN2_W, DGSCT_P_BN2_B, DGSCT_P_BN2_RM, DGSCT_P_BN2_RV,   /* bn2.* [C]                  */
  DGSCT_P_LNB_W, DGSCT_P_LNB_B,                                   /* ln_before.{weight,bias} [C]  */
  DGSCT_P_LNP_W, DGSCT_P_LNP_B,                                   /* ln_post.{weight,bias} [C]    */
  DGSCT_P_WT, DGSCT_P_BT,       /* temporal_gated.0.{weight [1][C], bias [1]}  (pretrain/few/zero-shot flavour) */
  DGSCT_P_COUNT
};

typedef struct dgsct_adapter_desc {
  int32_t BT;        /* frames = B*T, T fastest ('(b t)' flattening, net_trans.py:854)            */
  int32_t T;         /* frames per clip (10; 5 for AVS) -- only checked (BT % T == 0)             */
  int32_t N, C;      /* own modality: tokens, width   (conv_dim_out, input_dim == linear_out)     */
  int32_t No, Co;    /* other modality: tokens, width (conv_dim_in, linear_in)                    */
  int32_t tk;        /* latent tokens (num_tk / opt.num_tokens), 1..1024; > 32: csrc/attn_wide.cpp */
  int32_t r, g;      /* reduction_factor (opt.Adapter_downsample), opt.num_conv_group             */
  int32_t dtype;     /* DGSCT_F32 | DGSCT_BF16 | DGSCT_BF16_FP8: activation storage + MFMA operand type */
  int32_t remap;     /* DGSCT_REMAP_*                                                             */
  int32_t use_bn, use_gate, ln_before, ln_post;
  int32_t gate_before_ln_post;   /* AVS-S4/MS3 order (PVT_AVSModel.py:308-313)                    */
  int32_t temporal;  /* + gamma*sigmoid(temporal_gated(a)) in the modulation; returns tmap        */
  int32_t training;  /* BatchNorm: batch statistics + running-stat update (1) or running stats (0) */
  float alpha, beta, gamma;      /* modulation weights (0.3, 0.05, 0 for AVE; net_trans.py:611)   */
  float eps, bn_momentum;        /* 1e-5, 0.1                                                     */
} dgsct_adapter_desc;

typedef struct dgsct_sizes {
  int64_t prep_bytes;     /* dgsct_prepare output                                                  */
  int64_t saved_bytes;    /* activations kept from forward to backward (one per adapter call)      */
  int64_t ws_fwd_bytes;   /* forward scratch (reusable across adapters on one stream)              */
  int64_t ws_bwd_bytes;   /* backward scratch                                                      */
  int64_t grad_floats;    /* length of the flat fp32 gradient buffer                               */
  int64_t grad_offset[DGSCT_P_COUNT];   /* float offset of each parameter's gradient, -1 = no gradient */
  int64_t grad_numel[DGSCT_P_COUNT];
} dgsct_sizes;

int dgsct_version(void);
const char* dgsct_arch(void);            /* "gfx950" */
const char* dgsct_last_error(void);      /* thread-local, valid until the next failing call on this thread */

int dgsct_query(const dgsct_adapter_desc* desc, dgsct_sizes* out);

/* A HIP stream in its own priority class (-1 high, 0 normal, +1 low).  The HIP runtime multiplexes all streams of one
 * priority onto a small pool of hardware queues (4 by default), and which streams end up sharing a queue depends on
 * creation order (e.g. whether RCCL was initialised first): two "concurrent" streams on one queue serialise.  Queue
 * pools are per priority, so the caller stream (normal), the second adapter stream (high) and the weight-gradient
 * stream (low, off the critical path) are guaranteed distinct queues.  No reference counterpart (the reference runs
 * everything on torch's current stream). */
int dgsct_stream_create(int priority_class, void** stream);
int dgsct_stream_destroy(void* stream);

/* params[DGSCT_P_COUNT]: fp32 device pointers.  Writes `prep` (prep_bytes).  Must be re-run whenever a
 * parameter changed (after every optimizer step); cheap (one pass over the weights). */
int dgsct_prepare(const dgsct_adapter_desc* desc, float* const* params, void* prep, void* stream);

/* out [BT][N][C] (dtype), map [BT][N] fp32 (softmax over N: the 2nd return value of the reference),
 * tmap [BT] fp32 or NULL.  When desc.training and use_bn, bn running stats in params[] are updated
 * in place (num_batches_tracked is the caller's). */
int dgsct_adapter_forward(const dgsct_adapter_desc* desc, float* const* params, const void* prep,
                          const void* X, const void* Y, void* out, float* map, float* tmap,
                          void* saved, void* ws, void* stream);

/* Same, with the caller's residual add fused into the last kernel (reference call sites
 * net_trans.py:894-898,903-906: `f = f + adapter(...)[0].squeeze(-1).permute(0,2,1)`):
 *   out = residual + adapter(X, Y)        residual [BT][N][C] (dtype), may alias X (identity / pre-block skip).
 * aux_stream (optional, any other stream of the device): the audio-query branch (a = mean_N Yp, aq1, aq2), which depends
 * on Yp only, runs on it beside the two token attentions; the call forks from and joins back into `stream`.
 * residual == NULL and aux_stream == NULL is dgsct_adapter_forward. */
int dgsct_adapter_forward_ex(const dgsct_adapter_desc* desc, float* const* params, const void* prep,
                             const void* X, const void* Y, const void* residual, void* out, float* map, float* tmap,
                             void* saved, void* ws, void* stream, void* aux_stream);

/* dOut [BT][N][C] (dtype); dMap [BT][N] fp32 or NULL; dTmap [BT] fp32 or NULL.
 * Writes dX [BT][N][C], dY [BT][No][Co] (dtype) and the flat fp32 gradient buffer `grads`
 * (grad_floats; overwritten, not accumulated). */
int dgsct_adapter_backward(const dgsct_adapter_desc* desc, float* const* params, const void* prep,
                           const void* X, const void* Y, const void* saved,
                           const void* dOut, const float* dMap, const float* dTmap,
                           void* dX, void* dY, float* grads, void* ws, void* stream);

/* Same, with an optional second stream: weight / bias gradients (which feed nothing downstream) are issued on
 * `aux_stream` and overlap the data-gradient chain; the call forks from and joins back into `stream`, so the caller sees
 * ordinary single-stream semantics.  aux_stream == NULL is dgsct_adapter_backward.
 * `skip_into_dx` is a flag word (0 / 1 keep their round-1 meaning):
 *   DGSCT_BWD_SKIP_INTO_DX (1): the forward was `out = X + adapter(X, Y)` (residual aliased X), so dX also receives dOut
 *     (dX = dOut + d adapter / dX) -- saves the caller's gradient-accumulation pass over [BT][N][C];
 *   DGSCT_BWD_NO_JOIN (2, round 5; ignored without an aux stream): the call does NOT join the aux stream back into `stream`.  dX
 *     and dY are complete in `stream` order as always; `grads` is complete, and `ws`, `saved`, X, Y, dOut may be reused / freed, only
 *     once everything this call enqueued on `aux_stream` has run -- the CALLER orders that (an event recorded on aux_stream after the
 *     call, waited for before the next use).  The chain of the caller's next call on `stream` then never waits for this call's
 *     weight gradients (0.9 ms of the 50 ms AVE step); dg-sct_amd/ops.py does it with two alternating workspaces per stream. */
#define DGSCT_BWD_SKIP_INTO_DX 1
#define DGSCT_BWD_NO_JOIN 2
int dgsct_adapter_backward_ex(const dgsct_adapter_desc* desc, float* const* params, const void* prep,
                              const void* X, const void* Y, const void* saved,
                              const void* dOut, const float* dMap, const float* dTmap,
                              void* dX, void* dY, float* grads, void* ws, void* stream, void* aux_stream,
                              int skip_into_dx);

/* ---- the two adapters of a position as one backward (round 5) ---------------------------------------
 * Every caller runs an audio and a visual adapter on the same pair of maps and adds each result to its own map
 * (`f_a = f_a + audio_adapter(f_a, f_v)`, `f_v = f_v + vis_adapter(f_v, f_a)`: net_trans.py:891-906 and the same lines of the other
 * tasks), so   d f_a = dX(audio call) + dY(visual call)   and   d f_v = dX(visual call) + dY(audio call).
 * autograd forms those sums with one more pass over [BT][N][C] per call; here the sum is the epilogue of the product that writes dY.
 * That product is the last link of a call's data-gradient chain, so a call is issued in two parts:
 *   flags |= DGSCT_BWD_HOLD_DY : everything except the dY product (dY is not written).  `dx_ready_event` (a hipEvent_t, may be NULL)
 *                                is recorded on `stream` as soon as dX is complete -- long before the call's last kernel;
 *   flags |= DGSCT_BWD_ONLY_DY : only the dY product: dY = dy_residual + d adapter / dY, issued on `stream` behind `dy_wait_event`
 *                                (hipEvent_t, may be NULL).  Reads `ws` as the HOLD_DY part of the same call left it (same desc, params,
 *                                prep, ws, stream; X / Y / saved / dOut / dX / grads are not touched and may be NULL).
 * Order on the host: HOLD_DY part of both calls (each on its own stream), then the ONLY_DY parts, each with the OTHER call's dX as
 * dy_residual and the other call's dx_ready_event as dy_wait_event: the waits are then enqueued after the records they refer to.
 * dy_residual: [BT][No][Co] of desc.dtype (the other adapter's dX has exactly this shape), NULL = none.  Without either flag the
 * call is dgsct_adapter_backward_ex (dy_residual and the events still honoured). */
#define DGSCT_BWD_HOLD_DY 4
#define DGSCT_BWD_ONLY_DY 8
typedef struct dgsct_bwd_opts {
  int32_t flags;                 /* DGSCT_BWD_* */
  const void* dy_residual;
  void* dx_ready_event;          /* hipEvent_t */
  void* dy_wait_event;           /* hipEvent_t */
} dgsct_bwd_opts;
int dgsct_adapter_backward_ex2(const dgsct_adapter_desc* desc, float* const* params, const void* prep,
                               const void* X, const void* Y, const void* saved,
                               const void* dOut, const float* dMap, const float* dTmap,
                               void* dX, void* dY, float* grads, void* ws, void* stream, void* aux_stream,
                               const dgsct_bwd_opts* opts);

/* ---- spatial-map pooling of the task heads ----------------------------------
 * Replaces `f_v = torch.bmm(f_v_spatial_att_maps, f_v)` / `f_a = torch.bmm(f_a_spatial_att_maps, f_a)`
 * (DG-SCT/AVE/nets/net_trans.py:922-924; same lines in AVVP/nets/mgn.py and pretrain/nets/net_trans.py): the maps
 * returned by the LAST p2 adapters pool the final token maps into one feature vector per frame.
 *   forward : pooled[b][c] = sum_n map[b][n] * F[b][n][c]
 *   backward: dF[b][n][c]  = map[b][n] * dPooled[b][c];   dMap[b][n] = sum_c F[b][n][c] * dPooled[b][c]
 * F, dF: [BT][N][C] contiguous, dtype DGSCT_F32 | DGSCT_BF16; map, dMap: fp32 [BT][N]; pooled, dPooled: fp32 [BT][C].
 * C must be a multiple of 4 (every backbone width is).  Sums are fp32.  dF / dMap may be NULL (not needed).  Asynchronous on `stream`; returns 0 or an error code. */
int dgsct_map_pool_forward(int dtype, int BT, int N, int C, const void* F, const float* map, float* pooled, void* stream);
int dgsct_map_pool_backward(int dtype, int BT, int N, int C, const void* F, const float* map, const float* dPooled,
                            void* dF, float* dMap, void* stream);

/* ---- fused window attention of the FROZEN backbone blocks ------------------------------------
 * Replaces the body of HTS-AT's `WindowAttention.forward` between its qkv and proj Linears (DG-SCT/AVE/nets/htsat.py:103-131)
 * together with the window partition / reverse and the cyclic `torch.roll`s of the block around it (htsat.py:196-229), and the same
 * part of the timm Swin-V2 block the AVE loop calls at DG-SCT/AVE/nets/net_trans.py:894 (cosine form: q, k passed in normalised):
 *   O[b][p(w,i)][h][:] = sum_j softmax_j( scale[h] * q_i . k_j + bm[w % nwm][h][i][j] ) v_j
 * qkv  : bf16 [B][H*W][3][heads][hd] -- the qkv projection of the UN-partitioned, un-rolled token-major map;
 * bm   : fp32 [nwm][heads][n][n], n = ws*ws -- relative-position bias plus (nwm = number of windows) the additive shift mask;
 *        nwm = 1 for un-shifted blocks.  Frozen: no gradient is produced for it or for `scale` (fp32 [heads]);
 * out  : bf16 [B][H*W][heads][hd];   lse: fp32 [B][windows][heads][n] (kept for backward);
 * backward: dqkv laid out like qkv, every element written (no pre-zeroing needed).
 * p(w,i) is the map position of token i of window w after the block's roll by -shift: ((wy*ws + iy + shift) % H, (wx*ws + ix + shift) % W).
 * Limits: ws*ws <= 144 and a multiple of 4, hd in {8,16,24,32}, H % ws == W % ws == 0.  Asynchronous on `stream`; 0 or an error code. */
int dgsct_window_attn_forward(int B, int H, int W, int ws, int shift, int heads, int hd, int nwm, const void* qkv, const float* bm,
                              const float* scale, void* out, float* lse, void* stream);
int dgsct_window_attn_backward(int B, int H, int W, int ws, int shift, int heads, int hd, int nwm, const void* qkv, const float* bm,
                               const float* scale, const void* out, const float* lse, const void* dout, void* dqkv, void* stream);
/* ..._ex: the same with `flags`.  DGSCT_WATTN_COSINE: q and k rows are L2-normalised inside the kernel, x / max(|x|, 1e-12) as
 * F.normalize does (the timm Swin-V2 block's cosine attention: qkv is then the RAW projection, and backward returns the gradient of
 * the raw q / k through the normalisation) -- no normalised copy of the map, no autograd nodes for it. */
#define DGSCT_WATTN_COSINE 1
int dgsct_window_attn_forward_ex(int B, int H, int W, int ws, int shift, int heads, int hd, int nwm, int flags, const void* qkv, const float* bm,
                                 const float* scale, void* out, float* lse, void* stream);
int dgsct_window_attn_backward_ex(int B, int H, int W, int ws, int shift, int heads, int hd, int nwm, int flags, const void* qkv, const float* bm,
                                  const float* scale, const void* out, const float* lse, const void* dout, void* dqkv, void* stream);

/* ---- LayerNorm (+ residual) of the FROZEN backbone blocks ------------------------------------
 * Replaces `self.norm1(x)` / `self.norm2(x)` of the HTS-AT block (DG-SCT/AVE/nets/htsat.py:192, :236) and, with `residual`, the whole
 * residual line of a timm Swin-V2 half-block as the AVE loop writes it, `f_v = f_v + blk.norm1(blk._attn(f_v))`
 * (DG-SCT/AVE/nets/net_trans.py:894, :903).  The row kernels are the adapter tail's (csrc/prims_hip.hip: tail_fwd / tail_bwd).
 *   forward : out[r][:] = LN(x[r][:]; w, b, eps) (+ residual[r][:]);   mu / rstd: fp32 [rows], kept for backward
 *   backward: dx = LayerNorm backward of dout;  dw[c] += sum_r dout * xhat,  db[c] += sum_r dout   (dw, db: fp32 [C], caller-zeroed;
 *             the residual's gradient is dout itself)
 * x, residual, out, dout, dx: [rows][C] contiguous, dtype DGSCT_F32 | DGSCT_BF16; w, b: fp32 [C].  C % 4 == 0 (bf16: % 8), C <= 1536.
 * scratch: dgsct_layer_norm_scratch_floats(C) floats (per-workgroup partial sums of dw / db; NULL: atomics instead).
 * Asynchronous on `stream`; 0 or an error code. */
int64_t dgsct_layer_norm_scratch_floats(int C);
int dgsct_layer_norm_forward(int dtype, int64_t rows, int C, const void* x, const float* w, const float* b, float eps, const void* residual,
                             void* out, float* mu, float* rstd, void* stream);
int dgsct_layer_norm_backward(int dtype, int64_t rows, int C, const void* dout, const void* x, const float* w, const float* b, const float* mu,
                              const float* rstd, float eps, void* dx, float* dw, float* db, float* scratch, void* stream);

/* ---- gate application of the post-backbone TemporalAttention ---------------------------
 * Replaces the tail of `TemporalAttention.forward` (DG-SCT/AVE/nets/net_trans.py:240-251; AVVP nets/mgn.py:148-159):
 *   audio_gate = audio_gated(audio_key_value_feature); video_gate = video_gated(video_key_value_feature)     [T,B,1]
 *   video_query_output += audio_gate * video_query_output * gamma;  audio_query_output += video_gate * audio_query_output * gamma
 *   audio_visual_gate = audio_gate * video_gate
 * with `*_gated` = nn.Sequential(nn.Linear(d_model, 1), nn.Sigmoid()).  All tensors fp32, rows R = T * B, D = d_model
 * (a multiple of 4, <= 1024).  forward writes out_v, out_a [R][D], gate [R] and the two gates ga, gv [R] (kept for backward);
 * backward writes the input gradients and OVERWRITES dwa, dwv [D], dba, dbv [1].  dg may be NULL. */
int dgsct_temporal_gate_forward(int R, int D, float gamma, const float* akv, const float* vkv, const float* vq, const float* aq,
                                const float* wa, const float* ba, const float* wv, const float* bv, float* out_v, float* out_a,
                                float* gate, float* ga, float* gv, void* stream);
int dgsct_temporal_gate_backward(int R, int D, float gamma, const float* akv, const float* vkv, const float* vq, const float* aq,
                                 const float* wa, const float* wv, const float* ga, const float* gv, const float* dOv,
                                 const float* dOa, const float* dg, float* dakv, float* dvkv, float* dvq, float* daq, float* dwa,
                                 float* dba, float* dwv, float* dbv, void* stream);

/* Per-frame scalar gate on a feature block: the tail of the AVVP / AVS copies of TemporalAttention
 * (`x + gate * x * gamma` with one gate per frame: AVVP/nets/mgn.py:155-156 on [B*10][128] features,
 * avs_s4/model/PVT_AVSModel.py:572-577 on the four [B*5][256][H][W] decoder maps).
 *   forward : y[r][i] = x[r][i] * (1 + gamma * g[r])                       r < rows, i < inner
 *   backward: dx = dy * (1 + gamma * g[r]) (NULL: skipped);  dg[r] = gamma * sum_i dy[r][i] x[r][i] (NULL: skipped)
 * dtype DGSCT_F32 | DGSCT_BF16 for x / y / dy / dx; g, dg fp32; inner * element size must be a multiple of 16 bytes. */
int dgsct_frame_scale_forward(int dtype, int rows, int64_t inner, float gamma, const void* x, const float* g, void* y, void* stream);
int dgsct_frame_scale_backward(int dtype, int rows, int64_t inner, float gamma, const void* x, const float* g, const void* dy,
                               void* dx, float* dg, void* stream);

/* ---- introspection / test hooks (used by tests/ only) ------------------------------------------ */
/* i-th named region of the `saved` buffer; returns 0 and fills name/offset/bytes, or 1 past the end. */
int dgsct_saved_region(const dgsct_adapter_desc* desc, int i, char* name, int name_cap, int64_t* offset, int64_t* bytes);

/* One GEMM of the engine (see csrc/prims.h: struct Gemm).  Plain-C mirror for unit tests. */
typedef struct dgsct_gemm_args {
  int32_t mode;           /* DGSCT_F32 | DGSCT_BF16 */
  int32_t M, N, K, KB, batch, splitk, atomic;      /* atomic: 1 = accumulate into D with fp32 atomics (split-K); 2 = the same
                                                        contract AND D is pre-zeroed with this product as its only writer (an unsplit
                                                        pass may then store rows plainly) */
  const void* A; int64_t lda; int32_t a_kmajor; int64_t a_bs, a_kbs;
  const void* B; int64_t ldb; int32_t b_kmajor; int64_t b_bs, b_kbs;
  void* D; int32_t ddt; int64_t ldd, dbs;
  float alpha; const float* alpha_ptr;
  const float* bias_m; const float* bias_n; int64_t bias_n_bs; int32_t m_mod;
  const float* r1_m; const float* r1_n;
  int32_t act;
  const void* R; int32_t rdt; int64_t ldr, rbs; float beta;
  const void* mask; int64_t ldmask, maskbs;
  const void* R2;                                   /* second residual (dtype/ld/stride of R), weight 1            */
  const float* sm_scale; float* sm_dot;             /* act 3/4 (column softmax epilogues, transposed output)       */
} dgsct_gemm_args;
int dgsct_test_gemm(const dgsct_gemm_args* a, void* stream);

/* One fused latent-token attention kernel (csrc/attn.hip; reference net_trans.py:572-589 and its autograd), for unit
 * tests against torch and for tools/attn_bench.py.  op: 0 tokattn_fwd, 1 xattn_fwd, 2 xattn_bwd, 3 tokattn_bwd,
 * 4 pack T0 into T0pk (what dgsct_prepare does).
 * Activations X / Yp / dX1 / out / R2: [B][N][C] in `mode`'s element type; T0 [tk][C], tok / dtok / dT0b [B][tk][C], lse [B][tk],
 * a / da [B][C]: fp32.  Accumulated outputs (a, dtok, dT0b, dgate) must be zeroed by the caller.  scratch: at least
 * dgsct_test_attn_scratch_floats(B, N, C, tk) floats. */
typedef struct dgsct_attn_args {
  int32_t mode, B, N, C, tk;
  const void* X; const void* Yp; const void* dX1; const void* R2; void* out;
  const float* T0; float* tok; float* lse; float* a; void* aE; const float* gate_av;
  float* dtok; float* dgate; const float* da; float invN; float* dT0b; float* scratch;
  void* tokpk;         /* optional: 96 * B * C bf16 (packed latent tokens: written by op 0, read by ops 1-2) */
  void* T0pk;          /* optional: 96 * C bf16 (packed my_tokens: written by op 4, read by op 3)                 */
  void* dtokpk;        /* optional: 96 * B * C bf16 of scratch (op 3)                                             */
} dgsct_attn_args;
int64_t dgsct_test_attn_scratch_floats(int B, int N, int C, int tk);
int dgsct_test_attn(int op, const dgsct_attn_args* a, void* stream);

/* The fp8 projection kernel on its own (csrc/gemm_fp8.hip), for unit tests: quantises W (fp32 [N][K]) per tensor into
 * `w8` (N*K bytes) / `scale` (>= 2 floats of scratch) exactly like dgsct_prepare does, then
 * D (bf16 [M][N]) = act(A (bf16 [M][K]) . W^T + bias).  K a multiple of 16, N a multiple of 8. */
int dgsct_test_gemm_fp8(int M, int N, int K, const void* A, const float* W, const float* bias, int relu, void* D, void* w8,
                        float* scale, void* stream);

/* ---- measurement hook (bench.py roofline leg) ---------------------------------------------------
 * While enabled, every launch of the MFMA GEMM family issued from the calling thread is bracketed by
 * HIP events on its own stream; collect() synchronises them and returns the number of launches, the sum
 * of their durations and the sum of their useful FLOPs (2*M*N*K per batch), then clears the log. */
/* Test / tuning hook: set an internal switch, returns its previous value (-1: unknown key).
 *   "gemm8": 0 = 8-wave deep-product GEMM kernel off, 1 = on for shapes that fill the chip (default), 2 = on for every
 *   eligible shape (lets the unit tests reach it with small matrices); value < 0 only queries.
 *   "skinny": 0 / 1 = the K-split kernel of the [BT, C] gate-MLP products off / on (default on).
 *   "rowfuse": 0 / 1 = the fused row passes of stages 0-1 (modulation + ln_before + down-projection + BN1 sums in one
 *   kernel; BatchNorm-2's backward inside the narrow projection's pass) off / on (default on; off = the separate launches).
 *   "gatefuse": 0 / 1 = the fused gate / bottleneck passes of the early stages (fused_gate.hip) off / on; 2 = on, and the forward
 *   also stores vq2 (for tests that read the device's ReLU decisions back).
 *   "vq1fuse": 0 / 1 = vq1 never stored at C = 96 / 128 (vq1_fwd_k / vq1_bwd_k) off / on; 2 = on + the tensor stored as well;
 *   3 = on, with dWv1 accumulated inside the backward pass (experiment, slower).  Forward and backward of one call must run
 *   under the same setting of "gatefuse" / "vq1fuse" (0 vs non-0): the backward of a fused forward has no stored tensor to read.
 *   "skfuse": 0 / 1 = elementwise neighbours folded into the gate-MLP products off / on.
 *   "bnfold": 0 / 1 = BatchNorm finalisation inside its consumers off / on.
 *   "gemmtall": 0 / 1 = gemm_tall.hip for the weight gradients over the token rows off / on (from 262 144 rows: stage 0);
 *   2 = on from 16 384 rows (tests).
 *   "callprof": 1 / 0 = record an event pair around every adapter call on its stream / stop; 2 = dump "kind N C stream t0 t1" (us)
 *   to $DGSCT_CALL_PROF (default /tmp/dgsct_callprof.txt) and clear. */
int dgsct_test_tune(const char* key, int value);

int dgsct_prof_enable(int on);
int dgsct_prof_collect(int64_t* launches, double* total_ms, double* total_flops);

#ifdef __cplusplus
}
#endif
#endif /* DGSCT_H */
