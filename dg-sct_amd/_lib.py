"""ctypes binding of libdgsct.so (include/dgsct.h).  No CPU fallback: if the HIP library is missing
or does not load, importing the product path raises -- build it with ``python -m dgsct_amd.build``
(or ``__graft_entry__.build()``)."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdgsct.so")

F32, BF16, BF16_FP8 = 0, 1, 2
REMAP_CONV, REMAP_FIXED = 0, 1

# parameter table order == enum in include/dgsct.h; values are the reference state_dict names
PARAM_NAMES: List[str] = [
    "gate", "my_tokens", "gate_av", "conv_adapter.weight", "conv_adapter.bias", "fc.weight", "fc.bias",
    "fc_affine_audio_1.weight", "fc_affine_audio_1.bias", "fc_affine_video_1.weight", "fc_affine_video_1.bias",
    "fc_affine_bottleneck.weight", "fc_affine_bottleneck.bias", "fc_affine_video_2.weight", "fc_affine_video_2.bias",
    "fc_affine_audio_2.weight", "fc_affine_audio_2.bias", "fc_affine_v_s_att.weight", "fc_affine_v_s_att.bias",
    "fc_affine_v_c_att.weight", "fc_affine_v_c_att.bias", "down_sampler.weight", "up_sampler.weight",
    "bn1.weight", "bn1.bias", "bn1.running_mean", "bn1.running_var",
    "bn2.weight", "bn2.bias", "bn2.running_mean", "bn2.running_var",
    "ln_before.weight", "ln_before.bias", "ln_post.weight", "ln_post.bias",
    "temporal_gated.0.weight", "temporal_gated.0.bias",
]
P_COUNT = len(PARAM_NAMES)
P_INDEX = {n: i for i, n in enumerate(PARAM_NAMES)}
P_WN = P_INDEX["conv_adapter.weight"]


class AdapterDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("BT", "T", "N", "C", "No", "Co", "tk", "r", "g", "dtype", "remap", "use_bn", "use_gate",
                                         "ln_before", "ln_post", "gate_before_ln_post", "temporal", "training")] + \
               [(n, C.c_float) for n in ("alpha", "beta", "gamma", "eps", "bn_momentum")]


class Sizes(C.Structure):
    _fields_ = [("prep_bytes", C.c_int64), ("saved_bytes", C.c_int64), ("ws_fwd_bytes", C.c_int64),
                ("ws_bwd_bytes", C.c_int64), ("grad_floats", C.c_int64),
                ("grad_offset", C.c_int64 * P_COUNT), ("grad_numel", C.c_int64 * P_COUNT)]


class GemmArgs(C.Structure):
    _fields_ = [("mode", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("KB", C.c_int32),
                ("batch", C.c_int32), ("splitk", C.c_int32), ("atomic", C.c_int32),
                ("A", C.c_void_p), ("lda", C.c_int64), ("a_kmajor", C.c_int32), ("a_bs", C.c_int64), ("a_kbs", C.c_int64),
                ("B", C.c_void_p), ("ldb", C.c_int64), ("b_kmajor", C.c_int32), ("b_bs", C.c_int64), ("b_kbs", C.c_int64),
                ("D", C.c_void_p), ("ddt", C.c_int32), ("ldd", C.c_int64), ("dbs", C.c_int64),
                ("alpha", C.c_float), ("alpha_ptr", C.c_void_p),
                ("bias_m", C.c_void_p), ("bias_n", C.c_void_p), ("bias_n_bs", C.c_int64), ("m_mod", C.c_int32),
                ("r1_m", C.c_void_p), ("r1_n", C.c_void_p), ("act", C.c_int32),
                ("R", C.c_void_p), ("rdt", C.c_int32), ("ldr", C.c_int64), ("rbs", C.c_int64), ("beta", C.c_float),
                ("mask", C.c_void_p), ("ldmask", C.c_int64), ("maskbs", C.c_int64),
                ("R2", C.c_void_p), ("sm_scale", C.c_void_p), ("sm_dot", C.c_void_p)]


class AttnArgs(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("mode", "B", "N", "C", "tk")] + \
               [(n, C.c_void_p) for n in ("X", "Yp", "dX1", "R2", "out", "T0", "tok", "lse", "a", "aE", "gate_av", "dtok", "dgate",
                                          "da")] + [("invN", C.c_float), ("dT0b", C.c_void_p), ("scratch", C.c_void_p), ("tokpk", C.c_void_p), ("T0pk", C.c_void_p),
                                                               ("dtokpk", C.c_void_p)]


EXPORTS = ["dgsct_test_gemm_fp8", "dgsct_temporal_gate_forward", "dgsct_temporal_gate_backward", "dgsct_test_attn", "dgsct_test_attn_scratch_floats", "dgsct_version", "dgsct_arch", "dgsct_last_error", "dgsct_query", "dgsct_prepare", "dgsct_adapter_forward",
           "dgsct_adapter_forward_ex", "dgsct_adapter_backward", "dgsct_adapter_backward_ex", "dgsct_adapter_backward_ex2", "dgsct_saved_region", "dgsct_test_gemm", "dgsct_test_tune", "dgsct_frame_scale_forward", "dgsct_frame_scale_backward", "dgsct_prof_enable", "dgsct_prof_collect",
           "dgsct_stream_create", "dgsct_stream_destroy", "dgsct_map_pool_forward", "dgsct_map_pool_backward",
           "dgsct_window_attn_forward", "dgsct_window_attn_backward", "dgsct_window_attn_forward_ex", "dgsct_window_attn_backward_ex",
           "dgsct_layer_norm_scratch_floats", "dgsct_layer_norm_forward", "dgsct_layer_norm_backward"]

_PP = C.POINTER(C.c_void_p)


class BwdOpts(C.Structure):
    """dgsct_bwd_opts (include/dgsct.h)"""
    _fields_ = [("flags", C.c_int32), ("dy_residual", C.c_void_p), ("dx_ready_event", C.c_void_p), ("dy_wait_event", C.c_void_p)]


BWD_SKIP_INTO_DX, BWD_NO_JOIN, BWD_HOLD_DY, BWD_ONLY_DY = 1, 2, 4, 8


class Lib:
    """One loaded instance of the C ABI."""

    def __init__(self, path: str = LIB_PATH):
        if not os.path.exists(path):
            raise RuntimeError(f"dg-sct_amd: HIP library not found at {path}; there is no CPU fallback. "
                               f"Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
        self.path = path
        self.c = C.CDLL(path)
        for name in EXPORTS:
            if not hasattr(self.c, name):
                raise RuntimeError(f"dg-sct_amd: {path} does not export {name}")
        c = self.c
        c.dgsct_version.restype = C.c_int
        c.dgsct_arch.restype = C.c_char_p
        c.dgsct_last_error.restype = C.c_char_p
        c.dgsct_query.argtypes = [C.POINTER(AdapterDesc), C.POINTER(Sizes)]
        c.dgsct_prepare.argtypes = [C.POINTER(AdapterDesc), _PP, C.c_void_p, C.c_void_p]
        c.dgsct_adapter_forward.argtypes = [C.POINTER(AdapterDesc), _PP, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        c.dgsct_adapter_backward.argtypes = [C.POINTER(AdapterDesc), _PP, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]
        c.dgsct_adapter_backward_ex.argtypes = c.dgsct_adapter_backward.argtypes + [C.c_void_p, C.c_int]
        c.dgsct_adapter_backward_ex2.argtypes = c.dgsct_adapter_backward.argtypes + [C.c_void_p, C.POINTER(BwdOpts)]
        fa = list(c.dgsct_adapter_forward.argtypes)
        c.dgsct_adapter_forward_ex.argtypes = fa[:5] + [C.c_void_p] + fa[5:] + [C.c_void_p]
        c.dgsct_saved_region.argtypes = [C.POINTER(AdapterDesc), C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64),
                                         C.POINTER(C.c_int64)]
        c.dgsct_test_gemm.argtypes = [C.POINTER(GemmArgs), C.c_void_p]
        c.dgsct_test_attn.argtypes = [C.c_int, C.POINTER(AttnArgs), C.c_void_p]
        c.dgsct_test_gemm_fp8.argtypes = [C.c_int] * 3 + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4
        c.dgsct_temporal_gate_forward.argtypes = [C.c_int, C.c_int, C.c_float] + [C.c_void_p] * 14
        c.dgsct_temporal_gate_backward.argtypes = [C.c_int, C.c_int, C.c_float] + [C.c_void_p] * 20
        c.dgsct_frame_scale_forward.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_float] + [C.c_void_p] * 4
        c.dgsct_frame_scale_backward.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_float] + [C.c_void_p] * 6
        c.dgsct_test_attn_scratch_floats.argtypes = [C.c_int] * 4
        c.dgsct_test_attn_scratch_floats.restype = C.c_int64
        c.dgsct_stream_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        c.dgsct_stream_destroy.argtypes = [C.c_void_p]
        c.dgsct_map_pool_forward.argtypes = [C.c_int] * 4 + [C.c_void_p] * 4
        c.dgsct_map_pool_backward.argtypes = [C.c_int] * 4 + [C.c_void_p] * 6
        c.dgsct_window_attn_forward.argtypes = [C.c_int] * 8 + [C.c_void_p] * 6
        c.dgsct_window_attn_backward.argtypes = [C.c_int] * 8 + [C.c_void_p] * 8
        c.dgsct_window_attn_forward_ex.argtypes = [C.c_int] * 9 + [C.c_void_p] * 6
        c.dgsct_window_attn_backward_ex.argtypes = [C.c_int] * 9 + [C.c_void_p] * 8
        c.dgsct_layer_norm_scratch_floats.argtypes = [C.c_int]
        c.dgsct_layer_norm_scratch_floats.restype = C.c_int64
        c.dgsct_layer_norm_forward.argtypes = [C.c_int, C.c_int64, C.c_int] + [C.c_void_p] * 3 + [C.c_float] + [C.c_void_p] * 5
        c.dgsct_layer_norm_backward.argtypes = [C.c_int, C.c_int64, C.c_int] + [C.c_void_p] * 6 + [C.c_float] + [C.c_void_p] * 5
        if c.dgsct_version() != 100:
            raise RuntimeError("dg-sct_amd: libdgsct version mismatch")

    # ------------------------------------------------------------------
    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.c.dgsct_last_error().decode()}")

    def query(self, desc: AdapterDesc) -> Sizes:
        s = Sizes()
        self._check(self.c.dgsct_query(C.byref(desc), C.byref(s)), "dgsct_query")
        return s

    @staticmethod
    def ptr_table(ptrs: Sequence[Optional[int]]):
        arr = (C.c_void_p * P_COUNT)()
        for i, p in enumerate(ptrs):
            arr[i] = p if p else None
        return arr

    def prepare(self, desc, ptrs, prep: int, stream: int):
        self._check(self.c.dgsct_prepare(C.byref(desc), C.cast(ptrs, _PP), prep, stream), "dgsct_prepare")

    def forward(self, desc, ptrs, prep, X, Y, out, amap, tmap, saved, ws, stream, residual=None, aux_stream=None):
        self._check(self.c.dgsct_adapter_forward_ex(C.byref(desc), C.cast(ptrs, _PP), prep, X, Y, residual, out, amap, tmap,
                                                    saved, ws, stream, aux_stream), "dgsct_adapter_forward")

    def backward(self, desc, ptrs, prep, X, Y, saved, dOut, dMap, dTmap, dX, dY, grads, ws, stream, aux_stream=None,
                 skip_into_dx=False, no_join=False):
        flags = (1 if skip_into_dx else 0) | (2 if no_join else 0)       # DGSCT_BWD_SKIP_INTO_DX | DGSCT_BWD_NO_JOIN
        self._check(self.c.dgsct_adapter_backward_ex(C.byref(desc), C.cast(ptrs, _PP), prep, X, Y, saved, dOut, dMap, dTmap,
                                                     dX, dY, grads, ws, stream, aux_stream, flags),
                    "dgsct_adapter_backward")

    def backward_hold_dy(self, desc, ptrs, prep, X, Y, saved, dOut, dMap, dTmap, dX, grads, ws, stream, aux_stream=None,
                         skip_into_dx=False, no_join=False, dx_ready_event=None):
        """everything of the backward except the product that writes dY (DGSCT_BWD_HOLD_DY); `dx_ready_event` (raw hipEvent_t) is
        recorded on `stream` once dX is complete"""
        o = BwdOpts((BWD_SKIP_INTO_DX if skip_into_dx else 0) | (BWD_NO_JOIN if no_join else 0) | BWD_HOLD_DY, None, dx_ready_event, None)
        self._check(self.c.dgsct_adapter_backward_ex2(C.byref(desc), C.cast(ptrs, _PP), prep, X, Y, saved, dOut, dMap, dTmap,
                                                      dX, None, grads, ws, stream, aux_stream, C.byref(o)),
                    "dgsct_adapter_backward_ex2 (HOLD_DY)")

    def backward_only_dy(self, desc, ptrs, prep, dY, ws, stream, dy_residual=None, dy_wait_event=None):
        """the held-back product: dY = dy_residual + d adapter / dY, behind `dy_wait_event` (DGSCT_BWD_ONLY_DY)"""
        o = BwdOpts(BWD_ONLY_DY, dy_residual, None, dy_wait_event)
        self._check(self.c.dgsct_adapter_backward_ex2(C.byref(desc), C.cast(ptrs, _PP), prep, None, None, None, None, None, None,
                                                      None, dY, None, ws, stream, None, C.byref(o)),
                    "dgsct_adapter_backward_ex2 (ONLY_DY)")

    def saved_regions(self, desc):
        out = {}
        i = 0
        name = C.create_string_buffer(64)
        off, nb = C.c_int64(), C.c_int64()
        while self.c.dgsct_saved_region(C.byref(desc), i, name, 64, C.byref(off), C.byref(nb)) == 0:
            out[name.value.decode()] = (off.value, nb.value)
            i += 1
        return out

    def stream_create(self, priority_class: int) -> int:
        """raw hipStream_t in its own priority class (-1 high, 0 normal, +1 low) = its own hardware-queue pool"""
        out = C.c_void_p()
        self._check(self.c.dgsct_stream_create(int(priority_class), C.byref(out)), "dgsct_stream_create")
        return int(out.value or 0)

    def map_pool_forward(self, dtype: int, BT: int, N: int, C_: int, F: int, amap: int, pooled: int, stream: int):
        self._check(self.c.dgsct_map_pool_forward(dtype, BT, N, C_, F, amap, pooled, stream), "dgsct_map_pool_forward")

    def map_pool_backward(self, dtype: int, BT: int, N: int, C_: int, F: int, amap: int, dpooled: int, dF, dmap, stream: int):
        self._check(self.c.dgsct_map_pool_backward(dtype, BT, N, C_, F, amap, dpooled, dF, dmap, stream), "dgsct_map_pool_backward")

    def window_attn_forward(self, geom, qkv, bm, scale, out, lse, stream, flags=0):
        """geom = (B, H, W, ws, shift, heads, hd, nwm); pointers as ints; flags: WATTN_COSINE (include/dgsct.h: dgsct_window_attn_forward_ex)"""
        self._check(self.c.dgsct_window_attn_forward_ex(*geom, flags, qkv, bm, scale, out, lse, stream), "dgsct_window_attn_forward")

    def window_attn_backward(self, geom, qkv, bm, scale, out, lse, dout, dqkv, stream, flags=0):
        self._check(self.c.dgsct_window_attn_backward_ex(*geom, flags, qkv, bm, scale, out, lse, dout, dqkv, stream), "dgsct_window_attn_backward")

    def layer_norm_forward(self, dtype, rows, C_, x, w, b, eps, residual, out, mu, rstd, stream):
        self._check(self.c.dgsct_layer_norm_forward(dtype, rows, C_, x, w, b, eps, residual, out, mu, rstd, stream), "dgsct_layer_norm_forward")

    def layer_norm_backward(self, dtype, rows, C_, dout, x, w, b, mu, rstd, eps, dx, dw, db, scratch, stream):
        self._check(self.c.dgsct_layer_norm_backward(dtype, rows, C_, dout, x, w, b, mu, rstd, eps, dx, dw, db, scratch, stream),
                    "dgsct_layer_norm_backward")

    def prof_enable(self, on: bool):
        self.c.dgsct_prof_enable(int(on))

    def prof_collect(self):
        n, ms, fl = C.c_int64(), C.c_double(), C.c_double()
        self.c.dgsct_prof_collect(C.byref(n), C.byref(ms), C.byref(fl))
        return n.value, ms.value, fl.value

    def frame_scale_forward(self, dtype, rows, inner, gamma, x, g, y, stream):
        self._check(self.c.dgsct_frame_scale_forward(dtype, rows, inner, gamma, x, g, y, stream), "dgsct_frame_scale_forward")

    def frame_scale_backward(self, dtype, rows, inner, gamma, x, g, dy, dx, dg, stream):
        self._check(self.c.dgsct_frame_scale_backward(dtype, rows, inner, gamma, x, g, dy, dx, dg, stream), "dgsct_frame_scale_backward")

    def temporal_gate_forward(self, R, D, gamma, *ptrs):
        self._check(self.c.dgsct_temporal_gate_forward(R, D, gamma, *ptrs), "dgsct_temporal_gate_forward")

    def temporal_gate_backward(self, R, D, gamma, *ptrs):
        self._check(self.c.dgsct_temporal_gate_backward(R, D, gamma, *ptrs), "dgsct_temporal_gate_backward")

    def test_gemm_fp8(self, M, N, K, A, W, bias, relu, D, w8, scale, stream):
        self._check(self.c.dgsct_test_gemm_fp8(M, N, K, A, W, bias, int(relu), D, w8, scale, stream), "dgsct_test_gemm_fp8")

    def test_attn(self, op: int, args: "AttnArgs", stream: int):
        self._check(self.c.dgsct_test_attn(int(op), C.byref(args), stream), "dgsct_test_attn")

    def test_tune(self, key: str, value: int) -> int:
        self.c.dgsct_test_tune.argtypes = [C.c_char_p, C.c_int]
        self.c.dgsct_test_tune.restype = C.c_int
        return int(self.c.dgsct_test_tune(key.encode(), int(value)))

    def test_gemm(self, args: GemmArgs, stream: int):
        self._check(self.c.dgsct_test_gemm(C.byref(args), stream), "dgsct_test_gemm")


_DEFAULT: Optional[Lib] = None


def default_lib() -> Lib:
    """The product library (hand-written gfx950 kernels).  Raises if it has not been built."""
    global _DEFAULT
    if _DEFAULT is None:
        _DEFAULT = Lib(LIB_PATH)
    return _DEFAULT
