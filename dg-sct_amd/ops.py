"""Functional layer over the C ABI: buffer ownership (torch allocates, the library only borrows),
stream selection, and the autograd.Function that replaces the reference's ~45-op autograd graph
(DG-SCT/AVE/nets/net_trans.py:552-674) by one forward and one backward library call."""
from __future__ import annotations

import contextlib
import dataclasses
import os
import threading
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import AdapterDesc, Lib, P_COUNT, P_INDEX, PARAM_NAMES


@dataclasses.dataclass(frozen=True)
class AdapterSpec:
    """Static description of one adapter (everything in dgsct_adapter_desc except BT/dtype/training)."""
    N: int
    C: int
    No: int
    Co: int
    tk: int = 32
    r: int = 8
    g: int = 2
    use_bn: bool = True
    use_gate: bool = True
    ln_before: bool = True
    ln_post: bool = True
    gate_before_ln_post: bool = False
    remap: str = "conv"          # "conv" | "bicubic"
    alpha: float = 0.3
    beta: float = 0.05
    gamma: float = 0.0
    temporal: bool = False
    T: int = 10
    eps: float = 1e-5
    bn_momentum: float = 0.1
    fp8: bool = False            # bf16 + fp8 (e4m3) MFMA operands for fc / fc_affine_video_1 / fc_affine_video_2 (BASELINE configs[4])

    def desc(self, BT: int, dtype: torch.dtype, training: bool) -> AdapterDesc:
        key = (self, BT, dtype, bool(training))
        d = _DESC_CACHE.get(key)
        if d is not None:
            return d
        d = _DESC_CACHE[key] = self._make_desc(BT, dtype, training)
        return d

    def _make_desc(self, BT: int, dtype: torch.dtype, training: bool) -> AdapterDesc:
        d = AdapterDesc()
        d.BT, d.T, d.N, d.C, d.No, d.Co, d.tk, d.r, d.g = BT, self.T, self.N, self.C, self.No, self.Co, self.tk, self.r, self.g
        d.dtype = (_lib.BF16_FP8 if self.fp8 else _lib.BF16) if dtype == torch.bfloat16 else _lib.F32
        d.remap = _lib.REMAP_CONV if self.remap == "conv" else _lib.REMAP_FIXED
        d.use_bn, d.use_gate, d.ln_before, d.ln_post = int(self.use_bn), int(self.use_gate), int(self.ln_before), int(self.ln_post)
        d.gate_before_ln_post, d.temporal, d.training = int(self.gate_before_ln_post), int(self.temporal), int(training)
        d.alpha, d.beta, d.gamma, d.eps, d.bn_momentum = self.alpha, self.beta, self.gamma, self.eps, self.bn_momentum
        return d


_DESC_CACHE: Dict[Tuple, AdapterDesc] = {}
_SIZE_CACHE: Dict[Tuple, object] = {}


def _sizes(lib: Lib, d: AdapterDesc):
    """dgsct_query results are pure functions of the descriptor and of the two layout switches the library reads from the environment
    at every call (test hooks: DGSCT_WIDE_ATTN, DGSCT_VQ1_DW): ask once per (library, descriptor, switches)."""
    key = (id(lib), id(d), os.environ.get("DGSCT_WIDE_ATTN"), os.environ.get("DGSCT_VQ1_DW"))
    s = _SIZE_CACHE.get(key)
    if s is None:
        s = _SIZE_CACHE[key] = lib.query(d)
    return s


# ---------------------------------------------------------------------------------------------
# scratch: one growing buffer per (device, stream); calls on a stream serialise, so sharing is safe.
_WS: Dict[Tuple, torch.Tensor] = {}
_WS_LOCK = threading.Lock()


def _stream_of(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


_AUX: Dict[Tuple, "torch.cuda.Stream"] = {}
USE_AUX_STREAM = os.environ.get("DGSCT_NO_AUX", "0") != "1"


COMPUTE_PRIORITY_CLASS = int(os.environ.get("DGSCT_COMPUTE_PRIORITY", "-1"))      # second adapter stream (AdapterStack)
# aux streams carry work nothing waits for until the end of the call (weight gradients, dX): LOW priority, so the
# dispatcher prefers the kernels of the dependency chain whenever both are runnable (76.8 vs 77.8 ms per step), and a
# third queue pool, so main (normal) / second adapter stream (high) / aux (low) can never share a hardware queue
AUX_PRIORITY_CLASS = int(os.environ.get("DGSCT_AUX_PRIORITY", "1"))


def priority_stream(lib: Lib, device: torch.device, priority_class: int) -> "torch.cuda.Stream":
    """A torch handle on a library-created HIP stream of the given priority class (-1 high / 0 / +1 low).  Streams of
    different classes never share a hardware queue (include/dgsct.h, dgsct_stream_create), whatever else (RCCL, torch's
    stream pool) has created streams before."""
    with torch.cuda.device(device):
        raw = lib.stream_create(priority_class)
    return torch.cuda.ExternalStream(raw, device=device)


_SIDE: Dict[int, "torch.cuda.Stream"] = {}


def side_stream(lib: Lib, device: torch.device) -> "torch.cuda.Stream":
    """THE second adapter stream of a device (AdapterStack runs the audio adapter of a pair on it).  One per device for the
    whole process, like the aux streams below: every extra stream an instance created would land on some hardware queue
    of its pool, and two busy streams on one queue serialise (measured: a second AdapterStack with its own streams ran
    its step in 124 ms instead of 73)."""
    with _WS_LOCK:
        s = _SIDE.get(device.index)
        if s is None:
            s = _SIDE[device.index] = priority_stream(lib, device, COMPUTE_PRIORITY_CLASS)
    return s


def _aux_stream(lib: Lib, t: torch.Tensor, stream: int) -> Optional[int]:
    """one low-priority side stream per (device, caller stream): weight gradients overlap the data-gradient chain on it"""
    if not (USE_AUX_STREAM and t.is_cuda):
        return None
    key = (t.device.index, stream)
    with _WS_LOCK:
        s = _AUX.get(key)
        if s is None:
            s = _AUX[key] = priority_stream(lib, t.device, AUX_PRIORITY_CLASS)
    return s.cuda_stream


def _workspace(device: torch.device, stream: int, nbytes: int, slot: int = 0) -> torch.Tensor:
    key = (device.type, device.index, stream, slot)
    with _WS_LOCK:
        buf = _WS.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(int(nbytes * 1.0) + 256, dtype=torch.uint8, device=device)
            _WS[key] = buf
    return buf


def release_workspaces():
    drain_aux()
    with _WS_LOCK:
        _WS.clear()


# ---- deferred join of the weight-gradient (aux) stream (round 5) -------------------------------------------------------------
# dgsct_adapter_backward_ex(DGSCT_BWD_NO_JOIN) returns with its weight gradients still running on the aux stream: the chain of the
# NEXT backward call on the same stream starts at once instead of waiting for them (0.9 ms of the 50 ms AVE step, most of it the
# remap weight gradient of stage 0).  What that leaves to the caller is done here:
#   * two workspaces per stream, used alternately: call k + 1 never overwrites what call k's aux kernels still read;
#   * an event recorded on the aux stream behind call k is waited for by the main stream at the END of call k + 1 (it has long
#     completed by then); only then are call k's buffers (saved activations, inputs, its workspace) let go;
#   * the gradients of the LAST calls are ordered by drain_aux(): queued as an autograd end-of-backward callback (the stream that
#     called backward() waits for every pending aux event), and called by the data-parallel reducer before it launches a collective.
# Only the flat-parameter path defers (autograd adopts the gradient buffer without reading it); DGSCT_DEFER_AUX=0 switches it off.
DEFER_AUX_JOIN = os.environ.get("DGSCT_DEFER_AUX", "1") != "0"
_PENDING: Dict[Tuple, Tuple] = {}          # (device index, stream) -> (event on aux, tensors kept alive)
_SLOT: Dict[Tuple, int] = {}


def drain_aux(device: Optional[torch.device] = None):
    """Order the CURRENT stream -- and each stream whose backward calls deferred their join -- after every weight-gradient stream that
    still has work in flight.  Cheap when nothing is pending."""
    if not _PENDING:
        return
    want = None if device is None else (device.index if device.index is not None else torch.cuda.current_device())
    with _WS_LOCK:
        items = [(k, v) for k, v in _PENDING.items() if want is None or k[0] == want]
        for k, _ in items:
            del _PENDING[k]
    for (dev, stream), (ev, keep) in items:
        torch.cuda.current_stream(dev).wait_event(ev)
        prod = torch.cuda.ExternalStream(stream, device=torch.device("cuda", dev)) if stream else torch.cuda.default_stream(dev)
        prod.wait_event(ev)
        del keep


def _wait_pending(dev: torch.device, stream: int):
    """A call that does NOT defer (forward, joined backward) is about to reuse workspace slot 0 of this stream: if the previous
    backward on the stream deferred its join, its weight-gradient kernels on the aux stream may still be reading that workspace
    (stream_fork only orders aux behind main, never the reverse).  Order the current stream behind that call's aux event first, and
    let its buffers go.  A dictionary lookup when nothing is pending."""
    if not _PENDING:
        return
    key = (dev.index, stream)
    with _WS_LOCK:
        prev = _PENDING.pop(key, None)
    if prev is not None:
        torch.cuda.current_stream(dev).wait_event(prev[0])
        del prev


def _queue_drain():
    """drain_aux() when the autograd engine has run the last node of this backward pass.  Queued by EVERY deferring call: backward nodes
    run on the engine's per-device worker threads and the callbacks on whichever thread finishes the graph task, so a once-per-pass
    flag cannot live in thread-local storage; a drain that finds nothing pending costs a dictionary lookup.  The engine runs final
    callbacks on the stream that was current around the user's backward() call, which is the stream that reads the gradients next."""
    try:
        torch.autograd.Variable._execution_engine.queue_callback(drain_aux)
    except RuntimeError:                   # not inside a backward pass (raw calls from tests / tools): the caller drains
        pass


def _dev_guard(t: torch.Tensor):
    """The library launches on the CURRENT HIP device (kernel launches, occupancy queries, event creation): make it the
    device that owns the tensors and streams of this call (a module on cuda:1 called while cuda:0 is current)."""
    return torch.cuda.device(t.device) if t.is_cuda else contextlib.nullcontext()


def _ptrs(params: List[Optional[torch.Tensor]]):
    return Lib.ptr_table([p.data_ptr() if p is not None else None for p in params])


def check_param(name: str, p: Optional[torch.Tensor], device) -> Optional[torch.Tensor]:
    if p is None:
        return None
    if p.dtype != torch.float32 or not p.is_contiguous() or p.device != device:
        raise RuntimeError(f"dg-sct_amd: parameter {name} must be a contiguous fp32 tensor on {device}")
    return p


def prepare(lib: Lib, spec: AdapterSpec, params: List[Optional[torch.Tensor]], dtype: torch.dtype, device) -> torch.Tensor:
    """fp32 master parameters -> MFMA-operand copies + derived bias vectors (dgsct_prepare)."""
    d = spec.desc(spec.T, dtype, False)
    sz = _sizes(lib, d)
    prep = torch.empty(max(int(sz.prep_bytes), 256), dtype=torch.uint8, device=device)
    if os.environ.get("DGSCT_POISON", "0") == "1":
        prep.fill_(0xFF)
    some = next(p for p in params if p is not None)
    with _dev_guard(some):
        lib.prepare(d, _ptrs(params), prep.data_ptr(), _stream_of(some))
    return prep


_POISON = os.environ.get("DGSCT_POISON", "0") == "1"      # test hook: fill every buffer handed to the library with NaN


def _poison(*tensors):
    """DGSCT_POISON=1: every output / scratch / saved buffer starts as NaN bit patterns, so a kernel that reads memory it
    (or an earlier kernel of the call) did not write shows up as NaN in the results instead of depending on what the
    caching allocator happened to hand out."""
    for t in tensors:
        if t is not None:
            t.view(torch.uint8).fill_(0xFF)


def _mark_stream_use(*tensors):
    """Tell torch's caching allocator that these tensors are read on the CURRENT stream.  With AdapterStack's two adapter
    streams an input map (or an incoming gradient) is usually allocated on the other stream; without this, the block can be
    handed out again on its home stream while kernels enqueued here are still reading it."""
    for t in tensors:
        if t is not None and t.is_cuda:
            t.record_stream(torch.cuda.current_stream(t.device))


def raw_forward(lib: Lib, spec: AdapterSpec, params, prep, X: torch.Tensor, Y: torch.Tensor, training: bool,
                residual: Optional[torch.Tensor] = None):
    """X [BT,N,C], Y [BT,No,Co] contiguous, same dtype (fp32|bf16).  Returns (out, map, tmap, saved, desc).
    residual [BT,N,C] (may be X itself): out = residual + adapter(X, Y), the add fused into the last kernel."""
    BT = X.shape[0]
    if X.shape != (BT, spec.N, spec.C) or Y.shape != (BT, spec.No, spec.Co):
        raise RuntimeError(f"dg-sct_amd: expected X [BT,{spec.N},{spec.C}] and Y [BT,{spec.No},{spec.Co}], got "
                           f"{tuple(X.shape)} and {tuple(Y.shape)}")
    if X.dtype != Y.dtype or X.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("dg-sct_amd: X and Y must both be float32 or both bfloat16")
    if residual is not None and (residual.shape != X.shape or residual.dtype != X.dtype or not residual.is_contiguous()):
        raise RuntimeError("dg-sct_amd: residual must be a contiguous tensor of X's shape and dtype")
    _mark_stream_use(X, Y, residual, prep)
    d = spec.desc(BT, X.dtype, training)
    sz = _sizes(lib, d)
    dev = X.device
    out = torch.empty_like(X)
    amap = torch.empty(BT, spec.N, dtype=torch.float32, device=dev)
    tmap = torch.empty(BT, dtype=torch.float32, device=dev) if spec.temporal else None
    saved = torch.empty(int(sz.saved_bytes), dtype=torch.uint8, device=dev)
    stream = _stream_of(X)
    _wait_pending(dev, stream)                 # slot 0 may still be read by a deferred backward's weight gradients
    ws = _workspace(dev, stream, int(sz.ws_fwd_bytes))
    if _POISON:
        _poison(out, amap, tmap, saved, ws)
    with _dev_guard(X):
        lib.forward(d, _ptrs(params), prep.data_ptr(), X.data_ptr(), Y.data_ptr(), out.data_ptr(), amap.data_ptr(),
                    tmap.data_ptr() if tmap is not None else None, saved.data_ptr(), ws.data_ptr(), stream,
                    residual.data_ptr() if residual is not None else None, _aux_stream(lib, X, stream))
    return out, amap, tmap, saved, d


def raw_backward(lib: Lib, spec: AdapterSpec, d: AdapterDesc, params, prep, X, Y, saved, dOut, dMap, dTmap, flat_out=False,
                 skip_into_dx=False, defer_join=False, hold_dy=False, dx_event=None, slot_base=0):
    """defer_join: the call returns without joining its weight-gradient stream (see DEFER_AUX_JOIN above); `grads` is complete only
    after drain_aux() -- dX / dY are ordered on the current stream as always.
    hold_dy: everything but the product that writes dY (pair backward below): returns (dX, ws, grads) -- `ws` is what
    raw_backward_dy() needs; dx_event (raw hipEvent_t) is recorded on the call's stream once dX is complete."""
    _mark_stream_use(X, Y, saved, dOut, dMap, dTmap, prep)
    sz = _sizes(lib, d)
    dev = X.device
    dX = torch.empty_like(X)
    dY = None if hold_dy else torch.empty_like(Y)
    grads = torch.empty(int(sz.grad_floats), dtype=torch.float32, device=dev)
    stream = _stream_of(X)
    aux = _aux_stream(lib, X, stream)
    defer = bool(defer_join and DEFER_AUX_JOIN and aux is not None)
    key = (dev.index, stream)
    slot = 0
    if defer:
        slot = _SLOT[key] = 1 - _SLOT.get(key, 1)
    else:
        _wait_pending(dev, stream)             # a joined call takes slot 0 whatever the deferred call before it used
        _SLOT.pop(key, None)                   # (the next deferring call starts at slot 0 again: nothing is pending then)
    ws = _workspace(dev, stream, int(sz.ws_bwd_bytes), slot_base + slot)
    if _POISON:
        _poison(dX, dY, grads, ws)
    with _dev_guard(X):
        if hold_dy:
            lib.backward_hold_dy(d, _ptrs(params), prep.data_ptr(), X.data_ptr(), Y.data_ptr(), saved.data_ptr(), dOut.data_ptr(),
                                 dMap.data_ptr() if dMap is not None else None, dTmap.data_ptr() if dTmap is not None else None,
                                 dX.data_ptr(), grads.data_ptr(), ws.data_ptr(), stream, aux, skip_into_dx, defer, dx_event)
        else:
            lib.backward(d, _ptrs(params), prep.data_ptr(), X.data_ptr(), Y.data_ptr(), saved.data_ptr(), dOut.data_ptr(),
                         dMap.data_ptr() if dMap is not None else None, dTmap.data_ptr() if dTmap is not None else None,
                         dX.data_ptr(), dY.data_ptr(), grads.data_ptr(), ws.data_ptr(), stream, aux, skip_into_dx, defer)
        if defer:
            ev = torch.cuda.Event()
            ev.record(_AUX[key])                                       # behind everything this call put on its aux stream
            with _WS_LOCK:
                prev = _PENDING.get(key)
                # (NOT `grads`: autograd adopts the flat gradient only while nothing else references it -- a second reference makes
                #  AccumulateGrad CLONE it on the calling stream, i.e. read it before the aux stream has finished writing)
                _PENDING[key] = (ev, (saved, X, Y, dOut, dMap, dTmap, prep, ws))
            if prev is not None:
                torch.cuda.current_stream(dev).wait_event(prev[0])     # the call before this one: long done; its buffers go now
                del prev
            _queue_drain()
    if hold_dy:
        return dX, ws, grads
    if flat_out:
        return dX, dY, grads
    lay = grad_layout(lib, d)
    per_param: List[Optional[torch.Tensor]] = [grads[off:off + n] if off >= 0 else None for off, n in lay]
    return dX, dY, per_param


def raw_backward_dy(lib: Lib, d: AdapterDesc, params, prep, ws, Y, residual=None, wait_event=None):
    """The product raw_backward(hold_dy=True) held back, on the current stream: dY = residual + d adapter / dY, issued behind
    `wait_event` (raw hipEvent_t).  `ws` is the workspace that call returned; Y only gives the shape / dtype / device."""
    _mark_stream_use(residual, prep)
    dY = torch.empty_like(Y)
    if _POISON:
        _poison(dY)
    with _dev_guard(Y):
        lib.backward_only_dy(d, _ptrs(params), prep.data_ptr(), dY.data_ptr(), ws.data_ptr(), _stream_of(Y),
                             residual.data_ptr() if residual is not None else None, wait_event)
    return dY


def grad_layout(lib: Lib, d: AdapterDesc):
    """[(float offset, numel)] per C-ABI parameter slot (offset -1: no gradient); the layout of the flat gradient buffer,
    which is also the layout of a flattened parameter (VisualAdapter.flatten_parameters)."""
    sz = _sizes(lib, d)
    lay = _GRAD_LAYOUT.get(id(sz))
    if lay is None:
        lay = _GRAD_LAYOUT[id(sz)] = [(int(sz.grad_offset[i]), int(sz.grad_numel[i])) for i in range(P_COUNT)]
    return lay


_GRAD_LAYOUT: Dict[int, list] = {}


class _AdapterFn(torch.autograd.Function):
    """forward(X, Y, *params) -> (out, map[, tmap]); one library call each way."""

    @staticmethod
    def forward(ctx, lib, spec, training, prep, skip, res, X, Y, *params):
        plist = list(params)
        out, amap, tmap, saved, d = raw_forward(lib, spec, plist, prep, X, Y, training, X if skip else res)
        ctx.lib, ctx.spec, ctx.desc, ctx.prep = lib, spec, d, prep
        ctx.skip, ctx.has_res = bool(skip), res is not None
        ctx.saved_buf = saved
        ctx.save_for_backward(X, Y, *[p for p in plist if p is not None])
        ctx.present = [p is not None for p in plist]
        ctx.shapes = [tuple(p.shape) if p is not None else None for p in plist]
        ctx.set_materialize_grads(False)      # unused outputs (map of all but the last layer, tmap) arrive as None, not zeros
        if tmap is None:
            tmap = torch.empty(0, device=X.device)
        return out, amap, tmap

    @staticmethod
    def backward(ctx, dOut, dMap, dTmap):
        if ctx.saved_buf is None:
            raise RuntimeError("dg-sct_amd: backward through an adapter call twice is not supported "
                               "(the saved-activation buffer is consumed in place)")
        tensors = ctx.saved_tensors
        X, Y = tensors[0], tensors[1]
        it = iter(tensors[2:])
        plist = [next(it) if pres else None for pres in ctx.present]
        spec = ctx.spec
        if dOut is None:
            dOut = torch.zeros_like(X)
        dOut = dOut.contiguous()
        if dOut.dtype != X.dtype:
            dOut = dOut.to(X.dtype)
        dMap = dMap.contiguous().float() if dMap is not None else None
        dTm = dTmap.contiguous().float() if (spec.temporal and dTmap is not None and dTmap.numel()) else None
        dX, dY, grads = raw_backward(ctx.lib, spec, ctx.desc, plist, ctx.prep, X, Y, ctx.saved_buf, dOut, dMap, dTm,
                                     skip_into_dx=ctx.skip)
        ctx.saved_buf = None
        pg = []
        for i, g in enumerate(grads):
            if not ctx.present[i]:
                pg.append(None)
            elif g is None:
                pg.append(None)
            else:
                pg.append(g.view(ctx.shapes[i]))
        return (None, None, None, None, None, dOut if ctx.has_res else None, dX, dY, *pg)


class _AdapterFlatFn(torch.autograd.Function):
    """Same call, but every trainable tensor of the adapter lives in ONE flat fp32 parameter laid out like the
    library's gradient buffer: the backward returns that buffer as the parameter's gradient -- one tensor that owns its
    storage, so autograd adopts it without copying (1 AccumulateGrad instead of ~30 clones), the optimizer steps 48
    tensors instead of ~1900, and data-parallel all-reduce runs on it in place."""

    @staticmethod
    def forward(ctx, lib, spec, training, prep, plist, skip, res, X, Y, flat):
        out, amap, tmap, saved, d = raw_forward(lib, spec, plist, prep, X, Y, training, X if skip else res)
        ctx.lib, ctx.spec, ctx.desc, ctx.prep, ctx.plist = lib, spec, d, prep, plist
        ctx.skip, ctx.has_res = bool(skip), res is not None
        ctx.saved_buf = saved
        ctx.flat = flat
        _count_use(flat)
        ctx.save_for_backward(X, Y)
        ctx.set_materialize_grads(False)
        if tmap is None:
            tmap = torch.empty(0, device=X.device)
        return out, amap, tmap

    @staticmethod
    def backward(ctx, dOut, dMap, dTmap):
        if ctx.saved_buf is None:
            raise RuntimeError("dg-sct_amd: backward through an adapter call twice is not supported "
                               "(the saved-activation buffer is consumed in place)")
        X, Y = ctx.saved_tensors
        # The join of the weight-gradient stream is deferred only when autograd will ADOPT the returned buffer untouched: the
        # parameter has no gradient yet (else AccumulateGrad adds the new one into it on this stream, right now) and no tensor
        # hook wants to see it.  Gradient accumulation over several backward passes therefore joins from the second pass on.
        flat = ctx.flat
        can_defer = _may_adopt(flat)
        _use_done(flat)
        # (a DataParallel replica's flat tensor is NOT a leaf: its gradient is consumed by the broadcast's backward at once)
        ctx.flat = None
        spec = ctx.spec
        if dOut is None:
            dOut = torch.zeros_like(X)
        dOut = dOut.contiguous()
        if dOut.dtype != X.dtype:
            dOut = dOut.to(X.dtype)
        dMap = dMap.contiguous().float() if dMap is not None else None
        dTm = dTmap.contiguous().float() if (spec.temporal and dTmap is not None and dTmap.numel()) else None
        dX, dY, gflat = raw_backward(ctx.lib, spec, ctx.desc, ctx.plist, ctx.prep, X, Y, ctx.saved_buf, dOut, dMap, dTm,
                                     flat_out=True, skip_into_dx=ctx.skip, defer_join=can_defer)
        ctx.saved_buf = None
        return None, None, None, None, None, None, dOut if ctx.has_res else None, dX, dY, gflat


def _may_adopt(flat) -> bool:
    """True when autograd will ADOPT the gradient buffer returned for this flat parameter without reading it on the calling stream --
    the condition for leaving the weight-gradient (aux) stream un-joined.  It reads the buffer (clone / add / hook) when: the
    parameter already has a gradient (accumulation), a tensor hook or a post-accumulate-grad hook other than this package's reducer
    wants to see it (torch DDP's reducer hooks read .grad mid-backward; dp.GradAllReducer drains the aux streams first and marks its
    hooks), grad mode is on (create_graph=True: AccumulateGrad clones), or the same parameter is used by more than one call of the
    graph (the engine's input buffer SUMS the two buffers before accumulation: `_dgsct_uses`, counted in forward)."""
    if not (isinstance(flat, torch.Tensor) and flat.is_leaf and flat.grad is None):
        return False
    if getattr(flat, "_backward_hooks", None) or torch.is_grad_enabled():
        return False
    post = getattr(flat, "_post_accumulate_grad_hooks", None)
    if post and any(not getattr(h, "_dgsct_drains_aux", False) for h in post.values()):
        return False
    return getattr(flat, "_dgsct_uses", 1) <= 1


def _count_use(flat):
    """forward side of `_may_adopt`: how many calls of the graph being built read this flat parameter.  The count stands until the
    LAST of those calls has run its backward (every one of them must join); a forward whose graph is dropped without a backward
    leaves the count high, which only costs the deferral (safe side)."""
    if isinstance(flat, torch.Tensor) and flat.requires_grad and torch.is_grad_enabled():
        flat._dgsct_uses = getattr(flat, "_dgsct_uses", 0) + 1
        flat._dgsct_left = getattr(flat, "_dgsct_left", 0) + 1


def _use_done(flat):
    if isinstance(flat, torch.Tensor) and getattr(flat, "_dgsct_left", 0) > 0:
        flat._dgsct_left -= 1
        if flat._dgsct_left == 0:
            flat._dgsct_uses = 0


PAIR_BACKWARD = os.environ.get("DGSCT_NO_PAIR", "0") != "1"
_PAIR_EVENTS: Dict[int, list] = {}
_PAIR_RING = 16


def _pair_events(dev: torch.device):
    """two recorded-once events of a small per-device ring (a torch event has no handle before its first record); the library
    re-records them in the middle of the two calls of a pair"""
    ring = _PAIR_EVENTS.get(dev.index)
    if ring is None:
        ring = _PAIR_EVENTS[dev.index] = [[], 0]
        cur = torch.cuda.current_stream(dev)
        for _ in range(_PAIR_RING):
            e = torch.cuda.Event()
            e.record(cur)
            ring[0].append(e)
    i = ring[1]
    ring[1] = (i + 2) % _PAIR_RING
    return ring[0][i], ring[0][i + 1]


def _cotangent(dOut, X):
    if dOut is None:
        return torch.zeros_like(X)
    dOut = dOut.contiguous()
    return dOut if dOut.dtype == X.dtype else dOut.to(X.dtype)


class _PairFlatFn(torch.autograd.Function):
    """The audio and the visual adapter of one position (net_trans.py:891-906) as ONE autograd node:
        f_a' = f_a + audio_adapter(f_a, f_v),   f_v' = f_v + vis_adapter(f_v, f_a)
    so that  d f_a = dX(audio) + dY(visual)  and  d f_v = dX(visual) + dY(audio)  are formed by the products that write dY (their
    epilogues read the other call's dX: include/dgsct.h, dgsct_adapter_backward_ex2) instead of by autograd's accumulation -- one
    pass over [BT][N][C] per adapter call less, on the critical path between two positions.  `side`: the second adapter stream
    (None: both calls on the current stream, in order -- the host-emulated library of the CPU tests)."""

    @staticmethod
    def forward(ctx, lib, side, call_a, call_v, f_a, f_v, flat_a, flat_v):
        # call_x = (spec, training, prep, plist) of VisualAdapter._token_call
        def one(call, X, Y):
            spec, training, prep, plist = call
            out, amap, _, saved, d = raw_forward(lib, spec, plist, prep, X, Y, training, X)
            return out, amap, saved, d
        if side is not None:
            main = torch.cuda.current_stream(f_a.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                oa, ma, sa, da = one(call_a, f_a, f_v)
            ov, mv, sv, dv = one(call_v, f_v, f_a)
            main.wait_stream(side)
            oa.record_stream(main)
            ma.record_stream(main)
        else:
            oa, ma, sa, da = one(call_a, f_a, f_v)
            ov, mv, sv, dv = one(call_v, f_v, f_a)
        ctx.lib, ctx.side = lib, side
        ctx.calls = (call_a, call_v)
        ctx.descs = (da, dv)
        ctx.saved_bufs = (sa, sv)
        ctx.flats = (flat_a, flat_v)
        _count_use(flat_a)
        _count_use(flat_v)
        ctx.save_for_backward(f_a, f_v)
        ctx.set_materialize_grads(False)
        return oa, ma, ov, mv

    @staticmethod
    def backward(ctx, dOa, dMa, dOv, dMv):
        if ctx.saved_bufs is None:
            raise RuntimeError("dg-sct_amd: backward through an adapter call twice is not supported "
                               "(the saved-activation buffer is consumed in place)")
        f_a, f_v = ctx.saved_tensors
        lib, side = ctx.lib, ctx.side
        (spec_a, _, prep_a, pl_a), (spec_v, _, prep_v, pl_v) = ctx.calls
        d_a, d_v = ctx.descs
        s_a, s_v = ctx.saved_bufs
        defer = [_may_adopt(fl) for fl in ctx.flats]                    # (as in _AdapterFlatFn.backward)
        for fl in ctx.flats:
            _use_done(fl)
        ctx.flats = ctx.saved_bufs = None
        dOa, dOv = _cotangent(dOa, f_a), _cotangent(dOv, f_v)
        dMa = dMa.contiguous().float() if dMa is not None else None
        dMv = dMv.contiguous().float() if dMv is not None else None

        def main_part(spec, d, pl, prep, X, Y, saved, dO, dM, can_defer, ev, slot_base=0):
            return raw_backward(lib, spec, d, pl, prep, X, Y, saved, dO, dM, None, flat_out=True, skip_into_dx=True,
                                defer_join=can_defer, hold_dy=True, dx_event=ev, slot_base=slot_base)

        if side is not None:
            dev = f_a.device
            ev_a, ev_v = _pair_events(dev)
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                dXa, ws_a, g_a = main_part(spec_a, d_a, pl_a, prep_a, f_a, f_v, s_a, dOa, dMa, defer[0], ev_a.cuda_event)
            dXv, ws_v, g_v = main_part(spec_v, d_v, pl_v, prep_v, f_v, f_a, s_v, dOv, dMv, defer[1], ev_v.cuda_event)
            with torch.cuda.stream(side):                               # d f_v = dX(visual) + dY(audio call)
                df_v = raw_backward_dy(lib, d_a, pl_a, prep_a, ws_a, f_v, dXv, ev_v.cuda_event)
            df_a = raw_backward_dy(lib, d_v, pl_v, prep_v, ws_v, f_a, dXa, ev_a.cuda_event)     # d f_a = dX(audio) + dY(visual call)
            main.wait_stream(side)
            df_v.record_stream(main)
        else:
            dXa, ws_a, g_a = main_part(spec_a, d_a, pl_a, prep_a, f_a, f_v, s_a, dOa, dMa, defer[0], None)
            # (one stream = one workspace per slot: the second call must not overwrite what the first one's dY product still reads)
            dXv, ws_v, g_v = main_part(spec_v, d_v, pl_v, prep_v, f_v, f_a, s_v, dOv, dMv, defer[1], None, slot_base=2)
            df_v = raw_backward_dy(lib, d_a, pl_a, prep_a, ws_a, f_v, dXv)
            df_a = raw_backward_dy(lib, d_v, pl_v, prep_v, ws_v, f_a, dXa)
        return None, None, None, None, df_a, df_v, g_a, g_v


def pair_apply(lib: Lib, side, call_a, call_v, f_a: torch.Tensor, f_v: torch.Tensor, flat_a: torch.Tensor, flat_v: torch.Tensor):
    """(f_a + audio_adapter(f_a, f_v), map_a, f_v + vis_adapter(f_v, f_a), map_v); token-major contiguous maps of the adapters'
    compute dtype, both adapters with flat parameters and without a temporal gate (AdapterStack checks)."""
    return _PairFlatFn.apply(lib, side, call_a, call_v, f_a, f_v, flat_a, flat_v)


def adapter_apply(lib: Lib, spec: AdapterSpec, training: bool, prep: torch.Tensor, X: torch.Tensor, Y: torch.Tensor,
                  params: List[Optional[torch.Tensor]], flat: Optional[torch.Tensor] = None,
                  residual: Optional[torch.Tensor] = None, skip: bool = False):
    """residual: out = residual + adapter(X, Y) (any tensor of X's shape; its gradient is dOut).
    skip: out = X + adapter(X, Y) with the matching `dX += dOut` fused into backward."""
    if skip and residual is not None:
        raise RuntimeError("dg-sct_amd: pass either residual= or skip=True, not both")
    if flat is not None:
        out, amap, tmap = _AdapterFlatFn.apply(lib, spec, training, prep, params, skip, residual, X, Y, flat)
    else:
        out, amap, tmap = _AdapterFn.apply(lib, spec, training, prep, skip, residual, X, Y, *params)
    return out, amap, (tmap if spec.temporal else None)


# ---- spatial-map pooling of the task heads ------------------------------------------------------
class _MapPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lib, f, amap):
        BT, N, C_ = f.shape
        pooled = torch.empty(BT, C_, dtype=torch.float32, device=f.device)
        with _dev_guard(f):
            lib.map_pool_forward(0 if f.dtype == torch.float32 else 1, BT, N, C_, f.data_ptr(), amap.data_ptr(),
                                 pooled.data_ptr(), _stream_of(f))
        ctx.lib = lib
        ctx.save_for_backward(f, amap)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        f, amap = ctx.saved_tensors
        BT, N, C_ = f.shape
        dpooled = dpooled.contiguous().float()
        _mark_stream_use(f, amap, dpooled)
        df = torch.empty_like(f) if ctx.needs_input_grad[1] else None
        dmap = torch.empty_like(amap) if ctx.needs_input_grad[2] else None
        with _dev_guard(f):
            ctx.lib.map_pool_backward(0 if f.dtype == torch.float32 else 1, BT, N, C_, f.data_ptr(), amap.data_ptr(),
                                      dpooled.data_ptr(), df.data_ptr() if df is not None else None,
                                      dmap.data_ptr() if dmap is not None else None, _stream_of(f))
        return None, df, dmap


def map_pool(f: torch.Tensor, spatial_att_maps: torch.Tensor, lib: Optional[Lib] = None) -> torch.Tensor:
    """Drop-in for `torch.bmm(spatial_att_maps, f)` at the end of the reference's layer loop
    (DG-SCT/AVE/nets/net_trans.py:922-924): f [BT,N,C] (fp32 | bf16), spatial_att_maps [BT,1,N] (the second output of the
    last p2 adapter) -> [BT,1,C] in f's dtype.  One HIP reduction forward, two kernels backward (dgsct_map_pool_*)."""
    if not f.is_cuda:
        raise RuntimeError("dg-sct_amd: map_pool needs CUDA/HIP tensors (no CPU fallback)")
    BT, N, C_ = f.shape
    if spatial_att_maps.shape not in ((BT, 1, N), (BT, N)):
        raise RuntimeError(f"dg-sct_amd: spatial_att_maps must be [BT,1,N] = [{BT},1,{N}], got {tuple(spatial_att_maps.shape)}")
    if f.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("dg-sct_amd: map_pool supports float32 and bfloat16 maps")
    from ._lib import default_lib
    pooled = _MapPoolFn.apply(lib or default_lib(), f.contiguous(), spatial_att_maps.reshape(BT, N).contiguous().float())
    return pooled.to(f.dtype).unsqueeze(1)


# ---- fused window attention of the frozen backbone blocks -------------------------------------------------------
class _WindowAttnFn(torch.autograd.Function):
    """softmax(scale q k^T + bm) v per (frame, window, head) on the qkv projection of the UN-partitioned map; window partition and cyclic
    shift are address arithmetic inside the kernel (csrc/wattn.hip).  bm / scale are frozen tables: no gradient."""

    @staticmethod
    def forward(ctx, lib, qkv, bm, scale, H, W, ws, shift, heads, cosine=False):
        B, L, C3 = qkv.shape
        hd = C3 // (3 * heads)
        nW = (H // ws) * (W // ws)
        geom = (B, H, W, ws, shift, heads, hd, int(bm.shape[0]))
        out = torch.empty(B, L, heads * hd, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(B, nW, heads, ws * ws, dtype=torch.float32, device=qkv.device)
        _mark_stream_use(qkv, bm, scale)
        with _dev_guard(qkv):
            lib.window_attn_forward(geom, qkv.data_ptr(), bm.data_ptr(), scale.data_ptr(), out.data_ptr(), lse.data_ptr(), _stream_of(qkv),
                                    1 if cosine else 0)
        ctx.lib, ctx.geom, ctx.flags = lib, geom, 1 if cosine else 0
        ctx.save_for_backward(qkv, bm, scale, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, bm, scale, out, lse = ctx.saved_tensors
        dout = dout.contiguous()
        if dout.dtype != qkv.dtype:
            dout = dout.to(qkv.dtype)
        dqkv = torch.empty_like(qkv)
        _mark_stream_use(qkv, bm, scale, out, lse, dout)
        with _dev_guard(qkv):
            ctx.lib.window_attn_backward(ctx.geom, qkv.data_ptr(), bm.data_ptr(), scale.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                         dout.data_ptr(), dqkv.data_ptr(), _stream_of(qkv), ctx.flags)
        return None, dqkv, None, None, None, None, None, None, None, None


def window_attention_supported(qkv: torch.Tensor, ws: int, heads: int) -> bool:
    """what csrc/wattn.hip takes: bf16 (the host emulation of the CPU tests: bf16 too), head width 8 / 16 / 24 / 32, windows of <= 144 tokens"""
    if qkv.dtype != torch.bfloat16 or qkv.dim() != 3 or qkv.shape[-1] % (3 * heads):
        return False
    hd = qkv.shape[-1] // (3 * heads)
    return hd in (8, 16, 24, 32) and ws * ws <= 144 and (ws * ws) % 4 == 0


def window_attention(qkv: torch.Tensor, bm: torch.Tensor, scale: torch.Tensor, H: int, W: int, ws: int, shift: int, heads: int,
                     lib: Optional[Lib] = None, cosine: bool = False) -> torch.Tensor:
    """qkv [B, H*W, 3*heads*hd] bf16 (token-major map, NOT partitioned / rolled), bm [1 | nW, heads, n, n] fp32, scale [heads] fp32
    -> O [B, H*W, heads*hd] at the map positions (what window_reverse + roll-back of the reference block produce).
    cosine: q and k are L2-normalised per head inside the kernel (Swin-V2), qkv is the raw projection."""
    if not window_attention_supported(qkv, ws, heads):
        raise RuntimeError("dg-sct_amd: window_attention takes bf16 qkv with head width 8/16/24/32 and windows of <= 144 tokens")
    if bm.dtype != torch.float32 or scale.dtype != torch.float32 or bm.shape[1:] != (heads, ws * ws, ws * ws):
        raise RuntimeError("dg-sct_amd: window_attention: bm must be fp32 [1 | windows, heads, n, n] and scale fp32 [heads]")
    from ._lib import default_lib
    return _WindowAttnFn.apply(lib or default_lib(), qkv.contiguous(), bm.contiguous(), scale.contiguous(), H, W, ws, shift, heads, cosine)


# ---- LayerNorm (+ residual) of the frozen backbone blocks ----------------------------------------------------
_LN_SCRATCH: Dict[Tuple, torch.Tensor] = {}


class _LayerNormFn(torch.autograd.Function):
    """out = LN(x; w, b) (+ residual) on the adapter tail's row kernels (dgsct_layer_norm_*): one pass forward (row statistics kept), one
    pass backward; w / b fp32.  Gradients for w / b only when they ask for one (the blocks are frozen: main_trans.py:211-256)."""

    @staticmethod
    def forward(ctx, lib, x, w, b, eps, residual):
        C_ = x.shape[-1]
        rows = x.numel() // C_
        out = torch.empty_like(x)
        mu = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        _mark_stream_use(x, w, b, residual)
        with _dev_guard(x):
            lib.layer_norm_forward(0 if x.dtype == torch.float32 else 1, rows, C_, x.data_ptr(), w.data_ptr(), b.data_ptr(), float(eps),
                                   residual.data_ptr() if residual is not None else None, out.data_ptr(), mu.data_ptr(), rstd.data_ptr(),
                                   _stream_of(x))
        ctx.lib, ctx.eps, ctx.has_res = lib, float(eps), residual is not None
        ctx.save_for_backward(x, w, b, mu, rstd)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, b, mu, rstd = ctx.saved_tensors
        C_ = x.shape[-1]
        rows = x.numel() // C_
        dout = dout.contiguous()
        if dout.dtype != x.dtype:
            dout = dout.to(x.dtype)
        dx = torch.empty_like(x)
        dwb = torch.zeros(2, C_, dtype=torch.float32, device=x.device)
        key = (id(ctx.lib), x.device, _stream_of(x), C_)
        scr = _LN_SCRATCH.get(key)
        if scr is None:
            scr = _LN_SCRATCH[key] = torch.empty(int(ctx.lib.c.dgsct_layer_norm_scratch_floats(C_)), dtype=torch.float32, device=x.device)
        _mark_stream_use(x, w, b, mu, rstd, dout)
        with _dev_guard(x):
            ctx.lib.layer_norm_backward(0 if x.dtype == torch.float32 else 1, rows, C_, dout.data_ptr(), x.data_ptr(), w.data_ptr(),
                                        b.data_ptr(), mu.data_ptr(), rstd.data_ptr(), ctx.eps, dx.data_ptr(), dwb[0].data_ptr(),
                                        dwb[1].data_ptr(), scr.data_ptr(), _stream_of(x))
        return (None, dx, dwb[0] if ctx.needs_input_grad[2] else None, dwb[1] if ctx.needs_input_grad[3] else None, None,
                dout if ctx.has_res else None)


def layer_norm_supported(x: torch.Tensor) -> bool:
    C_ = x.shape[-1]
    return x.dtype in (torch.float32, torch.bfloat16) and C_ <= 1536 and C_ % (8 if x.dtype == torch.bfloat16 else 4) == 0 and x.numel() > 0


def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5, residual: Optional[torch.Tensor] = None,
               lib: Optional[Lib] = None) -> torch.Tensor:
    """`F.layer_norm(x, (C,), weight, bias, eps)` (+ residual) over the last axis of x [..., C] (fp32 | bf16); weight / bias fp32 [C]."""
    if not layer_norm_supported(x):
        raise RuntimeError("dg-sct_amd: layer_norm takes fp32 / bf16 rows of <= 1536 channels (a multiple of 4; bf16: 8)")
    if weight.dtype != torch.float32 or bias.dtype != torch.float32:
        raise RuntimeError("dg-sct_amd: layer_norm: weight / bias must be fp32")
    from ._lib import default_lib
    if residual is not None:
        residual = residual.contiguous()
        if residual.shape != x.shape or residual.dtype != x.dtype:
            raise RuntimeError("dg-sct_amd: layer_norm: residual must match x")
    return _LayerNormFn.apply(lib or default_lib(), x.contiguous(), weight.contiguous(), bias.contiguous(), eps, residual)
