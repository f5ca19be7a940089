"""Data-parallel adapter-gradient exchange: one process per GPU, RCCL over xGMI.

The reference has no multi-process training at all (single GPU, or single-process nn.DataParallel:
avs_s4/train.py:139, main_avst.py:236; SURVEY.md header).  Clips are independent through the whole
adapter path, so DP shards clips across ranks and the only collective is the all-reduce of the
adapter gradients (143 M params = 572 MB fp32 for AVE/Swin-B, 754 MB for Swin-L).

Zero-copy design: ``dgsct_adapter_backward`` writes ALL parameter gradients of one adapter call into one
flat fp32 buffer.  With flattened adapters (``VisualAdapter.flatten_parameters``) that buffer IS the gradient
of the adapter's single parameter, so the natural communication unit is that buffer: 48 all-reduces of
3-45 MB per step, issued in place (no flatten / unflatten kernels; the first version copied ~1900 tensors
in and out and cost 38 ms per step).  xGMI is point-to-point
(no switch): messages of tens of MB keep every link busy without the per-bucket latency dominating.
Buckets (one per backbone stage, backward order) only decide WHEN buffers are launched: with
``overlap=True`` a stage's buffers go to a side stream as soon as its last gradient has been produced,
so the stage-0/1 backward hides the stage-3/2 traffic.  Gradients are averaged (sum / world).
Un-flattened modules (the ~1900 individual tensors of the reference layout) fall back to one
flattened (torch.cat) all-reduce per bucket plus a copy back.  Parameters without a gradient do
not communicate (every rank runs the same graph, so every rank skips the same ones).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import os

import torch
import torch.distributed as dist


def init_process_group(device: torch.device, backend: str = "nccl", **kw):
    """``dist.init_process_group`` for one-process-per-GPU data parallelism over RCCL ("nccl" IS RCCL on ROCm), bound to
    ``device`` (eager communicator creation, no device guessing in barriers).

    The collective kernels run on ProcessGroupNCCL's internal stream, taken from torch's NORMAL-priority pool by default.
    The library's own side / aux streams live in the HIGH-priority pool (``dgsct_stream_create``), whose hardware queues
    are separate, so the collective stream cannot land on one of their queues; putting it in the high pool as well
    (``DGSCT_NCCL_HIGH_PRIORITY=1``) was measured to collide with the second adapter stream (97 -> 169 ms per step on one
    MI355X).  See DESIGN.md section 5 and tools/dp_host_probe.py."""
    if backend == "nccl":
        hi = os.environ.get("DGSCT_NCCL_HIGH_PRIORITY", "0") == "1"
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=hi)
        return dist.init_process_group(backend, device_id=device, pg_options=opts, **kw)
    return dist.init_process_group(backend, **kw)


class GradAllReducer:
    def __init__(self, buckets: Sequence[Sequence[torch.nn.Parameter]], process_group=None, overlap: bool = True,
                 comm_dtype: Optional[torch.dtype] = None, force: bool = False):
        """buckets: parameter groups in the order backward finishes them (stage 3 first)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force = force and dist.is_initialized()        # run the collectives even with one rank (self-test)
        self.active = self.world > 1 or self.force
        self.overlap = overlap
        self.comm_dtype = comm_dtype
        self.buckets: List[List[torch.nn.Parameter]] = [[p for p in b if p.requires_grad] for b in buckets]
        self.buckets = [b for b in self.buckets if b]
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._sent: List[set] = [set() for _ in self.buckets]      # id(param) of the gradients a launched bucket has sent
        self._late: List[list] = [[] for _ in self.buckets]        # parameters whose FIRST gradient came after the launch
        self._dirty = [False] * len(self.buckets)       # a gradient that was already sent was accumulated into again
        self.relaunches = 0                             # buckets that needed a second collective in finish() (late gradients)
        # Gradients a bucket must have seen before it may launch.  Not len(bucket): some parameters never receive one
        # (`gate_tk` in the ave/avvp/pretrain flavours, `conv_adapter.*` with the bicubic remap, `fc_caption.*`, ... --
        # SURVEY.md 8c), so their post-accumulate hooks never fire.  The first step counts who does (every rank runs the
        # same graph, so every rank learns the same numbers); from the second step on the hooks launch each bucket as
        # soon as its last gradient exists.
        self._expected: List[Optional[int]] = [None] * len(self.buckets)
        self._producers: List[set] = [set() for _ in self.buckets]
        self.hook_launches = 0                          # buckets launched from hooks (before finish()) in the last step
        self.paused = False                             # True: hooks do nothing (rank-local passes, e.g. bench.py's profiling leg)
        self.skip_exchange = False                      # True: finish() sends nothing either (bench.py's "step without the exchange" A/B leg)
        self._work: List[object] = []
        self._reduced: List[torch.Tensor] = []          # tensors to scale by 1/world after the wait
        self._copy_back: List[tuple] = []               # (flat, params) of the fallback path
        self._stream = None
        self._avg_in_collective = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self._hooks = []
        if self.active and overlap:
            for bi, b in enumerate(self.buckets):
                for p in b:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    # ------------------------------------------------------------------
    def _make_hook(self, bi: int):
        def hook(param):
            if self.paused:
                return
            if param.grad is not None and param.grad.is_cuda:
                # the gradient was produced on whatever stream this adapter's backward ran on (main or the audio-side
                # stream of AdapterStack).  Only REMEMBER the stream here: the communication stream waits for the
                # producer streams once per BUCKET, at launch (one event pair per stream and bucket -- 8 per step -- instead
                # of one per gradient: every cross-stream event costs the recording stream a barrier packet)
                self._producers[bi].add(torch.cuda.current_stream(param.grad.device))
            self._pending[bi] += 1
            if self._launched[bi]:
                # more gradient events than last step (the count the launch was keyed on): a parameter that only sometimes
                # gets a gradient is simply reduced later by finish(); one whose gradient is ALREADY in flight has just been
                # accumulated into under the collective -- that cannot be repaired, finish() raises
                if id(param) in self._sent[bi]:
                    self._dirty[bi] = True
                elif any(q is param for q in self._late[bi]):
                    # a late parameter accumulated into a SECOND time (three or more backward() calls in the step): listed once --
                    # twice in the message it would be averaged twice -- and the bucket is dirty like any other accumulation that
                    # happens while the step's collectives are under way
                    self._dirty[bi] = True
                else:
                    self._late[bi].append(param)
                return
            if self._expected[bi] is not None and self._pending[bi] == self._expected[bi]:
                self.hook_launches += 1
                self._launch(bi)
        hook._dgsct_drains_aux = True           # ops._may_adopt: this hook never reads .grad before _launch() has drained the aux streams
        return hook

    def _comm_stream(self, device):
        if self._stream is None:
            # Only event waits/records live on this stream (the collective kernels run on ProcessGroupNCCL's own stream).
            # Low priority class = a hardware queue shared with no compute stream (caller: normal, adapter side streams:
            # high): a barrier packet waiting for the slower adapter stream must not sit in the other one's queue.
            from . import _lib, ops
            self._stream = ops.priority_stream(_lib.default_lib(), device, +1)
        return self._stream

    def _all_reduce(self, tensors: List[torch.Tensor], producers=()):
        """one grouped, in-place all-reduce of `tensors` (ncclGroupStart/End: ONE work object and ONE pair of stream
        events for the whole group -- per-tensor calls cost 48 event pairs per step, which stalls the HIP launch path)"""
        # RCCL averages in the collective itself (ncclAvg); gloo (CPU tests) sums and finish() scales
        op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM

        def issue():
            if len(tensors) == 1 or not self._avg_in_collective:
                for t in tensors:
                    self._work.append(dist.all_reduce(t, op=op, group=self.group, async_op=True))
            else:
                with dist._coalescing_manager(self.group, async_ops=True) as cm:
                    for t in tensors:
                        dist.all_reduce(t, op=op, group=self.group)
                self._work.append(cm)

        if tensors[0].is_cuda:
            # the flat gradient buffers are completed on the adapters' weight-gradient (aux) streams, whose join is deferred
            # (ops.DEFER_AUX_JOIN): order this stream -- and the producing streams -- behind them before anything reads a gradient
            from . import ops as _ops
            _ops.drain_aux(tensors[0].device)
        if tensors[0].is_cuda and self.overlap and os.environ.get("DGSCT_DP_COMM_STREAM", "0") != "1":
            # Launch from the stream that produced the bucket's last gradient: ProcessGroupNCCL orders its own collective
            # stream behind an event on the CURRENT stream, so no extra stream is needed -- the current stream only has to
            # be behind the bucket's other producer (the second adapter stream of AdapterStack).  A dedicated communication
            # stream costs one more hardware queue: its pending event waits sat in the low-priority queue pool next to the
            # weight-gradient (aux) streams and stalled them (1 GPU, --force-dp: 100 vs 67 ms per step).
            cur = torch.cuda.current_stream(tensors[0].device)
            for ps in producers:
                if ps != cur:
                    cur.wait_stream(ps)
            issue()
        elif tensors[0].is_cuda and self.overlap:
            st = self._comm_stream(tensors[0].device)
            cur = torch.cuda.current_stream(tensors[0].device)
            st.wait_stream(cur)
            for ps in producers:
                if ps != cur:
                    st.wait_stream(ps)
            with torch.cuda.stream(st):
                issue()
        else:
            issue()
        if not self._avg_in_collective:
            self._reduced.extend(tensors)

    def _launch(self, bi: int):
        if self._launched[bi]:
            return
        self._launched[bi] = True
        grads = [p for p in self.buckets[bi] if p.grad is not None]
        self._sent[bi] = {id(p) for p in grads}
        self._reduce(bi, grads)

    def _reduce(self, bi: int, grads: List[torch.nn.Parameter]):
        big = [p for p in grads if p.grad.is_contiguous() and p.grad.dtype == torch.float32 and p.grad.numel() >= 4096
               and self.comm_dtype in (None, torch.float32)]
        if len(big) <= 64:
            # flattened adapters (VisualAdapter.flatten_parameters): the gradient of an adapter IS the library's flat
            # buffer -> reduce it in place, no staging copies
            if big:
                self._all_reduce([p.grad for p in big], self._producers[bi])
            loose = [p for p in grads if all(p is not q for q in big)]
        else:
            loose = grads
        if loose:                                       # everything else: one flattened message per bucket
            flat = torch.cat([p.grad.reshape(-1).to(self.comm_dtype or torch.float32) for p in loose])
            self._all_reduce([flat], self._producers[bi])
            self._copy_back.append((flat, loose))

    def _complete(self):
        """wait for everything enqueued so far, then average / copy back what it reduced"""
        for w in self._work:
            w.wait()
        if self._stream is not None:
            torch.cuda.current_stream(self._stream.device).wait_stream(self._stream)
        if self._reduced:
            torch._foreach_mul_(self._reduced, 1.0 / self.world)
        for flat, params in self._copy_back:
            off = 0
            for p in params:
                n = p.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        self._work, self._reduced, self._copy_back = [], [], []

    def finish(self):
        """Call after loss.backward(): launches what the hooks did not, waits, averages."""
        if not self.active:
            return
        if self.skip_exchange:
            self._reset()
            return
        for bi in range(len(self.buckets)):
            if self._dirty[bi]:
                self._complete()
                self._reset()
                raise RuntimeError(
                    "dg-sct_amd GradAllReducer: a gradient of bucket %d was accumulated into after the bucket's overlapped "
                    "all-reduce had been launched (a second backward() in the step, retain_graph, a shared parameter): the "
                    "collective ran on a half-accumulated buffer.  Set reducer.paused = True around every backward but the "
                    "last of a step (StackTrainer does), or build the reducer with overlap=False." % bi)
            n_grad = sum(1 for p in self.buckets[bi] if p.grad is not None)
            if self._launched[bi] and self._late[bi]:
                # first gradients that arrived after the hook-launch (a parameter that is only sometimes used): they were
                # not part of the message; every rank runs the same graph and sees the same late set
                late = [p for p in self._late[bi] if p.grad is not None]
                self._sent[bi] |= {id(p) for p in late}
                self._reduce(bi, late)
                self.relaunches += 1
            if self._hooks and self._expected[bi] != n_grad:
                self._expected[bi] = n_grad             # first step (or the set of used parameters changed)
            self._launch(bi)
        self._complete()
        self._reset()

    def _reset(self):
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._sent = [set() for _ in self.buckets]
        self._late = [[] for _ in self.buckets]
        self._dirty = [False] * len(self.buckets)
        self._producers = [set() for _ in self.buckets]
        self._work, self._reduced, self._copy_back = [], [], []
        self.last_hook_launches, self.hook_launches = self.hook_launches, 0

    def discard(self):
        """An iteration whose gradients are thrown away or kept accumulating (the `accum_itr` control flow, train.py).
        StackTrainer pauses the hooks for such iterations, so normally nothing was launched; if a caller left them on,
        complete what they launched -- every rank launched the same collectives -- INCLUDING the 1/world scale of the
        SUM back-ends, so an accumulating caller never keeps a half-finished sum."""
        if not self.active:
            return
        self._complete()
        self._reset()

    @staticmethod
    def stage_buckets(stack, split_positions: Optional[str] = None) -> List[List[torch.nn.Parameter]]:
        """One bucket per backbone stage of an AdapterStack, last stage first (backward order).

        ``split_positions`` (default: environment ``DGSCT_DP_BUCKETS``): ``"position"`` cuts every stage whose adapters hold more than an
        eighth of the gradient bytes into one bucket per POSITION (p2 adapters of a layer, then its p1 adapters: the order their
        gradients appear in backward).  Stage 0 holds 53 % of the payload (the token-remap weights) and is the LAST stage of backward:
        as one bucket its whole all-reduce is exposed behind the step; per position, three quarters of it run under the remaining
        stage-0 backward.  On ONE rank every extra hook-launched collective costs ~0.5 ms of event traffic, more
        than it can hide, so the default stays per stage; the switch exists so that an 8-GPU run can A/B it:
            DGSCT_DP_BUCKETS=position python -m torch.distributed.run ... bench.py --gpus 8"""
        mode = split_positions if split_positions is not None else os.environ.get("DGSCT_DP_BUCKETS", "stage")
        per_stage, idx = [], 0
        for s in stack.stages:
            pos = []                                                    # buckets of this stage in FORWARD order: (layer, p1), (layer, p2), ...
            for i in range(idx, idx + s["layers"]):
                pos.append(list(stack.audio_adapter_blocks_p1[i].parameters()) + list(stack.vis_adapter_blocks_p1[i].parameters()))
                pos.append(list(stack.audio_adapter_blocks_p2[i].parameters()) + list(stack.vis_adapter_blocks_p2[i].parameters()))
            per_stage.append(pos)
            idx += s["layers"]
        total = sum(p.numel() for st in per_stage for b in st for p in b) or 1
        out = []
        for pos in per_stage[::-1]:
            n = sum(p.numel() for b in pos for p in b)
            if mode == "position" and n * 8 > total:
                out += pos[::-1]
            else:
                out.append([p for b in pos for p in b])
        return out
