"""Data-parallel adapter-gradient exchange: one process per GPU, RCCL over xGMI.

The reference has no multi-process training at all (single GPU, or single-process nn.DataParallel:
avs_s4/train.py:139, main_avst.py:236; SURVEY.md header).  Clips are independent through the whole
adapter path, so DP shards clips across ranks and the only collective is the all-reduce of the
adapter gradients (188.5 M params = 754 MB fp32 for AVE/Swin-L).

Design for xGMI (point-to-point links, no switch): few large buckets (one per backbone stage, filled
in reverse order because backward reaches stage 3 first), each launched as soon as its last
gradient is produced, on a side stream so that the stage-0/1 backward (most of the FLOPs) hides
the stage-3/2 traffic.  Gradients are averaged (sum / world).  Parameters that never receive a
gradient (``gate_tk`` ...) are reduced as zeros so every rank issues identical collectives.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, buckets: Sequence[Sequence[torch.nn.Parameter]], process_group=None, overlap: bool = True,
                 comm_dtype: Optional[torch.dtype] = None):
        """buckets: parameter groups in the order backward finishes them (stage 3 first)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.overlap = overlap
        self.comm_dtype = comm_dtype
        self.buckets: List[List[torch.nn.Parameter]] = [[p for p in b if p.requires_grad] for b in buckets]
        self.buckets = [b for b in self.buckets if b]
        self.flat: List[torch.Tensor] = []
        self.views: List[List[torch.Tensor]] = []
        self._bucket_of: Dict[int, int] = {}
        for bi, b in enumerate(self.buckets):
            n = sum(p.numel() for p in b)
            dev = b[0].device
            flat = torch.zeros(n, dtype=comm_dtype or torch.float32, device=dev)
            self.flat.append(flat)
            vs, off = [], 0
            for p in b:
                vs.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
                self._bucket_of[id(p)] = bi
            self.views.append(vs)
        self._pending = [0] * len(self.buckets)
        self._ready: List[List[bool]] = [[False] * len(b) for b in self.buckets]
        self._work: List[Optional[object]] = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._stream = None
        self._hooks = []
        if self.world > 1 and overlap:
            for bi, b in enumerate(self.buckets):
                for pi, p in enumerate(b):
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi, pi)))

    # ------------------------------------------------------------------
    def _make_hook(self, bi: int, pi: int):
        def hook(param):
            self._ready[bi][pi] = True
            self._pending[bi] += 1
            if self._pending[bi] == len(self.buckets[bi]):
                self._launch(bi)
        return hook

    def _launch(self, bi: int):
        if self._launched[bi]:
            return
        self._launched[bi] = True
        flat = self.flat[bi]
        for p, v in zip(self.buckets[bi], self.views[bi]):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
        if flat.is_cuda and self.overlap:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            self._stream.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._stream):
                self._work[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self._work[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Call after loss.backward(): launches what the hooks did not, waits, writes averaged grads back."""
        if self.world <= 1:
            return
        for bi in range(len(self.buckets)):
            self._launch(bi)
        for bi, w in enumerate(self._work):
            if w is not None:
                w.wait()
        if self._stream is not None:
            torch.cuda.current_stream(self.flat[0].device).wait_stream(self._stream)
        inv = 1.0 / self.world
        for bi, b in enumerate(self.buckets):
            for p, v in zip(b, self.views[bi]):
                if p.grad is None:
                    continue        # stays None on every rank (same graph on every rank)
                p.grad.copy_(v).mul_(inv) if p.grad.dtype == v.dtype else p.grad.copy_(v.to(p.grad.dtype)).mul_(inv)
        self._pending = [0] * len(self.buckets)
        self._ready = [[False] * len(b) for b in self.buckets]
        self._work = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)

    @staticmethod
    def stage_buckets(stack) -> List[List[torch.nn.Parameter]]:
        """One bucket per backbone stage of an AdapterStack, last stage first."""
        out, idx = [], 0
        for s in stack.stages:
            ps = []
            for i in range(idx, idx + s["layers"]):
                for ml in (stack.audio_adapter_blocks_p1, stack.vis_adapter_blocks_p1, stack.audio_adapter_blocks_p2,
                           stack.vis_adapter_blocks_p2):
                    ps += list(ml[i].parameters())
            out.append(ps)
            idx += s["layers"]
        return out[::-1]
