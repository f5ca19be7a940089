"""Training-step plumbing around the adapter stack: what the reference's AVE script does between model and optimizer
(reference ``DG-SCT/AVE/main_trans.py``), restated for one-process-per-GPU data parallelism.

* ``trainable_by_name`` -- the freeze rule (``main_trans.py:211-256``): ``'ViT'``/``'swin'`` (optionally their norms) and
  ``'htsat'`` parameters are frozen; ``'adapter_blocks'``, ``'CMBS'``, ``'mlp_class'`` and ``'temporal_attn'`` train;
  ``'mlp_class'`` gets its own learning rate (``:257-260``).
* ``make_optimizer`` -- ``optim.Adam(param_group)`` + ``StepLR(step_size=decay_epoch, gamma=decay)`` (``:276-278``) with the
  values of ``AVE/train.sh`` (lr 5e-4, lr_mlp 5e-6, decay 0.35 every 3 epochs).
* ``shard_clips`` / ``seed_everything`` -- per-rank sharding and seeding: the reference has one process (``--seed 43``,
  ``DataLoader(shuffle=True)``); with W ranks every rank draws the SAME epoch permutation (seed + epoch) and takes a strided
  slice of it, so the union over ranks is the single-process epoch order.
* ``StackTrainer`` -- one optimisation step of an ``AdapterStack``: forward, backward, gradient all-reduce
  (``GradAllReducer``), optimizer step.  ``bench.py`` times exactly this object; the gloo world-2 test drives the same code.

The ``accum_itr`` quirk (``main_trans.py:110,135-136``; ``train.sh`` uses ``--accum_itr=2``): the reference calls
``optimizer.zero_grad()`` EVERY iteration and ``optimizer.step()`` every ``accum_itr``-th, so the gradients of the other
iterations are discarded -- "accumulation" never accumulates; the effective schedule is "train on every 2nd batch with
batch size 8".  ``accum_mode="reference"`` reproduces that behaviour bit for bit (needed to re-trace the published
82.18 %); ``"accumulate"`` is the evident intent (gradients summed over ``accum_itr`` iterations, one step); default:
``"reference"``, because parity with the reference is the contract -- documented in DESIGN.md section 9.
"""
from __future__ import annotations

import random
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

TRAIN_KEYS = ("adapter_blocks", "CMBS", "mlp_class", "temporal_attn")      # main_trans.py:242-256


def trainable_by_name(named_parameters: Iterable[Tuple[str, torch.nn.Parameter]], is_vit_ln: bool = False, lr: float = 5e-4,
                      lr_mlp: float = 5e-6) -> List[Dict]:
    """Sets ``requires_grad`` by the reference's name rule and returns its ``param_group`` list (one entry per parameter,
    like the reference; frozen parameters are listed too -- Adam skips tensors without gradients)."""
    groups = []
    for name, p in named_parameters:
        p.requires_grad = False
        if "ViT" in name or "swin" in name:
            p.requires_grad = bool(is_vit_ln) and "norm" in name
        elif "htsat" in name:
            p.requires_grad = False
        elif any(k in name for k in TRAIN_KEYS):
            p.requires_grad = True
        groups.append({"params": p, "lr": lr_mlp if "mlp_class" in name else lr})
    return groups


def make_optimizer(model: torch.nn.Module, lr: float = 5e-4, lr_mlp: float = 5e-6, decay: float = 0.35, decay_epoch: int = 3,
                   is_vit_ln: bool = False, fused: Optional[bool] = None):
    groups = trainable_by_name(model.named_parameters(), is_vit_ln, lr, lr_mlp)
    kw = {}
    if fused is None:
        fused = all(g["params"].is_cuda for g in groups)
    if fused:
        kw["fused"] = True
    opt = torch.optim.Adam(groups, **kw)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=decay_epoch, gamma=decay)
    return opt, sched


def seed_everything(seed: int = 43, rank: int = 0):
    """Model initialisation must be identical on every rank (same seed); data-side randomness is decorrelated by rank."""
    torch.manual_seed(seed)
    random.seed(seed * 1000 + rank)
    return torch.Generator().manual_seed(seed * 1000 + rank)


def shard_clips(n: int, rank: int, world: int, seed: int = 43, epoch: int = 0, shuffle: bool = True, drop_last: bool = False) -> List[int]:
    """Clip indices of `rank` for one epoch: every rank computes the same permutation (seed + epoch) and takes every
    world-th element; the tail is padded by wrapping around (or dropped) so that all ranks run the same number of steps."""
    g = torch.Generator().manual_seed(seed + epoch)
    order = torch.randperm(n, generator=g).tolist() if shuffle else list(range(n))
    if drop_last:
        order = order[: n - n % world]
    elif len(order) % world:
        order += order[: world - len(order) % world]
    return order[rank::world]


class StackTrainer:
    """forward + backward + (DP) all-reduce + optimizer step of an AdapterStack on fixed cotangents (the adapter-path
    benchmark of BASELINE.json) or on a loss callable."""

    def __init__(self, stack, optimizer=None, reducer=None, accum_itr: int = 1, accum_mode: str = "reference"):
        if accum_mode not in ("reference", "accumulate"):
            raise ValueError("accum_mode must be 'reference' or 'accumulate'")
        self.stack, self.opt, self.reducer = stack, optimizer, reducer
        self.accum_itr, self.accum_mode = max(1, int(accum_itr)), accum_mode
        self.params = [p for p in stack.parameters() if p.requires_grad]
        self.it = 0

    def fwd_bwd(self, feats, cots=None, mcots=None, loss_fn=None):
        outs, maps = self.stack(feats, **getattr(self, "block_kwargs", {}))      # (vis_block= / aud_block=: frozen backbone blocks, backbone.py)
        if loss_fn is not None:
            loss_fn(outs, maps).backward()
        else:
            tensors = [t for pair in outs for t in pair]
            grads = [g for pair in cots for g in pair]
            if mcots is not None:
                tensors += [maps[0], maps[1]]
                grads += [mcots[0], mcots[1]]
            torch.autograd.backward(tensors, grads)
        for fv, fa in feats:
            if fv.requires_grad:
                fv.grad = None
            if fa.requires_grad:
                fa.grad = None
        return outs, maps

    def zero_grad(self, set_to_none: bool = True):
        if self.opt is not None:
            self.opt.zero_grad(set_to_none=set_to_none)
        else:
            for p in self.params:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    def step(self, feats, cots=None, mcots=None, loss_fn=None, last: bool = False) -> bool:
        """One training iteration with the reference's control flow (main_trans.py:110-136).  Returns True when the
        optimizer stepped."""
        first_of_window = self.it % self.accum_itr == 0
        do_step = (self.it + 1) % self.accum_itr == 0 or last
        if self.accum_mode == "reference" or first_of_window:
            self.zero_grad()                                     # reference: every iteration (discards the previous one)
        if self.reducer is not None:
            # Only the stepping iteration communicates: its hooks (and finish()) see the gradients of the whole window in
            # "accumulate" mode, and in "reference" mode the iterations whose gradients are thrown away cost no all-reduce.
            self.reducer.paused = not do_step
        self.fwd_bwd(feats, cots, mcots, loss_fn)
        self.it += 1
        if do_step:
            if self.reducer is not None:
                self.reducer.finish()                            # mean over ranks of what this iteration (window) produced
            if self.opt is not None:
                self.opt.step()
        elif self.reducer is not None:
            self.reducer.discard()                               # (nothing was launched while paused)
        if last:
            self.it = 0                                          # the reference keys the window on batch_idx, which restarts every epoch (main_trans.py:135)
        return do_step
