"""CPU-only: host logic around the C ABI -- exported symbols, the nn.Module drop-in surface (names, shapes, errors),
the AVE layer-loop schedule against the reference-generated stack fixture (through the host-emulated primitives), and
the data-parallel gradient all-reduce with world_size 2 over gloo."""
import ctypes
import os
import re
import sys
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import GOLDEN, ROOT, load_golden, rel_err

sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
from build_emu import build_emu  # noqa: E402

import dgsct_amd  # noqa: E402
from dgsct_amd import AdapterStack, GradAllReducer, VisualAdapter  # noqa: E402
from dgsct_amd._lib import LIB_PATH, Lib  # noqa: E402
from dgsct_amd.stack import default_opt  # noqa: E402


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "dgsct.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(dgsct_[a-z_]+)\s*\(", hdr)))


def test_header_symbols_are_exported():
    syms = _declared_symbols()
    assert {"dgsct_query", "dgsct_prepare", "dgsct_adapter_forward", "dgsct_adapter_backward"} <= set(syms)
    if not os.path.exists(LIB_PATH):
        pytest.skip("libdgsct.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(LIB_PATH)              # loads without a GPU; no compute call is made
    for s in syms:
        assert hasattr(lib, s), f"libdgsct.so does not export {s}"
    lib.dgsct_arch.restype = ctypes.c_char_p
    assert lib.dgsct_arch() == b"gfx950"


def test_gemm8_isa_keeps_the_hand_ordered_reads_intact():
    """gemm8.hip orders inline-asm LDS reads by hand; hipcc does not know their results arrive late.  The checker compiles
    the file to gfx950 ISA (no GPU needed) and verifies nothing touches an in-flight register, no spill and no
    compiler-made vmcnt wait sits in the k-loop (a register-allocation change once broke exactly this: DESIGN.md 3.1b)."""
    import shutil
    import subprocess
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_gemm8_isa.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("ok   ") >= 20


def test_query_rejects_bad_descriptor():
    emu = Lib(build_emu())
    from dgsct_amd.ops import AdapterSpec
    bad = AdapterSpec(N=16, C=30, No=36, Co=16, tk=4)          # C not a multiple of 4 / r
    with pytest.raises(RuntimeError, match="bad descriptor"):
        emu.query(bad.desc(10, torch.float32, True))


FLAVOUR_OF = {"ave_orderA": "ave", "avvp": "avvp", "avs_s4": "avs_s4", "avs_ms3": "avs_ms3", "avqa": "avqa", "pretrain": "pretrain"}


@pytest.mark.parametrize("name", sorted(FLAVOUR_OF))
def test_module_state_dict_matches_reference(name):
    """parameter / buffer names and shapes are the checkpoint format (reference main_trans.py:306 loads by name)"""
    fx = load_golden(name)
    c = fx["cfg"]
    fl = FLAVOUR_OF[name]
    opt = SimpleNamespace(is_multimodal=1, num_conv_group=c["g"], is_before_layernorm=int(c["ln_before"] or fl.startswith("avs")),
                          is_post_layernorm=int(c["ln_post"]), num_tokens=c["tk"])
    kw = dict(num_tk=c["tk"]) if fl in ("ave", "avvp", "pretrain") else {}
    m = VisualAdapter(c["C"], c["C"], "bottleneck", reduction_factor=c["r"], opt=opt, use_bn=c["use_bn"], use_gate=c["use_gate"],
                      conv_dim_in=c["No"], conv_dim_out=c["N"], linear_in=c["Co"], linear_out=c["C"], flavour=fl, **kw)
    ours = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    ref = {k: tuple(v.shape) for k, v in fx["state0"].items()}
    assert ours == ref
    m.load_state_dict(fx["state0"])            # strict


def test_unknown_adapter_kind_raises_like_the_reference():
    opt = SimpleNamespace(is_multimodal=1, num_conv_group=2, is_before_layernorm=1, is_post_layernorm=1, num_tokens=4)
    with pytest.raises(NotImplementedError):
        VisualAdapter(32, 32, "lora", opt=opt, reduction_factor=8, conv_dim_in=36, conv_dim_out=16, linear_in=16, linear_out=32)


def test_ignored_ln_before_warns_once():
    """AVS flavours build ln_before and never apply it (PVT_AVSModel.py:239): mirrored, with a warning."""
    import warnings
    from dgsct_amd import adapter as A
    A._WARNED.clear()
    opt = SimpleNamespace(is_multimodal=1, num_conv_group=2, is_before_layernorm=1, is_post_layernorm=1, num_tokens=4)
    kw = dict(reduction_factor=8, opt=opt, conv_dim_in=36, conv_dim_out=16, linear_in=16, linear_out=32, flavour="avs_ms3")
    with pytest.warns(UserWarning, match="never applies"):
        m = VisualAdapter(32, 32, "bottleneck", **kw)
    assert hasattr(m, "ln_before") and not m.spec.ln_before
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        VisualAdapter(32, 32, "bottleneck", **kw)                # second construction: silent


def test_no_cpu_fallback():
    opt = SimpleNamespace(is_multimodal=1, num_conv_group=2, is_before_layernorm=1, is_post_layernorm=1, num_tokens=4)
    m = VisualAdapter(32, 32, "bottleneck", reduction_factor=8, opt=opt, num_tk=4, conv_dim_in=36, conv_dim_out=16,
                      linear_in=16, linear_out=32)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.randn(10, 32, 16, 1), torch.randn(10, 16, 36, 1))


def _stack_from_fixture(fx, emu, **kw):
    opt = default_opt(num_tokens=4)
    st = AdapterStack(fx["stages"], opt=opt, lib=emu, concurrent=False, **kw)
    st.load_state_dict(fx["state0"])
    return st


def test_stack_schedule_matches_reference_loop():
    """a-10: 12 adapters in the reference's ModuleLists + the AVE layer loop, identity backbone, vs. the reference."""
    emu = Lib(build_emu())
    fx = load_golden("stack_2stage")
    st = _stack_from_fixture(fx, emu).train()
    feats = [(a.clone().requires_grad_(True), b.clone().requires_grad_(True)) for a, b in fx["feats"]]
    outs, maps = st(feats)
    for (fv, fa), (rv, ra) in zip(outs, fx["outs"]):
        assert rel_err(fv, rv) < 1e-4 and rel_err(fa, ra) < 1e-4
    assert rel_err(maps[0], fx["maps"][0]) < 1e-4 and rel_err(maps[1], fx["maps"][1]) < 1e-4
    tensors = [t for pr in outs for t in pr] + [maps[0], maps[1]]
    grads = [g for pr in fx["cots"] for g in pr] + [fx["mcots"][0], fx["mcots"][1]]
    torch.autograd.backward(tensors, grads)
    for (fv, fa), (gv, ga) in zip(feats, fx["dfeats"]):
        assert rel_err(fv.grad, gv) < 1e-4 and rel_err(fa.grad, ga) < 1e-4
    got = {k: p.grad for k, p in st.named_parameters() if p.grad is not None}
    assert set(got) == set(fx["grads"])
    for k, g in fx["grads"].items():
        assert rel_err(got[k], g) < 1e-4, k
    assert all(int(m.bn1.num_batches_tracked) == 1 for m in st.vis_adapter_blocks_p1)


@pytest.mark.parametrize("mode", ["unfused", "frozen_blocks"])
def test_stack_residual_variants_agree(mode):
    """The fused residual (default, pinned against the reference fixture above) must agree with (a) the un-fused
    `f + adapter(...)` schedule and (b) stay consistent when frozen blocks sit between adapter and add (the
    `residual=` entry point: the block output, not x, is the addend)."""
    emu = Lib(build_emu())
    fx = load_golden("stack_2stage")

    def run(**kw):
        blocks = kw.pop("blocks", False)
        st = _stack_from_fixture(fx, emu, **kw).train()
        feats = [(a.clone().requires_grad_(True), b.clone().requires_grad_(True)) for a, b in fx["feats"]]
        vb = (lambda idx, half, f: 0.25 * torch.tanh(f)) if blocks else None
        ab = (lambda idx, f: f + 0.125 * torch.sin(f)) if blocks else None
        outs, maps = st(feats, vis_block=vb, aud_block=ab)
        torch.autograd.backward([t for pr in outs for t in pr] + [maps[0], maps[1]],
                                [g for pr in fx["cots"] for g in pr] + list(fx["mcots"]))
        return outs, feats, {k: p.grad for k, p in st.named_parameters() if p.grad is not None}

    blocks = mode == "frozen_blocks"
    o1, f1, g1 = run(fuse_residual=True, blocks=blocks)
    o2, f2, g2 = run(fuse_residual=False, blocks=blocks)
    for (a, b), (c, d) in zip(o1, o2):
        assert rel_err(a, c) < 1e-5 and rel_err(b, d) < 1e-5
    for (a, b), (c, d) in zip(f1, f2):
        assert rel_err(a.grad, c.grad) < 1e-4 and rel_err(b.grad, d.grad) < 1e-4
    assert set(g1) == set(g2)
    for k in g1:
        assert rel_err(g1[k], g2[k]) < 1e-4, k


@pytest.mark.parametrize("half", [False, True], ids=["full", "half_batch"])
def test_pair_node_matches_the_reference_and_the_two_node_path(half):
    """ops._PairFlatFn (both adapters of a position as one autograd node, the dY products issued last with the other call's dX as
    their residual: dgsct_adapter_backward_ex2 HOLD_DY / ONLY_DY) on the host-emulated library: against the reference stack fixture
    (outputs, maps, input gradients, every parameter gradient) and against the two-node path on a half batch whose inputs do not
    require gradients (one stream = one workspace: the second call of a pair once overwrote what the first one's dY product reads)."""
    from dgsct_amd import ops
    emu = Lib(build_emu())
    fx = load_golden("stack_2stage")
    BT = fx["feats"][0][0].shape[0]
    lo, hi = (0, BT // 2) if half else (0, BT)
    res = {}
    for pair in (False, True):
        st = AdapterStack(fx["stages"], opt=default_opt(num_tokens=4), lib=emu, concurrent=False, pair_backward=pair)
        st.load_state_dict(fx["state0"])
        st.flatten_parameters().train()
        feats = [(a[lo:hi].clone().requires_grad_(not half), b[lo:hi].clone().requires_grad_(not half)) for a, b in fx["feats"]]
        calls = []
        orig = ops.pair_apply
        ops.pair_apply = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            outs, maps = st(feats)
        finally:
            ops.pair_apply = orig
        assert len(calls) == (2 * sum(s["layers"] for s in fx["stages"]) if pair else 0)
        torch.autograd.backward([t for pr in outs for t in pr] + [maps[0], maps[1]],
                                [g[lo:hi] for pr in fx["cots"] for g in pr] + [fx["mcots"][0][lo:hi], fx["mcots"][1][lo:hi]])
        r = {"outs": outs, "maps": maps, "dfeats": [(a.grad, b.grad) for a, b in feats]}
        r["grads"] = {}
        for name, m in st.named_modules():
            if hasattr(m, "flat_param"):
                for pn, (off, cnt, shape) in m._flat_layout.items():
                    r["grads"][name + "." + pn] = m.flat_param.grad[off:off + cnt].view(shape)
        res[pair] = r
    a, b = res[False], res[True]
    for (x1, y1), (x2, y2) in zip(a["outs"], b["outs"]):
        assert torch.equal(x1, x2) and torch.equal(y1, y2)
    assert all(torch.equal(m1, m2) for m1, m2 in zip(a["maps"], b["maps"]))
    for k, g in a["grads"].items():
        assert rel_err(b["grads"][k], g) < 1e-5, k
    if not half:
        for (x1, y1), (x2, y2) in zip(a["dfeats"], b["dfeats"]):
            assert rel_err(x2, x1) < 1e-5 and rel_err(y2, y1) < 1e-5
        for (fv, fa), (gv, ga) in zip(b["dfeats"], fx["dfeats"]):
            assert rel_err(fv, gv) < 1e-4 and rel_err(fa, ga) < 1e-4
        for k, g in fx["grads"].items():
            assert rel_err(b["grads"][k], g) < 1e-4, k


def test_flattened_parameters_keep_the_checkpoint_format_and_the_gradients():
    """flatten_parameters(): one flat fp32 Parameter per adapter (gradient = the library's flat buffer, adopted by
    autograd without a copy); state_dict keys/values stay the reference's; gradients equal the reference's."""
    emu = Lib(build_emu())
    fx = load_golden("stack_2stage")
    st = _stack_from_fixture(fx, emu).flatten_parameters().train()
    sd = st.state_dict()
    assert set(sd) == set(fx["state0"])
    assert all(torch.equal(sd[k], fx["state0"][k]) for k in sd)
    st2 = AdapterStack(fx["stages"], opt=default_opt(num_tokens=4), lib=emu, concurrent=False).flatten_parameters()
    st2.load_state_dict(fx["state0"])                                   # strict, reference key names
    assert all("adapter_blocks" in n for n, _ in st2.named_parameters())   # reference freeze rule (main_trans.py:242)
    outs, maps = st2([(a.clone(), b.clone()) for a, b in fx["feats"]])
    torch.autograd.backward([t for pr in outs for t in pr] + [maps[0], maps[1]],
                            [g for pr in fx["cots"] for g in pr] + list(fx["mcots"]))
    n_checked = 0
    for n, m in st2.named_modules():
        if hasattr(m, "flat_param"):
            g = m.flat_param.grad
            assert g is not None and g._base is None                    # owns its storage: no AccumulateGrad clone
            for name, (off, cnt, shape) in m._flat_layout.items():
                ref = fx["grads"].get(n + "." + name)
                if ref is None:
                    assert float(g[off:off + cnt].abs().max()) == 0.0
                else:
                    assert rel_err(g[off:off + cnt].view(shape), ref) < 1e-4, n + "." + name
                    n_checked += 1
    assert n_checked == len(fx["grads"])


def fake_replicate(network):
    """What torch.nn.parallel.replicate does for ONE replica, minus the device broadcast (which needs GPUs): every module
    becomes ``_replicate_for_data_parallel()`` (shallow ``__dict__`` copy, EMPTY ``_parameters``), children are re-wired, and
    the parameters come back as plain NON-LEAF tensor attributes (``Broadcast.apply`` outputs; here ``p.clone()``, which
    keeps the autograd edge to the original parameter exactly like the broadcast does)."""
    modules = list(network.modules())
    index = {m: i for i, m in enumerate(modules)}
    copies = [m._replicate_for_data_parallel() for m in modules]
    memo = {}
    for i, m in enumerate(modules):
        r = copies[i]
        for key, child in m._modules.items():
            r._modules[key] = None if child is None else copies[index[child]]
        for key, prm in m._parameters.items():
            if prm is None:
                r._parameters[key] = None
            else:
                if prm not in memo:
                    memo[prm] = prm.clone()
                setattr(r, key, memo[prm])
        for key, buf in m._buffers.items():
            r._buffers[key] = buf
    return copies[0]


@pytest.mark.parametrize("flat", [False, True])
def test_data_parallel_replica_runs_and_routes_gradients(flat):
    """nn.DataParallel (reference AVS/AVQA: avs_s4/train.py:139, main_avst.py:236; SURVEY 8b "must be preserved:
    DataParallel.replicate") re-creates the module every forward as a replica with no Parameters.  The replica must
    compute what the module computes and send its gradients back to the module's parameters; with the parameter table
    cached on the ORIGINAL before replication, too."""
    emu = Lib(build_emu())
    fx = load_golden("ave_orderA")
    c = fx["cfg"]
    opt = SimpleNamespace(is_multimodal=1, num_conv_group=c["g"], is_before_layernorm=int(c["ln_before"]),
                          is_post_layernorm=int(c["ln_post"]), num_tokens=c["tk"])

    def make():
        m = VisualAdapter(c["C"], c["C"], "bottleneck", reduction_factor=c["r"], opt=opt, use_bn=c["use_bn"], use_gate=c["use_gate"],
                          num_tk=c["tk"], conv_dim_in=c["No"], conv_dim_out=c["N"], linear_in=c["Co"], linear_out=c["C"], lib=emu)
        m.load_state_dict(fx["state0"])
        return (m.flatten_parameters() if flat else m).train()

    X, Y = fx["X"], fx["Y"]
    x = lambda: X.permute(0, 2, 1).unsqueeze(-1).clone().requires_grad_(True)
    y = lambda: Y.permute(0, 2, 1).unsqueeze(-1).clone().requires_grad_(True)
    cot = torch.randn(X.shape[0], c["C"], c["N"], 1, generator=torch.Generator().manual_seed(1))
    mcot = torch.randn(X.shape[0], 1, c["N"], generator=torch.Generator().manual_seed(2))

    ref = make()
    xr, yr = x(), y()
    o, mp_ = ref(xr, yr)
    torch.autograd.backward([o, mp_], [cot, mcot])
    ref_grads = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}

    m = make()
    m(x(), y())                                   # caches the parameter table / prepared weights on the original first
    m.zero_grad(set_to_none=True)
    m.load_state_dict(fx["state0"])               # (undo the BN running-stat update of that call)
    rep = fake_replicate(m)
    assert rep is not m and len(list(rep.parameters())) == 0
    x1, y1 = x(), y()
    o1, mp1 = rep(x1, y1)
    assert rel_err(o1, o) < 1e-6 and rel_err(mp1, mp_) < 1e-6
    torch.autograd.backward([o1, mp1], [cot, mcot])
    assert rel_err(x1.grad, xr.grad) < 1e-6 and rel_err(y1.grad, yr.grad) < 1e-6
    got = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    assert set(got) == set(ref_grads) and got
    for n in ref_grads:
        assert rel_err(got[n], ref_grads[n]) < 1e-6, n
    # the replica owns its caches: nothing it built leaked into the original (shallow __dict__ copy)
    assert rep.__dict__.get("_ptab") is None
    assert m.__dict__["_ptab"] is not rep.__dict__.get("_ptab")
    assert rep._prep_cache is not m._prep_cache
    # and the original still works after its replica ran
    o2, _ = m(x(), y())
    assert torch.isfinite(o2).all()


# ---------------------------------------------------------------------------------------------------------------
def _dp_worker(rank, world, port, emu_path, q, flat, buckets="stage"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    emu = Lib(emu_path)
    fx = load_golden("stack_2stage")
    opt = default_opt(num_tokens=4, is_bn=0)                       # per-replica BN statistics differ by design
    st = AdapterStack(fx["stages"], opt=opt, lib=emu, concurrent=False)
    st.load_state_dict(fx["state0"], strict=False)
    if flat:
        st.flatten_parameters()
    red = GradAllReducer(GradAllReducer.stage_buckets(st, buckets))
    if buckets == "position":                                      # (layer, p1 / p2) buckets of the heavy stages: 2 per layer (DGSCT_DP_BUCKETS=position)
        assert len(red.buckets) == 2 * sum(s["layers"] for s in fx["stages"]), len(red.buckets)
    BT = fx["feats"][0][0].shape[0]
    lo, hi = rank * BT // world, (rank + 1) * BT // world
    feats = [(a[lo:hi].clone(), b[lo:hi].clone()) for a, b in fx["feats"]]
    for it in range(2):
        # two steps: the first learns how many gradients each bucket really receives (gate_tk & co never get one), from
        # the second on the hooks launch every bucket DURING backward (the overlap SURVEY.md 8e asks for)
        for p in st.parameters():
            p.grad = None
        outs, maps = st(feats)
        tensors = [t for pr in outs for t in pr]
        grads = [g[lo:hi] for pr in fx["cots"] for g in pr]
        torch.autograd.backward(tensors, grads)
        red.finish()
        assert red.last_hook_launches == (0 if it == 0 else len(red.buckets)), (it, red.last_hook_launches)
    if rank == 0:
        if flat:
            out = {}
            for n, m in st.named_modules():
                if hasattr(m, "flat_param"):
                    for name, (off, cnt, shape) in m._flat_layout.items():
                        out[n + "." + name] = m.flat_param.grad[off:off + cnt].view(shape).numpy().copy()
            q.put(out)
        else:
            q.put({k: p.grad.numpy().copy() for k, p in st.named_parameters() if p.grad is not None})   # by value
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("flat,buckets", [(False, "stage"), (True, "stage"), (True, "position")])
def test_dp_allreduce_gloo_world2(flat, buckets):
    """N > 1 path on CPU: clips sharded over 2 ranks, bucketed all-reduce (average) == single-rank gradient / 1
    of the concatenated batch divided by world (sum-of-clips loss), BN off."""
    emu_path = build_emu()
    fx = load_golden("stack_2stage")
    # single-process reference on the full batch
    emu = Lib(emu_path)
    opt = default_opt(num_tokens=4, is_bn=0)
    st = AdapterStack(fx["stages"], opt=opt, lib=emu, concurrent=False)
    st.load_state_dict(fx["state0"], strict=False)
    outs, maps = st([(a.clone(), b.clone()) for a, b in fx["feats"]])
    torch.autograd.backward([t for pr in outs for t in pr], [g for pr in fx["cots"] for g in pr])
    full = {k: p.grad.clone() for k, p in st.named_parameters() if p.grad is not None}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    port += 7 * int(flat) + 13 * int(buckets == "position")
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, emu_path, q, flat, buckets)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert set(full) <= set(got)
    for k in full:
        assert rel_err(torch.from_numpy(got[k]), full[k] / 2) < 1e-4, k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_flat_parameters_refresh_prepared_weights_after_optimizer_step(dtype):
    """the per-name tensors of a flattened adapter are views of `flat_param.data`, whose version counter an
    optimizer step never bumps -- the prepare cache must key on the flat parameter itself, or training silently keeps
    using the initial MFMA-operand weight copies / derived bias vectors."""
    emu = Lib(build_emu())
    fx = load_golden("ave_orderA")
    c = fx["cfg"]
    opt = SimpleNamespace(is_multimodal=1, num_conv_group=c["g"], is_before_layernorm=1, is_post_layernorm=1, num_tokens=c["tk"])

    def make():
        m = VisualAdapter(c["C"], c["C"], "bottleneck", reduction_factor=c["r"], opt=opt, use_bn=c["use_bn"], use_gate=c["use_gate"],
                          num_tk=c["tk"], conv_dim_in=c["No"], conv_dim_out=c["N"], linear_in=c["Co"], linear_out=c["C"],
                          flavour="ave", lib=emu, compute_dtype=dtype)
        m.load_state_dict(fx["state0"])
        return m.train()

    X = fx["X"].permute(0, 2, 1).unsqueeze(-1)
    Y = fx["Y"].permute(0, 2, 1).unsqueeze(-1)
    m = make().flatten_parameters()
    sgd = torch.optim.SGD(m.parameters(), lr=0.5)
    out0 = m(X, Y)[0]
    out0.backward(fx["dOut"].permute(0, 2, 1).unsqueeze(-1))
    sgd.step()
    out1 = m(X, Y)[0].detach()                 # must see the stepped weights
    ref = make()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    ref.load_state_dict(sd)
    out_ref = ref(X, Y)[0].detach()
    assert rel_err(out1, out_ref) < (1e-5 if dtype == torch.float32 else 2e-2)
    assert rel_err(out1, out0.detach()) > 1e-2, "the step did not change the output: stale prepared weights?"


def _late_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    big = [torch.nn.Parameter(torch.randn(8192)) for _ in range(3)]         # in-place path (numel >= 4096)
    small = [torch.nn.Parameter(torch.randn(7)) for _ in range(3)]          # torch.cat fallback path
    params = big + small
    red = GradAllReducer([big[:2] + small[:2], big[2:] + small[2:]])
    x = [torch.full_like(p, float(rank + 1 + i)) for i, p in enumerate(params)]
    loss = lambda use: sum((p * xi).sum() for p, xi, u in zip(params, x, use) if u)
    sometimes = [True, False, True, True, False, True]                      # big[1] and small[1] unused in step 1
    loss(sometimes).backward(); red.finish()                                # step 1: learns the expected counts (2 and 2)
    for p in params:
        p.grad = None
    loss([True] * 6).backward()                                             # step 2: bucket 0 launches after 2 events, 2 more arrive
    launched = red.hook_launches
    red.finish()
    out = (launched, red.relaunches, [p.grad.clone().numpy() for p in params])
    # step 3: a second backward in the same step accumulates into gradients that are already in flight -> must raise
    for p in params:
        p.grad = None
    loss([True] * 6).backward()
    loss([True] * 6).backward()
    try:
        red.finish()
        raised = False
    except RuntimeError as e:
        raised = "after the bucket" in str(e)
    if rank == 0:
        q.put(out + (raised,))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_late_gradients_are_reduced_and_dirty_buckets_raise():
    """the hooks launch a bucket when it has seen as many gradients as LAST step.  A parameter that
    only sometimes gets a gradient then arrives after the launch: finish() must still reduce it.  A gradient that is
    accumulated into again while its collective is in flight (second backward in the step) cannot be repaired: finish() raises."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_late_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    launched, relaunches, grads, raised = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert launched == 2 and relaunches == 1 and raised
    for i, g in enumerate(grads):
        want = ((1 + i) + (2 + i)) / 2                                      # mean over the two ranks of x_rank
        assert abs(float(g.min()) - want) < 1e-5 and abs(float(g.max()) - want) < 1e-5, (i, g[:3], want)


# ---------------------------------------------------------------------------------------------------------------
def _train_setup(emu, fx, lo, hi, flat):
    from dgsct_amd.train import StackTrainer, make_optimizer, seed_everything
    seed_everything(43, 0)
    st = AdapterStack(fx["stages"], opt=default_opt(num_tokens=4, is_bn=0), lib=emu, concurrent=False)
    st.load_state_dict(fx["state0"], strict=False)
    if flat:
        st.flatten_parameters()
    opt, sched = make_optimizer(st, lr=1e-2, lr_mlp=1e-3, fused=False)
    feats = [(a[lo:hi].clone(), b[lo:hi].clone()) for a, b in fx["feats"]]
    cots = [(a[lo:hi].clone() / (hi - lo), b[lo:hi].clone() / (hi - lo)) for a, b in fx["cots"]]     # mean over the local clips
    return st, opt, sched, feats, cots


def _train_worker(rank, world, port, emu_path, q, accum_itr, accum_mode="reference"):
    from dgsct_amd.train import StackTrainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    fx = load_golden("stack_2stage")
    BT = fx["feats"][0][0].shape[0]
    st, opt, sched, feats, cots = _train_setup(Lib(emu_path), fx, rank * BT // world, (rank + 1) * BT // world, True)
    red = GradAllReducer(GradAllReducer.stage_buckets(st))
    tr = StackTrainer(st, opt, red, accum_itr=accum_itr, accum_mode=accum_mode)
    calls = []
    orig = red._all_reduce
    red._all_reduce = lambda tensors, producers=(): (calls.append(tr.it), orig(tensors, producers))[1]
    stepped = [tr.step(feats, cots) for _ in range(4)]
    if rank == 0:
        # (tr.it at the time of each collective: only stepping iterations communicate -- ADVICE r2)
        q.put((stepped, {k: v.numpy().copy() for k, v in st.state_dict().items() if v.is_floating_point()}, calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("accum_itr,accum_mode", [(1, "reference"), (2, "reference"), (2, "accumulate")])
def test_training_steps_under_dp_match_single_process(accum_itr, accum_mode):
    """N optimizer steps (Adam + the reference's freeze rule and `accum_itr` control flow,
    main_trans.py:110,135-136,211-278) of the identity-backbone stack, clips sharded over 2 gloo ranks with overlapped
    bucketed gradient all-reduce, reproduce the single-process parameters.  accum_itr = 2 reproduces the reference's quirk:
    zero_grad every iteration, step every 2nd -- only the 2nd, 4th, ... batch ever reaches the weights."""
    from dgsct_amd.train import StackTrainer, shard_clips, trainable_by_name
    emu_path = build_emu()
    fx = load_golden("stack_2stage")
    BT = fx["feats"][0][0].shape[0]
    st, opt, sched, feats, cots = _train_setup(Lib(emu_path), fx, 0, BT, False)
    assert all(p.requires_grad for n, p in st.named_parameters())          # every name contains 'adapter_blocks'
    tr = StackTrainer(st, opt, None, accum_itr=accum_itr, accum_mode=accum_mode)
    stepped = [tr.step(feats, cots) for _ in range(4)]
    assert stepped == ([True] * 4 if accum_itr == 1 else [False, True, False, True])
    ref = {k: v.clone() for k, v in st.state_dict().items() if v.is_floating_point()}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + 11 * accum_itr + 5 * (accum_mode == "accumulate")
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, emu_path, q, accum_itr, accum_mode)) for r in range(2)]
    for p in procs:
        p.start()
    got_stepped, got, comm_its = q.get(timeout=240)
    # collectives are enqueued during fwd_bwd (hooks: tr.it still counts the iteration being run) or in finish() (tr.it
    # already incremented): with accum_itr = 2 nothing may be sent for the non-stepping iterations 0 and 2
    assert comm_its and all((it % accum_itr == accum_itr - 1) or (it % accum_itr == 0 and it > 0) for it in comm_its), comm_its
    if accum_itr == 2:
        assert not any(it in (0,) for it in comm_its)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert got_stepped == stepped
    moved = 0
    for k, v in ref.items():
        assert rel_err(torch.from_numpy(got[k]), v) < 2e-4, k
        moved += int((v - fx["state0"][k]).abs().max() > 1e-4) if k in fx["state0"] else 0
    assert moved > 20                                                       # the steps really changed the parameters
    # the name rule and the sampler
    class P:                                                                # noqa: E306
        requires_grad = True
    named = [(n, P()) for n in ("swin.layers.0.blocks.0.norm1.weight", "htsat.patch_embed.proj.weight", "vis_adapter_blocks_p1.0.gate",
                                "CMBS.video_input_proj.weight", "mlp_class.weight", "temporal_attn.v_fc.weight", "other.weight")]
    groups = trainable_by_name(named, lr=5e-4, lr_mlp=5e-6)
    assert [p.requires_grad for _, p in named] == [False, False, True, True, True, True, False]
    assert [g["lr"] for g in groups] == [5e-4, 5e-4, 5e-4, 5e-4, 5e-6, 5e-4, 5e-4]
    a, b = shard_clips(11, 0, 2, seed=43, epoch=3), shard_clips(11, 1, 2, seed=43, epoch=3)
    assert len(a) == len(b) == 6 and set(a) | set(b) == set(range(11))


@pytest.mark.parametrize("flavour,over", [("ave", {}), ("avqa", dict(tk=2, g=4)), ("avs_s4", {}), ("pretrain", {})])
def test_cpu_baseline_port_matches_the_explicit_oracle(flavour, over):
    """`oracle.forward_autograd` -- the op-for-op ATen port that bench.py's `cpu_baseline` leg times -- against the explicit
    forward / hand-derived backward the parity tests use (round 5: an edit of the explicit forward once leaked into the port and
    broke the default bench line on the GPU box; nothing on the CPU side ran it)."""
    from oracle import dgsct_oracle as O
    cfg = O.AdapterConfig(**{**dict(N=16, C=32, No=36, Co=16, tk=4, r=8, g=2), **O.FLAVOURS[flavour], **over})
    p = O.random_params(cfg, flavour, seed=1)
    if cfg.remap == "bicubic":
        p["_bicubic"] = O.bicubic_matrix(cfg.No, cfg.N)
    gen = torch.Generator().manual_seed(2)
    BT = 2 * max(cfg.T, 1)
    X, Y = torch.randn(BT, cfg.N, cfg.C, generator=gen), torch.randn(BT, cfg.No, cfg.Co, generator=gen)
    dOut, dMap = torch.randn(BT, cfg.N, cfg.C, generator=gen), torch.randn(BT, cfg.N, generator=gen)
    out_o, map_o, _, s = O.forward({k: v.clone() for k, v in p.items()}, X, Y, cfg, training=True)
    dX_o, dY_o, _ = O.backward({k: v.clone() for k, v in p.items()}, s, cfg, dOut, dMap, None, training=True)
    pa = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and not k.startswith("_") else v.clone())
          for k, v in p.items()}
    Xa, Ya = X.clone().requires_grad_(True), Y.clone().requires_grad_(True)
    out, amap, _ = O.forward_autograd(pa, Xa, Ya, cfg, training=True)
    torch.autograd.backward([out, amap], [dOut.reshape(out.shape), dMap.reshape(amap.shape)])
    assert rel_err(out.reshape(out_o.shape), out_o) < 1e-4 and rel_err(amap.reshape(map_o.shape), map_o) < 1e-4
    assert rel_err(Xa.grad, dX_o) < 1e-4 and rel_err(Ya.grad, dY_o) < 1e-4


@pytest.mark.parametrize("flavour", ["ave", "avs_s4", "avs_ms3", "pretrain", "avqa"])
@pytest.mark.parametrize("shape", [(16, 32, 36, 64), (36, 64, 16, 32)], ids=["orderA_or_B", "swapped"])
def test_rounding_aware_oracle_without_rounding_is_the_oracle(shape, flavour):
    """oracle/dgsct_oracle_bf16.evaluate() -- the oracle's arithmetic with a switchable bf16 rounding at every tensor the bf16 schedule
    stores, the yardstick of tests/test_bf16_masked_gpu.py -- is pinned here: with nothing rounded it must reproduce the (reference-pinned)
    oracle's forward and backward exactly, on its own ReLU decisions and on pinned ones; with the device's rounding points switched on it
    must move (the switch is connected) but stay within bf16 distance."""
    from oracle import dgsct_oracle as O
    from oracle import dgsct_oracle_bf16 as OB
    N, C, No, Co = shape
    BT = 3
    cfg = O.AdapterConfig(**{**dict(N=N, C=C, No=No, Co=Co, tk=4, r=8, g=2), **O.FLAVOURS[flavour]})
    p = O.random_params(cfg, flavour, seed=3, scale=0.577)
    if cfg.remap == "bicubic":
        p["_bicubic"] = O.bicubic_matrix(No, N)
    gen = torch.Generator().manual_seed(4)
    X, Y = torch.randn(BT, N, C, generator=gen), torch.randn(BT, No, Co, generator=gen)
    dOut, dMap = torch.randn(BT, N, C, generator=gen), torch.randn(BT, N, generator=gen)
    out_o, map_o, _, s = O.forward({k: v.clone() for k, v in p.items()}, X, Y, cfg, training=True)
    dTm = torch.randn(BT, generator=gen) if cfg.temporal else None
    dX_o, dY_o, g_o = O.backward(p, s, cfg, dOut, dMap, dTm, training=True)
    r = OB.evaluate(cfg, p, X, Y, dOut, dMap, OB.Q([]), dTmap=dTm)
    assert rel_err(r["out"], out_o) < 1e-5 and rel_err(r["map"], map_o) < 1e-5
    assert rel_err(r["dX"], dX_o) < 1e-5 and rel_err(r["dY"], dY_o) < 1e-5
    assert set(r["g"]) == {k for k, v in g_o.items() if v is not None}
    for k, g in r["g"].items():
        if k == "gate" and cfg.gate_before_ln_post:
            continue                         # (LN(gate * O): analytically ~0, the oracle evaluates this one residue in float64)
        assert rel_err(g.reshape(g_o[k].shape), g_o[k]) < 2e-5, k
    # pinned decisions (its own, fed back in): the same evaluation
    r2 = OB.evaluate(cfg, p, X, Y, dOut, dMap, OB.Q([]), masks=r["masks"], dTmap=dTm)
    assert rel_err(r2["dX"], dX_o) < 1e-5 and rel_err(r2["dY"], dY_o) < 1e-5
    rq = OB.evaluate(cfg, p, X, Y, dOut, dMap, OB.Q(OB.DEVICE_ROUNDING), masks=r["masks"], dTmap=dTm)
    e = float((rq["dY"] - dY_o).norm() / dY_o.norm())
    assert 1e-4 < e < 5e-2, e


def test_pair_node_in_eval_mode_without_grad_and_over_two_backward_passes():
    """the one-node path of a position outside its training-step comfort zone (host-emulated library): eval mode under no_grad (BatchNorm
    running statistics, no autograd context), a mixed stack in which one adapter of a position is not flattened (that position falls back to
    two nodes), and two backward passes accumulating into the same flat gradients."""
    emu = Lib(build_emu())
    fx = load_golden("stack_2stage")
    feats = [(a.clone(), b.clone()) for a, b in fx["feats"]]

    def make(pair, flatten="all"):
        st = AdapterStack(fx["stages"], opt=default_opt(num_tokens=4), lib=emu, concurrent=False, pair_backward=pair)
        st.load_state_dict(fx["state0"])
        if flatten == "all":
            st.flatten_parameters()
        elif flatten == "mixed":
            for ml in (st.audio_adapter_blocks_p1, st.vis_adapter_blocks_p1, st.audio_adapter_blocks_p2):
                for m in ml:
                    m.flatten_parameters()
            st.vis_adapter_blocks_p2[0].flatten_parameters()          # every other p2 visual adapter keeps its ~30 tensors
        return st

    # eval + no_grad
    a, b = make(False).eval(), make(True).eval()
    with torch.no_grad():
        oa, ma = a(feats)
        ob, mb = b(feats)
    for (x1, y1), (x2, y2) in zip(oa, ob):
        assert torch.equal(x1, x2) and torch.equal(y1, y2)
    assert torch.equal(ma[0], mb[0]) and torch.equal(ma[1], mb[1])

    # mixed flattening + two accumulating passes
    def grads_after_two_passes(st):
        st.train()
        for _ in range(2):
            outs, maps = st(feats)
            torch.autograd.backward([t for pr in outs for t in pr] + [maps[0], maps[1]],
                                    [g for pr in fx["cots"] for g in pr] + [fx["mcots"][0], fx["mcots"][1]])
        out = {}
        for name, m in st.named_modules():
            if hasattr(m, "flat_param") and "_flat_views" in m.__dict__:
                for pn, (off, cnt, shape) in m._flat_layout.items():
                    out[name + "." + pn] = m.flat_param.grad[off:off + cnt].view(shape).clone()
        for k, p in st.named_parameters():
            if p.grad is not None and not k.endswith("flat_param"):
                out[k] = p.grad.clone()
        return out
    g0 = grads_after_two_passes(make(False, "mixed"))
    g1 = grads_after_two_passes(make(True, "mixed"))
    assert set(g0) == set(g1) and len(g0) >= len(fx["grads"])
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 1e-5, k
