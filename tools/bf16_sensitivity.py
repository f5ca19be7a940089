"""Where does the bf16 gradient error of the adapter come from?  (evidence for DESIGN.md section 7; TEST TOOLING -- imports oracle/)

The oracle's forward/backward with a switchable bf16 rounding at every tensor the bf16 schedule stores or feeds to an MFMA
(`Q([...names...])`), at one real AVE shape against the un-rounded fp32 evaluation.  Findings it reproduces (rel-L2 of dX / dY):

* rounding ONLY X1 (the adapter's main-path activation) to bf16 costs ~3 % on dX, ONLY X3 ~2.6 %, ONLY the bottleneck weight
  Wd ~2.7 %, while rounding the bottleneck pre-activation Zp itself costs 0.4 %: the error is not proportional to the
  perturbation -- the bottleneck ReLU flips the mask of the units whose pre-activation the perturbation carries across zero
  (a fraction ~eps of them), and each flip changes that unit's gradient by 100 %: error ~ sqrt(eps);
* rounding ONLY the remap weights Wn / Wc, or ONLY Yp, moves dY by 1.5-3 % at C = 512 and by 7-10 % at C = 1024: the
  latent-token softmaxes are un-scaled (logits ~ sqrt(C) ~ 18-32), so a 2^-9 perturbation of Yp / tok is amplified ~20x
  before it reaches X1 and that ReLU;
* keeping the main path X1 -> X3 -> Zp, the latent tokens and Wd in fp32 ("new scheme") takes dX from ~4.5 % to ~2-3 %;
  the rest needs Yp / T / Wn / Wc un-rounded, i.e. the remap GEMMs (half of all FLOPs) at 2-3x their bf16 cost.

usage: python tools/bf16_sensitivity.py [N,C,No,Co]        (default 144,512,256,384; CPU, ~1 min)
"""
import sys, torch, torch.nn.functional as F
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dgsct_oracle as O
torch.set_num_threads(8)
def mk(N,C,No,Co,BT=10,seed=0):
    cfg = O.AdapterConfig(N=N,C=C,No=No,Co=Co,tk=32,r=8,g=2)
    p = O.random_params(cfg,'ave',seed=seed,scale=0.577)
    gen = torch.Generator().manual_seed(seed+1)
    X = torch.randn(BT,N,C,generator=gen).bfloat16().float(); Y = torch.randn(BT,No,Co,generator=gen).bfloat16().float()
    dOut = torch.randn(BT,N,C,generator=gen).bfloat16().float(); dMap = torch.randn(BT,N,generator=gen)
    return cfg,p,X,Y,dOut,dMap
from oracle.dgsct_oracle_bf16 import Q, evaluate as run
def l2(a,b): return ((a-b).norm()/b.norm().clamp_min(1e-30)).item()
def report(tag, r, ref):
    gs = sorted(((l2(r['g'][k].reshape(-1), ref['g'][k].reshape(-1)),k) for k in ref['g'] if k not in('ln_before.bias',)), reverse=True)
    print(f"{tag:34s} out {l2(r['out'],ref['out']):.4f} dX {l2(r['dX'],ref['dX']):.4f} dY {l2(r['dY'],ref['dY']):.4f} | "+' '.join(f"{k.replace('fc_affine_','').replace('.weight','.w').replace('.bias','.b')}:{e:.3f}" for e,k in gs[:6]))



def pinned_table(shape):
    """would computing the two un-scaled softmax logits from an UN-ROUNDED Yp take the C-linear term out of the
    bf16 backward?  Each scenario rounds a subset of tensors; the reference is the fp32 evaluation differentiated on the
    scenario's own ReLU decisions (masks pinned, as tests/test_bf16_masked_gpu.py does), so what is left is rounding, not flips."""
    args = mk(*shape)
    WS = ['W:Wn','W:Wc','W:Wd','W:Wu','W:audio_1','W:audio_2','W:video_1','W:video_2','W:bottleneck','W:v_c_att']
    remap = ['W:Wn','W:Wc','T']
    scen = [("only Yp rounded", Q(['Yp'])), ("only the remap operands (Wn, Wc, T) rounded", Q(remap)),
            ("Yp + remap operands", Q(['Yp'] + remap)),
            ("everything the bf16 schedule rounds", Q(['*','-tok','-tokS','-tokV','-T0','-X1m','-P2m'])),
            ("  ... but Yp kept in fp32", Q(['*','-tok','-tokS','-tokV','-T0','-X1m','-P2m','-Yp'])),
            ("  ... but Yp AND the remap operands in fp32", Q(['*','-tok','-tokS','-tokV','-T0','-X1m','-P2m','-Yp','-W:Wn','-W:Wc','-T'])),
            ("  ... but only the remap operands in fp32 (Yp rounded)", Q(['*','-tok','-tokS','-tokV','-T0','-X1m','-P2m','-W:Wn','-W:Wc','-T']))]
    print(f"shape (N,C,No,Co) = {shape}; rel-L2 against the fp32 evaluation on the scenario's own ReLU masks")
    for tag, q in scen:
        r = run(*args, q)
        ref = run(*args, Q([]), masks=r['masks'])
        gk = ['conv_adapter.weight','fc.weight','my_tokens']
        print(f"{tag:58s} dX {l2(r['dX'],ref['dX']):.4f}  dY {l2(r['dY'],ref['dY']):.4f}  " + ' '.join(f"{k}:{l2(r['g'][k].reshape(-1),ref['g'][k].reshape(-1)):.4f}" for k in gk))


if __name__=='__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'pinned':
        for shp in ([tuple(int(x) for x in a.split(',')) for a in sys.argv[2:]] or [(144,512,256,384),(36,1024,64,768)]):
            pinned_table(shp)
        sys.exit(0)
    shape = tuple(int(x) for x in sys.argv[1].split(',')) if len(sys.argv)>1 else (144,512,256,384)
    args = mk(*shape)
    WS = ['W:Wn','W:Wc','W:Wd','W:Wu','W:audio_1','W:audio_2','W:video_1','W:video_2','W:bottleneck','W:v_c_att']
    ALL = WS+['T','Yp','T0','P1','tok','tokS','tokV','P2','P2m','X1','X1m','aE','aq','vq1','m1','q','Xc','vq2','X3','Zp','Z','Op','out','dO','dZ','dX3','dX1','dvq2','dXc','dpre','dvq1','dS2','dX','dtok','dS1','dYp','dT','dY']
    def without(*xs): return [a for a in ALL if a not in xs]
    ref = run(*args, Q([]))
    print('== one tensor rounded at a time (generic fp32 weights)')
    for n in ['X1m','X3','W:Wd','Zp','W:Wn','W:Wc','Yp','T','tok','P1','P2']:
        report('only '+n, run(*args,Q([n] + (['X1'] if n=='X1m' else []))), ref)
    NEW = without('X1m','P2m','X3','W:Wd','tok','tokS','tokV','T0')
    for wexact in (False, True):
        cfg,p,X,Y,dOut,dMap = args
        if wexact:
            p = {k:(v.bfloat16().float() if v.is_floating_point() and v.dim()>=2 else v) for k,v in p.items()}
        a2 = (cfg,p,X,Y,dOut,dMap)
        ref = run(*a2, Q([]))
        print('== weights bf16-representable' if wexact else '== generic fp32 weights')
        report('  every stored tensor bf16', run(*a2,Q(ALL)), ref)
        report('  fp32 main path + latent tokens + Wd', run(*a2,Q(NEW)), ref)
        report('  ... + Yp, T un-rounded', run(*a2,Q([n for n in NEW if n not in('P1','Yp','T')])), ref)
