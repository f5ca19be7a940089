"""TEST INFRASTRUCTURE (like everything under oracle/): a ROUNDING-AWARE evaluation of the adapter oracle.

What it is.  `dgsct_oracle.forward / backward` restate the reference's VisualAdapter (net_trans.py:552-674) in fp32.  The device's bf16
path differs from that by (a) ReLU decisions that fall the other way for pre-activations within rounding of zero and (b) the bf16
rounding of every tensor it stores or feeds to an MFMA.  (a) is removed by pinning the device's own decisions (`masks=`).  (b) is
DETERMINISTIC: rounding `Wn`, `T = Wn Y`, `Yp`, ... to bf16 gives the same values on the device and here, up to the rare element whose
fp32 accumulation order moves it across a rounding boundary.  `evaluate()` is the oracle's arithmetic with a switchable rounding
`q(name, x)` at each such tensor; with the device's rounding points (`DEVICE_ROUNDING`) what is left between it and the device is
accumulation order and the handful of points where the two disagree about what is kept in fp32 -- under 1.1 % relative L2 on every
gradient at every AVE width (5.9 % against the fp32 oracle at C = 1024), which is what `tests/test_bf16_masked_gpu.py` asserts.

How it is pinned.  `evaluate(..., Q([]))` (nothing rounded) must equal `dgsct_oracle.forward / backward` -- asserted on CPU by
`tests/test_host_cpu.py::test_rounding_aware_oracle_without_rounding_is_the_oracle` -- and that oracle is pinned against vectors generated
by the reference itself (oracle/make_golden.py).  Restated here: the 'ave' / 'avvp' flavour (conv remap; every BASELINE config) and the AVS
flavours' differences (bicubic remap operator, no ln_before, gate in front of ln_post), the temporal gate of 'pretrain', 'avqa' without
BatchNorm; not eval-mode BN.

Names of the rounding points: W:<weight> bf16 weight copies; T, Yp remap intermediate / result; T0 my_tokens; P1, P2 the two softmaxes;
tok latent tokens (tokS: as the logit operand, tokV: as the value operand); X1 (X1m: the copy the modulation reads), aE, aq, vq1, m1, q,
Xc, vq2, X3, Zp, Z, Op, out; cotangents dO, dZ, dX3, dX1, dvq2, dXc, dpre, dvq1, dS2, dX, dtok, dS1, dYp, dT, dY.
"""
import torch
import torch.nn.functional as F

from . import dgsct_oracle as O

WEIGHTS = ['W:Wn', 'W:Wc', 'W:Wd', 'W:Wu', 'W:audio_1', 'W:audio_2', 'W:video_1', 'W:video_2', 'W:bottleneck', 'W:v_c_att']
ALL = WEIGHTS + ['T', 'Yp', 'T0', 'P1', 'tok', 'tokS', 'tokV', 'P2', 'P2m', 'X1', 'X1m', 'aE', 'aq', 'vq1', 'm1', 'q', 'Xc', 'vq2', 'X3', 'Zp', 'Z',
                 'Op', 'out', 'dO', 'dZ', 'dX3', 'dX1', 'dvq2', 'dXc', 'dpre', 'dvq1', 'dS2', 'dX', 'dtok', 'dS1', 'dYp', 'dT', 'dY']
# What the device keeps in fp32: the latent tokens (`my_tokens`, `tok` and its use in the logit products) and the two
# softmaxes' probabilities (registers of the attention kernels).  Everything else in ALL is a bf16 tensor in `saved` / the workspace or a
# bf16 MFMA operand.
DEVICE_FP32 = ['T0', 'tok', 'tokS', 'P1', 'P2']
DEVICE_ROUNDING = [n for n in ALL if n not in DEVICE_FP32]


class Q:
    def __init__(self, names): self.names=set(names)
    def __call__(self, name, x):
        if name in self.names or '*' in self.names and ('-'+name) not in self.names: return x.bfloat16().float()
        return x
def evaluate(cfg,p,X,Y,dOut,dMap,q,masks=None,dTmap=None):
    """forward + backward of the adapter (cfg.remap conv | bicubic, cfg.ln_before, cfg.gate_before_ln_post, cfg.temporal, cfg.use_bn; BatchNorm
    in training mode) with `q(name, tensor)` applied at
    every tensor the bf16 schedule stores or feeds to an MFMA; masks: pinned ReLU decisions (keys aq1, aq2, vq1, q, vq2, Z) or None.
    Returns dict(out, map, dX, dY, g = {parameter name: gradient}, masks = the ReLU decisions used)."""
    B,N,C = X.shape; R=B*N
    used = {}
    def M(name, x):            # the ReLU decision of `name`: this run's own, or a pinned one (masks=...)
        m = masks[name] if masks is not None else (x > 0)
        used[name] = m
        return m
    relu = lambda name, x: x * M(name, x)
    W = lambda n: q('W:'+n.replace('fc_affine_','').replace('.weight',''), p[n])    # bf16 weight copies
    Wc,bc = p['fc.weight'],p['fc.bias']
    conv = cfg.remap == 'conv'
    if conv:
        Wn = p['conv_adapter.weight'].reshape(N,cfg.No); bn = p['conv_adapter.bias']
        rowb,colb,colb2 = bn, Wc.sum(1), bc
    else:                                          # AVS-S4: fc, then the fixed bicubic operator (dgsct_oracle.forward, F1)
        Wn = p['_bicubic']; bn = None
        rowb,colb,colb2 = Wn.sum(1), bc, torch.zeros_like(bc)
    order = cfg.remap_order()
    if order=='A':
        T1 = q('T', torch.einsum('mn,bnk->bmk', q('W:Wn',Wn), Y)); Yp = T1 @ q('W:Wc',Wc).t()
    else:
        T2t = q('T', torch.einsum('ck,bnk->bcn', q('W:Wc',Wc), Y)); Yp = torch.einsum('mn,bcn->bmc', q('W:Wn',Wn), T2t)
    Yp = q('Yp', Yp + rowb[None,:,None]*colb[None,None,:] + colb2)
    T0 = p['my_tokens']
    S1 = torch.einsum('tc,bnc->btn', q('T0',T0), Yp)
    P1f = torch.softmax(S1,-1); P1 = q('P1',P1f)
    tok = q('tok', T0[None] + P1 @ Yp)
    a = Yp.mean(1)
    S2 = X @ q('tokS',tok).transpose(1,2)
    P2 = q('P2', torch.softmax(S2,-1))
    gav = p['gate_av']
    X1full = X + gav*(q('P2m',torch.softmax(S2,-1)) @ q('tokV',tok)); X1 = q('X1', X + gav*(P2 @ q('tokV',tok))); X1m = q('X1m', X1full)
    aE = q('aE',a)
    aq1 = q('aq', relu('aq1', F.linear(aE, W('fc_affine_audio_1.weight'), p['fc_affine_audio_1.bias'])))
    aq2 = q('aq', relu('aq2', F.linear(aE, W('fc_affine_audio_2.weight'), p['fc_affine_audio_2.bias'])))
    vq1 = q('vq1', relu('vq1', F.linear(X1, W('fc_affine_video_1.weight'), p['fc_affine_video_1.bias'])))
    mvq1 = vq1.mean(1); m1 = q('m1', aq1*mvq1)
    qq = q('q', relu('q', F.linear(m1, W('fc_affine_bottleneck.weight'), p['fc_affine_bottleneck.bias'])))
    ch = torch.sigmoid(F.linear(qq, W('fc_affine_v_c_att.weight'), p['fc_affine_v_c_att.bias']))
    Xc = q('Xc', X1*(1+ch[:,None,:]))
    vq2 = q('vq2', relu('vq2', F.linear(Xc, W('fc_affine_video_2.weight'), p['fc_affine_video_2.bias'])))
    ws,bs = p['fc_affine_v_s_att.weight'].reshape(-1), p['fc_affine_v_s_att.bias']
    sl = (vq2*(aq2*ws)[:,None,:]).sum(-1)+bs; sg = torch.sigmoid(sl); amap = torch.softmax(torch.tanh(sl),-1)
    mod = cfg.alpha*ch[:,None,:] + cfg.beta*sg[:,:,None] + (1-cfg.alpha)
    tg = None
    if cfg.temporal:                               # pretrain flavour (dgsct_oracle.forward: temporal gate), fp32 on the device
        wt, bt = p['temporal_gated.0.weight'].reshape(-1), p['temporal_gated.0.bias']
        tg = torch.sigmoid(a @ wt + bt); mod = mod + cfg.gamma*tg[:,None,None]
    X2 = X1m*mod
    if cfg.ln_before:
        X3f, xh_b, rstd_b = O._ln(X2, p['ln_before.weight'], p['ln_before.bias'], cfg.eps)
    else:
        X3f = X2
    X3 = q('X3',X3f)
    Wd = p['down_sampler.weight'].reshape(cfg.ds, C//cfg.g); Wu = p['up_sampler.weight'].reshape(C, cfg.ds//cfg.g)
    Zp = q('Zp', O._groupmm(X3, q('W:Wd',Wd), cfg.g))
    def bn_(x,name):
        w,b = p[name+'.weight'],p[name+'.bias']; xf=x.reshape(R,-1); mu=xf.mean(0); var=((xf-mu)**2).mean(0); rstd=torch.rsqrt(var+cfg.eps); xh=(x-mu)*rstd
        return xh*w+b, xh, rstd
    Zb,zh,rstd1 = bn_(Zp,'bn1') if cfg.use_bn else (Zp,None,None); Z = q('Z',relu('Z', Zb))
    Op = q('Op', O._groupmm(Z, q('W:Wu',Wu), cfg.g))
    Oo,oh,rstd2 = bn_(Op,'bn2') if cfg.use_bn else (Op,None,None)
    gate = p['gate']
    g={}
    if cfg.gate_before_ln_post:                    # AVS order: gate, then ln_post (dgsct_oracle.forward F11 / backward B11)
        G = Oo*gate
        L,xh_p,rstd_p = O._ln(G, p['ln_post.weight'], p['ln_post.bias'], cfg.eps); out = q('out', L)
        dG,g['ln_post.weight'],g['ln_post.bias'] = O._ln_bwd(dOut,xh_p,rstd_p,p['ln_post.weight'])
        g['gate'] = (dG*Oo).sum().reshape(1)       # (a cancellation residue: not compared)
        dO = q('dO', dG*gate)
    else:
        L,xh_p,rstd_p = O._ln(Oo, p['ln_post.weight'], p['ln_post.bias'], cfg.eps)
        out = q('out', L*gate)
        g['gate']=(dOut*L).sum().reshape(1); dL = dOut*gate
        dO,g['ln_post.weight'],g['ln_post.bias'] = O._ln_bwd(dL,xh_p,rstd_p,p['ln_post.weight']); dO = q('dO',dO)
    def bn_bwd(dy,xh,rstd,name):
        w=p[name+'.weight']; dyf,xhf=dy.reshape(R,-1),xh.reshape(R,-1); dw=(dyf*xhf).sum(0); db=dyf.sum(0)
        g[name+'.weight'],g[name+'.bias']=dw,db
        return w*rstd*(dy-db/R-xh*(dw/R))
    dOp = q('dO', bn_bwd(dO,oh,rstd2,'bn2')) if cfg.use_bn else dO
    dZ,dWu = O._groupmm_bwd(dOp,Z,q('W:Wu',Wu),cfg.g); g['up_sampler.weight']=dWu.reshape(p['up_sampler.weight'].shape); dZ=q('dZ',dZ)
    dZb = dZ*used['Z']; dZp = q('dZ', bn_bwd(dZb,zh,rstd1,'bn1')) if cfg.use_bn else q('dZ', dZb)
    dX3,dWd = O._groupmm_bwd(dZp,X3,q('W:Wd',Wd),cfg.g); g['down_sampler.weight']=dWd.reshape(p['down_sampler.weight'].shape); dX3=q('dX3',dX3)
    if cfg.ln_before:
        dX2,g['ln_before.weight'],g['ln_before.bias'] = O._ln_bwd(dX3,xh_b,rstd_b,p['ln_before.weight'])
    else:
        dX2 = dX3
    dX1 = q('dX1', dX2*mod); dmod = dX2*X1m
    dch = cfg.alpha*dmod.sum(1); dsg = cfg.beta*dmod.sum(2)
    da_t = None
    if cfg.temporal:
        dtg = cfg.gamma*dmod.sum((1,2))
        if dTmap is not None: dtg = dtg + dTmap
        dpre_t = dtg*tg*(1-tg)
        g['temporal_gated.0.weight'] = (dpre_t[:,None]*a).sum(0).reshape(p['temporal_gated.0.weight'].shape)
        g['temporal_gated.0.bias'] = dpre_t.sum().reshape(1)
        da_t = dpre_t[:,None]*wt[None,:]
    dsl = dsg*sg*(1-sg); dt = amap*(dMap-(amap*dMap).sum(-1,keepdim=True)); dsl = dsl + dt*(1-torch.tanh(sl)**2)
    u = (dsl[:,:,None]*vq2).sum(1); g['fc_affine_v_s_att.bias']=dsl.sum().reshape(1); g['fc_affine_v_s_att.weight']=(u*aq2).sum(0)
    daq2 = u*ws
    dvq2 = q('dvq2', dsl[:,:,None]*(aq2*ws)[:,None,:]*used['vq2'])
    dXc = q('dXc', dvq2 @ W('fc_affine_video_2.weight'))
    g['fc_affine_video_2.weight'] = dvq2.reshape(R,-1).t() @ Xc.reshape(R,C); g['fc_affine_video_2.bias']=dvq2.reshape(R,-1).sum(0)
    dX1 = q('dX1', dX1 + dXc*(1+ch[:,None,:])); dch = dch + (dXc*X1).sum(1)
    dpre_c = q('dpre', dch*ch*(1-ch))
    g['fc_affine_v_c_att.weight']=dpre_c.t()@qq; g['fc_affine_v_c_att.bias']=dpre_c.sum(0)
    dq = q('dpre', (dpre_c @ W('fc_affine_v_c_att.weight'))*used['q'])
    g['fc_affine_bottleneck.weight']=dq.t()@m1; g['fc_affine_bottleneck.bias']=dq.sum(0)
    dm1 = dq @ W('fc_affine_bottleneck.weight'); daq1 = dm1*mvq1; dmvq1 = dm1*aq1
    dvq1 = q('dvq1', (dmvq1/N)[:,None,:]*used['vq1'])
    dX1 = q('dX1', dX1 + dvq1 @ W('fc_affine_video_1.weight'))
    g['fc_affine_video_1.weight']=dvq1.reshape(R,C).t()@X1.reshape(R,C); g['fc_affine_video_1.bias']=dvq1.reshape(R,C).sum(0)
    dpa1 = q('dpre', daq1*used['aq1']); dpa2 = q('dpre', daq2*used['aq2'])
    g['fc_affine_audio_1.weight']=dpa1.t()@aE; g['fc_affine_audio_1.bias']=dpa1.sum(0); g['fc_affine_audio_2.weight']=dpa2.t()@aE; g['fc_affine_audio_2.bias']=dpa2.sum(0)
    da = dpa1 @ W('fc_affine_audio_1.weight') + dpa2 @ W('fc_affine_audio_2.weight')
    if da_t is not None: da = da + da_t
    U = dX1 @ q('tokS',tok).transpose(1,2)
    g['gate_av']=(P2*U).sum().reshape(1); dP2 = gav*U
    dS2 = q('dS2', P2*(dP2-(P2*dP2).sum(-1,keepdim=True)))
    dX = q('dX', dX1 + dS2 @ q('tokV',tok))
    dtok = gav*(P2.transpose(1,2)@dX1) + dS2.transpose(1,2)@X
    dtokE = q('dtok',dtok)
    dP1 = dtokE @ Yp.transpose(1,2)
    dS1 = q('dS1', P1*(dP1-(P1*dP1).sum(-1,keepdim=True)))
    g['my_tokens'] = dtok.sum(0) + torch.einsum('btn,bnc->tc',dS1,Yp)
    dYp = q('dYp', P1.transpose(1,2)@dtokE + torch.einsum('btn,tc->bnc',dS1,q('T0',T0)) + (da/N)[:,None,:])
    if conv:
        wcsum = Wc.sum(1)
        g['fc.bias']=dYp.sum((0,1)); g['conv_adapter.bias']=torch.einsum('bmc,c->m',dYp,wcsum); dwcsum=torch.einsum('bmc,m->c',dYp,bn)
    else:
        g['fc.bias']=torch.einsum('bmc,m->c',dYp,Wn.sum(1)); dwcsum=None
    if order=='A':
        dT1 = q('dT', dYp @ q('W:Wc',Wc)); dWc = torch.einsum('bmc,bmk->ck',dYp,T1); dY = torch.einsum('mn,bmk->bnk',q('W:Wn',Wn),dT1); dWn = torch.einsum('bmk,bnk->mn',dT1,Y)
    else:
        dT2t = q('dT', torch.einsum('bmc,mn->bcn',dYp,q('W:Wn',Wn))); dWn = torch.einsum('bmc,bcn->mn',dYp,T2t); dY = torch.einsum('bcn,ck->bnk',dT2t,q('W:Wc',Wc)); dWc = torch.einsum('bcn,bnk->ck',dT2t,Y)
    g['fc.weight']=dWc+dwcsum[:,None] if dwcsum is not None else dWc
    if conv: g['conv_adapter.weight']=dWn
    return dict(out=out,map=amap,dX=dX,dY=q('dY',dY),g=g,masks=used)
