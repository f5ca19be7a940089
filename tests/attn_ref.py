"""torch fp32 reference of the four fused latent-token attention kernels (csrc/attn.hip) + the ctypes plumbing to call one
of them through dgsct_test_attn.  Shared by tests/test_attn_gpu.py and tools/attn_bench.py."""
import torch

from dgsct_amd._lib import AttnArgs


def ref_tokattn_fwd(Yp, T0):
    S = torch.einsum("tc,bnc->btn", T0, Yp)
    P = torch.softmax(S, -1)
    return T0[None] + P @ Yp, torch.logsumexp(S, -1), Yp.mean(1)


def ref_xattn_fwd(X, tok, g):
    P = torch.softmax(X @ tok.transpose(1, 2), -1)
    return X + g * (P @ tok)


def ref_xattn_bwd(X, dX1, tok, g, R2=None):
    P = torch.softmax(X @ tok.transpose(1, 2), -1)
    U = dX1 @ tok.transpose(1, 2)
    dgate = (P * U).sum()
    dS = P * (g * U - (P * g * U).sum(-1, keepdim=True))
    dX = dX1 + dS @ tok
    if R2 is not None:
        dX = dX + R2
    dtok = g * (P.transpose(1, 2) @ dX1) + dS.transpose(1, 2) @ X
    return dX, dtok, dgate


def ref_tokattn_bwd(Yp, T0, dtok, da, invN):
    S = torch.einsum("tc,bnc->btn", T0, Yp)
    P = torch.softmax(S, -1)
    dP = dtok @ Yp.transpose(1, 2)
    dS = P * (dP - (P * dP).sum(-1, keepdim=True))
    dYp = P.transpose(1, 2) @ dtok + torch.einsum("btn,tc->bnc", dS, T0) + (da * invN)[:, None, :]
    dT0b = dS @ Yp
    return dYp, dT0b


def _p(t):
    return t.data_ptr() if t is not None else None


class AttnCall:
    """device buffers + argument struct of one dgsct_test_attn call"""

    def __init__(self, lib, dtype, B, N, C, tk, dev, seed=0, g=0.3):
        gen = torch.Generator().manual_seed(seed)
        r = lambda *s: torch.randn(*s, generator=gen)
        self.lib, self.dtype, self.dims = lib, dtype, (B, N, C, tk)
        self.X, self.Yp, self.dX1, self.R2 = (r(B, N, C).to(dtype).float() for _ in range(4))
        self.Yp = (self.Yp * 0.4).to(dtype).float()
        self.T0 = torch.rand(tk, C, generator=gen)
        self.dtok_in = r(B, tk, C)
        self.da = r(B, C)
        self.g = g
        d = lambda t, dt=None: t.to(dev, dt or torch.float32).contiguous()
        self.d = dict(X=d(self.X, dtype), Yp=d(self.Yp, dtype), dX1=d(self.dX1, dtype), R2=d(self.R2, dtype), T0=d(self.T0),
                      da=d(self.da), gate_av=torch.tensor([g], device=dev), out=torch.empty(B, N, C, device=dev, dtype=dtype),
                      tok=torch.empty(B, tk, C, device=dev), lse=torch.empty(B, tk, device=dev), a=torch.zeros(B, C, device=dev),
                      aE=torch.empty(B, C, device=dev, dtype=dtype), dtok=torch.zeros(B, tk, C, device=dev),
                      dgate=torch.zeros(1, device=dev), dT0b=torch.zeros(B, tk, C, device=dev),
                      scratch=torch.empty(int(lib.c.dgsct_test_attn_scratch_floats(B, N, C, tk)), device=dev),
                      tokpk=torch.zeros(96 * B * C, device=dev, dtype=torch.bfloat16),
                      T0pk=torch.zeros(96 * C, device=dev, dtype=torch.bfloat16),
                      dtokpk=torch.zeros(96 * B * C, device=dev, dtype=torch.bfloat16))
        a = AttnArgs()
        a.mode = 1 if dtype == torch.bfloat16 else 0
        a.B, a.N, a.C, a.tk = B, N, C, tk
        for k, v in self.d.items():
            setattr(a, k, _p(v))
        a.invN = 1.0 / N
        self.args = a

    def run(self, op, stream=None):
        import torch as _t
        s = _t.cuda.current_stream().cuda_stream if stream is None else stream
        self.lib.test_attn(op, self.args, s)
