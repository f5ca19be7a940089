"""Drop-in ``TemporalAttention`` -- the post-backbone temporal gating of the AVE / AVVP / AVS models.

Mirrors reference ``DG-SCT/AVE/nets/net_trans.py:182-251`` (called at ``:928-930`` on the map-pooled per-frame features;
same class in ``DG-SCT/AVVP/nets/mgn.py:107-159`` and, once per decoder scale, ``avs_s4/model/PVT_AVSModel.py:447-582``)
and its building blocks ``RNNEncoder`` / ``InternalTemporalRelationModule`` / ``CrossModalRelationAttModule``
(``net_trans.py:44-92``) and ``Encoder`` / ``Decoder`` / ``EncoderLayer`` / ``DecoderLayer`` (``nets/models.py:14-170``):
same constructor defaults, same attribute names -- hence the same ``state_dict`` keys, including the never-called
``encoder_layer`` / ``decoder_layer`` prototypes the reference keeps next to their deep copies -- and the same forward.

Arithmetic: the bi-LSTMs, multi-head attentions, LayerNorms and Linears on ``[B, 10, <= 1536]`` are stock PyTorch-ROCm
(negligible cost); the tail of the forward -- the two ``Linear(d_model, 1) + Sigmoid`` gates, the
``x + gate * x * gamma`` updates and the product gate (``:240-249``) -- is ONE hand-written HIP kernel each way
(``dgsct_temporal_gate_forward/backward``, csrc/temporal.hip) when the tensors are on a ROCm device; CPU tensors raise.
"""
from __future__ import annotations

import copy

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Dropout, LayerNorm, Linear, ModuleList, MultiheadAttention

from . import _lib
from .ops import _dev_guard


def _get_clones(module, N):
    return ModuleList([copy.deepcopy(module) for _ in range(N)])


class EncoderLayer(nn.Module):
    """nets/models.py:74-113"""

    def __init__(self, d_model, nhead, dim_feedforward=1024, dropout=0.1):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = Linear(d_model, dim_feedforward)
        self.dropout = Dropout(dropout)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.norm1 = LayerNorm(d_model)
        self.norm2 = LayerNorm(d_model)
        self.dropout1 = Dropout(dropout)
        self.dropout2 = Dropout(dropout)

    def forward(self, src):
        src = self.norm1(src + self.dropout1(self.self_attn(src, src, src)[0]))
        return self.norm2(src + self.dropout2(self.linear2(self.dropout(F.relu(self.linear1(src))))))


class DecoderLayer(nn.Module):
    """nets/models.py:116-156 (the `self_attn` of the reference is constructed and never called)"""

    def __init__(self, d_model, nhead, dim_feedforward=1024, dropout=0.1):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = Linear(d_model, dim_feedforward)
        self.dropout = Dropout(dropout)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.norm1 = LayerNorm(d_model)
        self.norm2 = LayerNorm(d_model)
        self.dropout1 = Dropout(dropout)
        self.dropout2 = Dropout(dropout)

    def forward(self, tgt, memory):
        memory = torch.cat([memory, tgt], dim=0)
        tgt = self.norm1(tgt + self.dropout1(self.multihead_attn(tgt, memory, memory)[0]))
        return self.norm2(tgt + self.dropout2(self.linear2(self.dropout(F.relu(self.linear1(tgt))))))


class Encoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, src):
        for layer in self.layers:
            src = layer(src)
        return self.norm(src) if self.norm else src


class Decoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, tgt, memory):
        for layer in self.layers:
            tgt = layer(tgt, memory)
        return self.norm(tgt) if self.norm else tgt


class RNNEncoder(nn.Module):
    """net_trans.py:44-57"""

    def __init__(self, audio_dim, video_dim, d_model, num_layers):
        super().__init__()
        self.d_model = d_model
        self.audio_rnn = nn.LSTM(audio_dim, int(d_model / 2), num_layers=num_layers, batch_first=True, bidirectional=True,
                                 dropout=0.2)
        self.visual_rnn = nn.LSTM(video_dim, d_model, num_layers=num_layers, batch_first=True, bidirectional=True, dropout=0.2)

    def forward(self, audio_feature, visual_feature):
        return self.audio_rnn(audio_feature)[0], self.visual_rnn(visual_feature)[0]


class InternalTemporalRelationModule(nn.Module):
    """net_trans.py:60-75"""

    def __init__(self, input_dim, d_model, feedforward_dim):
        super().__init__()
        self.encoder_layer = EncoderLayer(d_model=d_model, nhead=4, dim_feedforward=feedforward_dim)
        self.encoder = Encoder(self.encoder_layer, num_layers=2)
        self.affine_matrix = nn.Linear(input_dim, d_model)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, feature):
        return self.encoder(self.affine_matrix(feature))


class CrossModalRelationAttModule(nn.Module):
    """net_trans.py:78-92"""

    def __init__(self, input_dim, d_model, feedforward_dim):
        super().__init__()
        self.decoder_layer = DecoderLayer(d_model=d_model, nhead=4, dim_feedforward=feedforward_dim)
        self.decoder = Decoder(self.decoder_layer, num_layers=1)
        self.affine_matrix = nn.Linear(input_dim, d_model)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, query_feature, memory_feature):
        return self.decoder(self.affine_matrix(query_feature), memory_feature)


class _TemporalGateFn(torch.autograd.Function):
    """(akv, vkv, vq, aq, wa, ba, wv, bv) -> (out_v, out_a, gate): one HIP kernel each way (csrc/temporal.hip)."""

    @staticmethod
    def forward(ctx, lib, gamma, akv, vkv, vq, aq, wa, ba, wv, bv):
        shape = vq.shape
        D = shape[-1]
        ts = [t.contiguous().float() for t in (akv, vkv, vq, aq)]
        R = ts[0].numel() // D
        wa, wv = wa.contiguous().float().reshape(-1), wv.contiguous().float().reshape(-1)
        ba, bv = ba.contiguous().float(), bv.contiguous().float()
        dev = vq.device
        out_v, out_a = torch.empty(R, D, device=dev), torch.empty(R, D, device=dev)
        gate, ga, gv = (torch.empty(R, device=dev) for _ in range(3))
        with _dev_guard(vq):
            lib.temporal_gate_forward(R, D, float(gamma), *[t.data_ptr() for t in ts], wa.data_ptr(), ba.data_ptr(), wv.data_ptr(),
                                      bv.data_ptr(), out_v.data_ptr(), out_a.data_ptr(), gate.data_ptr(), ga.data_ptr(),
                                      gv.data_ptr(), torch.cuda.current_stream(dev).cuda_stream if vq.is_cuda else None)
        ctx.lib, ctx.gamma, ctx.dims = lib, float(gamma), (R, D, shape)
        ctx.save_for_backward(*ts, wa, wv, ga, gv)
        return out_v.view(shape), out_a.view(shape), gate.view(*shape[:-1], 1)

    @staticmethod
    def backward(ctx, dOv, dOa, dg):
        akv, vkv, vq, aq, wa, wv, ga, gv = ctx.saved_tensors
        R, D, shape = ctx.dims
        dev = vq.device
        dOv = (dOv if dOv is not None else torch.zeros(shape, device=dev)).contiguous().float()
        dOa = (dOa if dOa is not None else torch.zeros(shape, device=dev)).contiguous().float()
        dg = dg.contiguous().float() if dg is not None else None
        dakv, dvkv, dvq, daq = (torch.empty(R, D, device=dev) for _ in range(4))
        dwa, dwv = torch.empty(D, device=dev), torch.empty(D, device=dev)
        dba, dbv = torch.empty(1, device=dev), torch.empty(1, device=dev)
        with _dev_guard(vq):
            ctx.lib.temporal_gate_backward(R, D, ctx.gamma, akv.data_ptr(), vkv.data_ptr(), vq.data_ptr(), aq.data_ptr(), wa.data_ptr(),
                                           wv.data_ptr(), ga.data_ptr(), gv.data_ptr(), dOv.data_ptr(), dOa.data_ptr(),
                                           dg.data_ptr() if dg is not None else None, dakv.data_ptr(), dvkv.data_ptr(), dvq.data_ptr(),
                                           daq.data_ptr(), dwa.data_ptr(), dba.data_ptr(), dwv.data_ptr(), dbv.data_ptr(),
                                           torch.cuda.current_stream(dev).cuda_stream if vq.is_cuda else None)
        return (None, None, dakv.view(shape), dvkv.view(shape), dvq.view(shape), daq.view(shape), dwa.view(1, D), dba, dwv.view(1, D), dbv)


class TemporalAttention(nn.Module):
    """net_trans.py:182-251.  ``video_dim`` / ``audio_dim`` are the widths of the pooled backbone features (the reference
    hard-codes Swin-V2-L's 1536 and HTS-AT's 768)."""

    def __init__(self, video_dim: int = 1536, audio_dim: int = 768, lib=None):
        super().__init__()
        self._lib = lib
        self.beta = 0.4
        self.video_input_dim = 512
        self.audio_input_dim = 128
        self.video_fc_dim = 512
        self.audio_fc_dim = 128
        self.d_model = 256
        self.v_fc = nn.Linear(video_dim, self.video_fc_dim)
        self.a_fc = nn.Linear(audio_dim, self.audio_fc_dim)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(0.2)
        self.video_encoder = InternalTemporalRelationModule(input_dim=self.video_input_dim, d_model=self.d_model, feedforward_dim=1024)
        self.video_decoder = CrossModalRelationAttModule(input_dim=self.video_input_dim, d_model=self.d_model, feedforward_dim=1024)
        self.audio_encoder = InternalTemporalRelationModule(input_dim=self.d_model, d_model=self.d_model, feedforward_dim=1024)
        self.audio_decoder = CrossModalRelationAttModule(input_dim=self.d_model, d_model=self.d_model, feedforward_dim=1024)
        self.audio_visual_rnn_layer = RNNEncoder(audio_dim=self.audio_input_dim, video_dim=self.video_input_dim, d_model=self.d_model,
                                                 num_layers=1)
        self.audio_gated = nn.Sequential(nn.Linear(self.d_model, 1), nn.Sigmoid())
        self.video_gated = nn.Sequential(nn.Linear(self.d_model, 1), nn.Sigmoid())
        self.alpha = 0.1
        self.gamma = 0.1

    def _project(self, vis: torch.Tensor, aud: torch.Tensor):
        """per-modality input projections (net_trans.py:216-221): audio is a plain Linear, video Linear -> ReLU -> Dropout"""
        return self.dropout(self.relu(self.v_fc(vis))), self.a_fc(aud)

    def _time_major(self, vis: torch.Tensor, aud: torch.Tensor):
        """the joint bi-LSTM over the T frames, then [B, T, d] -> [T, B, d] for the attention blocks (:222-226)"""
        aud_seq, vis_seq = self.audio_visual_rnn_layer(aud, vis)
        return vis_seq.transpose(0, 1).contiguous(), aud_seq.transpose(0, 1).contiguous()

    def forward(self, visual_feature, audio_feature):
        """visual_feature [B, T, video_dim], audio_feature [B, T, audio_dim] ->
        (video_query_output [T, B, 256], audio_query_output [T, B, 256], audio_visual_gate [T, B, 1]).

        Each modality is self-encoded over time (memory) and cross-decoded against the OTHER modality's memory (queries):
        net_trans.py:228-238; the gates on the two memories and their application (:240-249) are one HIP kernel each way."""
        vis, aud = self._project(visual_feature, audio_feature)
        vis_t, aud_t = self._time_major(vis, aud)
        # call order video-enc, audio-dec, audio-enc, video-dec: in training mode the dropouts inside draw from the RNG
        # stream in this order, and a seeded run has to match the reference's draw for draw
        memory, queries = {}, {}
        memory["video"] = self.video_encoder(vis_t)
        queries["audio"] = self.audio_decoder(aud_t, memory["video"])
        memory["audio"] = self.audio_encoder(aud_t)
        queries["video"] = self.video_decoder(vis_t, memory["audio"])
        if not queries["video"].is_cuda and self._lib is None:
            raise RuntimeError("dg-sct_amd.TemporalAttention applies its gates with a HIP kernel; there is no CPU path "
                               "(move the module and its inputs to a ROCm device)")
        lib = self._lib or _lib.default_lib()
        gate_a, gate_v = self.audio_gated[0], self.video_gated[0]
        return _TemporalGateFn.apply(lib, self.gamma, memory["audio"], memory["video"], queries["video"], queries["audio"],
                                     gate_a.weight, gate_a.bias, gate_v.weight, gate_v.bias)


# ---- the other two copies of the class -----------------------------------------------------------
class _FrameScaleFn(torch.autograd.Function):
    """y[r] = x[r] * (1 + gamma * g[r]) for per-frame gates g [rows]: one HIP kernel each way (dgsct_frame_scale_*)."""

    @staticmethod
    def forward(ctx, lib, gamma, x, g):
        xc = x.contiguous()
        gc = g.reshape(-1).contiguous().float()
        rows = gc.numel()
        inner = xc.numel() // rows if rows else 0            # (an empty gate: nothing to scale, the library returns at once)
        y = torch.empty_like(xc)
        dt = _lib.BF16 if xc.dtype == torch.bfloat16 else _lib.F32
        with _dev_guard(xc):
            lib.frame_scale_forward(dt, rows, inner, float(gamma), xc.data_ptr(), gc.data_ptr(), y.data_ptr(),
                                    torch.cuda.current_stream(xc.device).cuda_stream if xc.is_cuda else None)
        ctx.lib, ctx.gamma, ctx.gshape, ctx.gdtype = lib, float(gamma), g.shape, g.dtype
        ctx.save_for_backward(xc, gc)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, gc = ctx.saved_tensors
        dyc = dy.contiguous()
        rows = gc.numel()
        inner = xc.numel() // rows if rows else 0
        dx = torch.empty_like(xc) if ctx.needs_input_grad[2] else None
        dg = torch.empty(rows, dtype=torch.float32, device=xc.device) if ctx.needs_input_grad[3] else None
        dt = _lib.BF16 if xc.dtype == torch.bfloat16 else _lib.F32
        with _dev_guard(xc):
            ctx.lib.frame_scale_backward(dt, rows, inner, ctx.gamma, xc.data_ptr(), gc.data_ptr(), dyc.data_ptr(),
                                         dx.data_ptr() if dx is not None else None, dg.data_ptr() if dg is not None else None,
                                         torch.cuda.current_stream(xc.device).cuda_stream if xc.is_cuda else None)
        return None, None, dx, (dg.view(ctx.gshape).to(ctx.gdtype) if dg is not None else None)


def frame_scale(x: torch.Tensor, g: torch.Tensor, gamma: float, lib=None) -> torch.Tensor:
    """``x + g * x * gamma`` with ONE gate per leading-dimension frame of ``x`` (g broadcast over everything else)."""
    if not x.is_cuda and lib is None:
        raise RuntimeError("dg-sct_amd.frame_scale runs on MI355X through libdgsct.so; there is no CPU path")
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("dg-sct_amd.frame_scale supports float32 and bfloat16")
    return _FrameScaleFn.apply(lib or _lib.default_lib(), gamma, x, g)


class _RNNEncoderFull(nn.Module):
    """AVVP/nets/mgn.py:39-52: unlike the AVE / AVS copies, BOTH bi-LSTMs have d_model hidden units."""

    def __init__(self, audio_dim, video_dim, d_model, num_layers):
        super().__init__()
        self.d_model = d_model
        self.audio_rnn = nn.LSTM(audio_dim, d_model, num_layers=num_layers, batch_first=True, bidirectional=True, dropout=0.2)
        self.visual_rnn = nn.LSTM(video_dim, d_model, num_layers=num_layers, batch_first=True, bidirectional=True, dropout=0.2)

    def forward(self, audio_feature, visual_feature):
        return self.audio_rnn(audio_feature)[0], self.visual_rnn(visual_feature)[0]


class TemporalAttentionAVVP(nn.Module):
    """AVVP copy, ``DG-SCT/AVVP/nets/mgn.py:107-159``: 128-wide features both ways, d_model 64, no input FCs, no decoders; the
    gates (computed from the OTHER modality's encoder output) scale the INPUT features: ``x + gate * x * gamma``, gamma 0.05."""

    def __init__(self, lib=None):
        super().__init__()
        self._lib = lib
        self.beta = 0.4
        self.video_input_dim = 128
        self.audio_input_dim = 128
        self.video_fc_dim = 128
        self.audio_fc_dim = 128
        self.d_model = 64
        self.video_encoder = InternalTemporalRelationModule(input_dim=self.video_input_dim, d_model=self.d_model, feedforward_dim=1024)
        self.audio_encoder = InternalTemporalRelationModule(input_dim=self.audio_input_dim, d_model=self.d_model, feedforward_dim=1024)
        self.audio_visual_rnn_layer = _RNNEncoderFull(audio_dim=self.audio_input_dim, video_dim=self.video_input_dim,
                                                      d_model=self.d_model, num_layers=1)
        self.audio_gated = nn.Sequential(nn.Linear(self.d_model, 1), nn.Sigmoid())
        self.video_gated = nn.Sequential(nn.Linear(self.d_model, 1), nn.Sigmoid())
        self.alpha = 0.05
        self.gamma = 0.05

    def forward(self, visual_feature, audio_feature):
        """[B, 10, 128] each -> (video_query_output, audio_query_output), same shapes"""
        a_rnn, v_rnn = self.audio_visual_rnn_layer(audio_feature, visual_feature)
        video_kv = self.video_encoder(v_rnn.transpose(1, 0).contiguous())
        audio_kv = self.audio_encoder(a_rnn.transpose(1, 0).contiguous())
        audio_gate = self.audio_gated(audio_kv).transpose(1, 0)             # [B, T, 1]
        video_gate = self.video_gated(video_kv).transpose(1, 0)
        return (frame_scale(visual_feature, audio_gate, self.gamma, self._lib),
                frame_scale(audio_feature, video_gate, self.gamma, self._lib))


class TemporalAttentionAVS(nn.Module):
    """AVS copy, ``avs_s4/model/PVT_AVSModel.py:447-582``: one set of blocks per decoder scale (``ModuleList`` x 4), the visual
    inputs are the four [B*5, 256, H, W] maps (average-pooled to per-frame vectors for the gates), the audio gate of a scale
    scales that scale's whole MAP per frame, the four video gates are averaged and scale the audio feature; gamma 0.05.
    (`video_decoder` / `audio_decoder` outputs are computed by the reference and never used; they are constructed here for
    the checkpoint format and skipped in forward -- they receive no gradient in the reference either.)"""

    def __init__(self, lib=None):
        super().__init__()
        self._lib = lib
        self.gamma = 0.05
        self.video_input_dim = 256
        self.audio_input_dim = 128
        self.video_fc_dim = 256
        self.audio_fc_dim = 128
        self.d_model = 256
        self.v_fc = nn.ModuleList([nn.Linear(self.video_input_dim, self.video_fc_dim) for _ in range(4)])
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(0.2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.video_encoder = nn.ModuleList([InternalTemporalRelationModule(input_dim=512, d_model=self.d_model, feedforward_dim=1024) for _ in range(4)])
        self.video_decoder = nn.ModuleList([CrossModalRelationAttModule(input_dim=512, d_model=self.d_model, feedforward_dim=1024) for _ in range(4)])
        self.audio_encoder = nn.ModuleList([InternalTemporalRelationModule(input_dim=self.d_model, d_model=self.d_model, feedforward_dim=1024) for _ in range(4)])
        self.audio_decoder = nn.ModuleList([CrossModalRelationAttModule(input_dim=self.d_model, d_model=self.d_model, feedforward_dim=1024) for _ in range(4)])
        self.audio_visual_rnn_layer = nn.ModuleList([RNNEncoder(audio_dim=self.audio_input_dim, video_dim=self.video_input_dim,
                                                                d_model=self.d_model, num_layers=1) for _ in range(4)])
        self.audio_gated = nn.ModuleList([nn.Sequential(nn.Linear(self.d_model, 1), nn.Sigmoid()) for _ in range(4)])
        self.video_gated = nn.ModuleList([nn.Sequential(nn.Linear(self.d_model, 1), nn.Sigmoid()) for _ in range(4)])

    def forward(self, visual_feature_list, audio_feature):
        """visual_feature_list: four [B*5, 256, H_i, W_i] maps; audio_feature [B, 5, 128] ->
        ([x1, x2, x3, x4] gated maps, audio_feature [B*5, 128] gated)"""
        bs = audio_feature.size(0)
        T = 5
        audio_rnn_input = audio_feature
        audio_flat = audio_feature.reshape(-1, audio_feature.size(-1))
        outs, vgates = [], []
        for i, x in enumerate(visual_feature_list):
            xv = self.dropout(self.relu(self.v_fc[i](self.avgpool(x).reshape(bs, T, -1))))
            a_rnn, v_rnn = self.audio_visual_rnn_layer[i](audio_rnn_input, xv)
            a_in = a_rnn.transpose(1, 0).contiguous()                      # [5, B, 256]
            v_in = v_rnn.transpose(1, 0).contiguous()                      # [5, B, 512]
            video_kv = self.video_encoder[i](v_in)
            audio_kv = self.audio_encoder[i](a_in)
            audio_gate = self.audio_gated[i](audio_kv).transpose(1, 0).reshape(bs * T)
            vgates.append(self.video_gated[i](video_kv).transpose(1, 0).reshape(bs * T, 1))
            outs.append(frame_scale(x, audio_gate, self.gamma, self._lib))
        video_gate = (vgates[0] + vgates[1] + vgates[2] + vgates[3]) / 4
        return outs, frame_scale(audio_flat, video_gate, self.gamma, self._lib)
