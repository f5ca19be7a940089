"""CPU oracle of the post-backbone ``TemporalAttention`` (reference DG-SCT/AVE/nets/net_trans.py:182-251).  TEST INFRASTRUCTURE ONLY
(same rules as oracle/dgsct_oracle.py: tests/ and __graft_entry__.smoke() may import it, the product never does).

A functional restatement driven by a reference-named ``state_dict``: stock ATen LSTM / multi-head attention / LayerNorm /
Linear calls in the reference's order (eval mode: the Dropouts are identities), with the gate application written out as
plain arithmetic -- the part the product replaces by a HIP kernel:
    ga = sigmoid(akv @ wa^T + ba), gv = sigmoid(vkv @ wv^T + bv)                      net_trans.py:240-241
    gate = ga * gv; vq += ga * vq * gamma; aq += gv * aq * gamma                      :243-246
Parity status: PINNED by oracle/make_golden_temporal.py against the imported reference class (tests/golden/temporal.pt).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

GAMMA = 0.1     # TemporalAttention.gamma (net_trans.py:213)


def _mha(sd, pre, q, k, v, nhead=4):
    return F.multi_head_attention_forward(q, k, v, q.shape[-1], nhead, sd[pre + "in_proj_weight"], sd[pre + "in_proj_bias"], None, None,
                                          False, 0.0, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"], training=False,
                                          need_weights=False)[0]


def _ln(sd, pre, x):
    return F.layer_norm(x, (x.shape[-1],), sd[pre + "weight"], sd[pre + "bias"], 1e-5)


def _ff(sd, pre, x):
    return F.linear(F.relu(F.linear(x, sd[pre + "linear1.weight"], sd[pre + "linear1.bias"])), sd[pre + "linear2.weight"],
                    sd[pre + "linear2.bias"])


def _enc_layer(sd, pre, src):                                    # nets/models.py:100-113
    src = _ln(sd, pre + "norm1.", src + _mha(sd, pre + "self_attn.", src, src, src))
    return _ln(sd, pre + "norm2.", src + _ff(sd, pre, src))


def _dec_layer(sd, pre, tgt, memory):                            # nets/models.py:143-156
    memory = torch.cat([memory, tgt], dim=0)
    tgt = _ln(sd, pre + "norm1.", tgt + _mha(sd, pre + "multihead_attn.", tgt, memory, memory))
    return _ln(sd, pre + "norm2.", tgt + _ff(sd, pre, tgt))


def _bilstm(sd, pre, x, hidden):                                 # nn.LSTM(batch_first=True, bidirectional=True, num_layers=1)
    flat = [sd[pre + n + s] for s in ("_l0", "_l0_reverse") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    h0 = x.new_zeros(2, x.shape[0], hidden)
    return torch._VF.lstm(x, (h0, h0.clone()), flat, True, 1, 0.0, False, True, True)[0]


def forward(sd: Dict[str, torch.Tensor], visual_feature, audio_feature, gamma: float = GAMMA):
    """visual_feature [B, T, Dv], audio_feature [B, T, Da] -> (video_query_output, audio_query_output [T, B, 256], gate [T, B, 1])"""
    a = F.linear(audio_feature, sd["a_fc.weight"], sd["a_fc.bias"])
    v = F.relu(F.linear(visual_feature, sd["v_fc.weight"], sd["v_fc.bias"]))
    a_rnn = _bilstm(sd, "audio_visual_rnn_layer.audio_rnn.", a, 128).transpose(1, 0).contiguous()       # [T, B, 256]
    v_rnn = _bilstm(sd, "audio_visual_rnn_layer.visual_rnn.", v, 256).transpose(1, 0).contiguous()      # [T, B, 512]
    vkv = F.linear(v_rnn, sd["video_encoder.affine_matrix.weight"], sd["video_encoder.affine_matrix.bias"])
    for i in range(2):
        vkv = _enc_layer(sd, f"video_encoder.encoder.layers.{i}.", vkv)
    aq = _dec_layer(sd, "audio_decoder.decoder.layers.0.", F.linear(a_rnn, sd["audio_decoder.affine_matrix.weight"],
                                                                    sd["audio_decoder.affine_matrix.bias"]), vkv)
    akv = F.linear(a_rnn, sd["audio_encoder.affine_matrix.weight"], sd["audio_encoder.affine_matrix.bias"])
    for i in range(2):
        akv = _enc_layer(sd, f"audio_encoder.encoder.layers.{i}.", akv)
    vq = _dec_layer(sd, "video_decoder.decoder.layers.0.", F.linear(v_rnn, sd["video_decoder.affine_matrix.weight"],
                                                                    sd["video_decoder.affine_matrix.bias"]), akv)
    ga = torch.sigmoid(F.linear(akv, sd["audio_gated.0.weight"], sd["audio_gated.0.bias"]))
    gv = torch.sigmoid(F.linear(vkv, sd["video_gated.0.weight"], sd["video_gated.0.bias"]))
    return vq + ga * vq * gamma, aq + gv * aq * gamma, ga * gv
