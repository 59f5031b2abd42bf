"""CPU ORACLE for the AMID training hot path -- TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``amid_amd/`` imports, calls or falls back to it; the product
path raises when ``libamid_hip.so`` is missing.

What it is: a plain-tensor (torch fp32, CPU) restatement of the reference's
algorithm for the path ``model_seq.py`` / ``train_sr.py`` (item-embedding
gathers -> attention sequence encoder -> MLP scorer -> masked BCE -> backward ->
dense Adam), written op by op from the reference call sites cited on every
function (all ``file:line`` are into the read-only reference checkout).  The
arithmetic that the reference delegates to PyTorch (``nn.MultiheadAttention``,
``nn.LayerNorm``, ``nn.Conv1d(k=1)``, ``nn.BCELoss``, ``torch.optim.Adam``; no
version pinned by the reference, torch 2.10.0 used here) is restated with
explicit matmul / exp / sqrt ops so that every intermediate the HIP kernels
produce has a named counterpart.

Parity status: PINNED.  The reference ships no tests or golden vectors
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, generated in the build container by ``tests/golden/make_golden.py``
(imports the reference read-only with a ``.cuda()`` no-op shim) and committed
as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every
function below against them.

Parameters travel as a ``dict[str, torch.Tensor]`` keyed by the reference's own
``state_dict`` names.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

Params = Dict[str, torch.Tensor]
Masks = Optional[Dict[str, torch.Tensor]]

SASREC_HEADS = 8          # model_seq.py:348-350 (nn.MultiheadAttention(D, 8, 0.5))
SASREC_DROPOUT = 0.5      # model_seq.py:335,350,356
SASREC_LN_EPS = 1e-8      # model_seq.py:342,345,353
BERT_HIDDEN = 128         # model_seq.py:264-267 (hard-coded)
BERT_HEADS = 4
BERT_FF = 512
BERT_DROPOUT = 0.1
BERT_LN_EPS = 1e-6        # model_seq.py:118


# --------------------------------------------------------------------------
# counter-based RNG shared with the HIP kernels (our spec, not the reference's:
# GPU dropout cannot reproduce CPU bernoulli_, SURVEY.md section 7 "Dropout
# parity").  Philox4x32-10; key = seed, counter = (idx_lo, idx_hi, site, step).
# One call serves 8 elements (16-bit halves), see philox_keep_flat.
# --------------------------------------------------------------------------
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)


def philox4x32(ctr: np.ndarray, key: Tuple[int, int]) -> np.ndarray:
    """ctr: uint32 [n,4] -> uint32 [n,4]; 10 rounds."""
    c = ctr.astype(np.uint32).copy()
    k0 = np.uint32(key[0])
    k1 = np.uint32(key[1])
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c[:, 0].astype(np.uint64) * _PHILOX_M0
            p1 = c[:, 2].astype(np.uint64) * _PHILOX_M1
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = p0.astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = p1.astype(np.uint32)
            n0 = hi1 ^ c[:, 1] ^ k0
            n1 = lo1
            n2 = hi0 ^ c[:, 3] ^ k1
            n3 = lo0
            c = np.stack([n0, n1, n2, n3], axis=1)
            k0 = np.uint32(k0 + _PHILOX_W0)
            k1 = np.uint32(k1 + _PHILOX_W1)
    return c


def drop_bits(p: float) -> Tuple[int, int]:
    """(bits per decision b, threshold): smallest b in {1,2,4,8} with p * 2**b an integer, else 16 bits with the
    threshold rounded.  keep <=> field >= thr ; P(keep) = 1 - thr / 2**b.  (amid_amd/csrc/rng.h: drop_spec)"""
    for b in (1, 2, 4, 8):
        t = np.float32(p) * np.float32(1 << b)
        if float(t) == float(int(t)):
            return b, int(t)
    return 16, min(int(np.float32(p) * np.float32(65536.0) + np.float32(0.5)), 0xFFFF)


def drop_per_call(p: float) -> int:
    return 128 // drop_bits(p)[0]


def philox_keep_flat(n_elem: int, seed: int, site: int, step: int, p: float) -> np.ndarray:
    """Keep mask (float32 0/1) for linear element indices [0, n_elem).  One Philox call decides 128 / b
    elements: element e uses field (e % (128 / b)) of call (e // (128 / b)), fields packed LSB-first in the
    four 32-bit words of the call."""
    b, thr = drop_bits(p)
    per = 128 // b
    n_call = (n_elem + per - 1) // per
    idx = np.arange(n_call, dtype=np.uint64)
    ctr = np.stack([
        (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32),
        (idx >> np.uint64(32)).astype(np.uint32),
        np.full(n_call, site, dtype=np.uint32),
        np.full(n_call, step & 0xFFFFFFFF, dtype=np.uint32),
    ], axis=1)
    r = philox4x32(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))          # [n_call, 4]
    shifts = (np.arange(32 // b, dtype=np.uint32) * np.uint32(b))
    fields = (r[:, :, None] >> shifts[None, None, :]) & np.uint32((1 << b) - 1)   # [n_call, 4, 32/b]
    return (fields.reshape(-1)[:n_elem] >= np.uint32(thr)).astype(np.float32)


# mask-site numbering shared with amid_amd/csrc/rng.h
SITE_EMB = 0
SITE_ATTN = 1
SITE_FFN1 = 2
SITE_FFN2 = 3
SITE_SUB_IN = 4     # BERT4Rec input_sublayer dropout
SITE_SUB_OUT = 5    # BERT4Rec output_sublayer dropout
SITE_BLOCK = 6      # BERT4Rec TransformerBlock.dropout


def site_id(domain: int, layer: int, kind: int) -> int:
    """domain in {0,1}, layer in {0,1}, kind one of SITE_*."""
    return (domain * 2 + layer) * 8 + kind


def attn_row_stride(T: int, p: float) -> int:
    """Attention keep masks are indexed [b, h, i, j] with the row padded to a whole number of Philox calls."""
    per = drop_per_call(p)
    return (T + per - 1) // per * per


def philox_masks_sasrec(B: int, T: int, D: int, seed: int, step: int,
                        H: int = SASREC_HEADS, p: float = SASREC_DROPOUT) -> Dict[str, torch.Tensor]:
    """All SASRec train-mode keep masks exactly as the HIP kernels derive them."""
    out: Dict[str, torch.Tensor] = {}
    TP = attn_row_stride(T, p)
    for d in (0, 1):
        pre = f"sac{d + 1}"
        out[f"{pre}.emb"] = torch.from_numpy(
            philox_keep_flat(B * T * D, seed, site_id(d, 0, SITE_EMB), step, p).reshape(B, T, D))
        for l in (0, 1):
            a = philox_keep_flat(B * H * T * TP, seed, site_id(d, l, SITE_ATTN), step, p)
            out[f"{pre}.attn{l}"] = torch.from_numpy(a.reshape(B, H, T, TP)[..., :T].copy())
            out[f"{pre}.ffn1_{l}"] = torch.from_numpy(
                philox_keep_flat(B * T * D, seed, site_id(d, l, SITE_FFN1), step, p).reshape(B, T, D))
            out[f"{pre}.ffn2_{l}"] = torch.from_numpy(
                philox_keep_flat(B * T * D, seed, site_id(d, l, SITE_FFN2), step, p).reshape(B, T, D))
    return out


def philox_masks_bert4rec(B: int, T: int, seed: int, step: int, p: float = BERT_DROPOUT) -> Dict[str, torch.Tensor]:
    out: Dict[str, torch.Tensor] = {}
    D, H, F = BERT_HIDDEN, BERT_HEADS, BERT_FF
    TP = attn_row_stride(T, p)
    for d in (0, 1):
        pre = f"transform{d + 1}"
        for l in (0, 1):
            a = philox_keep_flat(B * H * T * TP, seed, site_id(d, l, SITE_ATTN), step, p)
            out[f"{pre}.{l}.attn"] = torch.from_numpy(a.reshape(B, H, T, TP)[..., :T].copy())
            out[f"{pre}.{l}.sub_in"] = torch.from_numpy(
                philox_keep_flat(B * T * D, seed, site_id(d, l, SITE_SUB_IN), step, p).reshape(B, T, D))
            out[f"{pre}.{l}.ffn"] = torch.from_numpy(
                philox_keep_flat(B * T * F, seed, site_id(d, l, SITE_FFN1), step, p).reshape(B, T, F))
            out[f"{pre}.{l}.sub_out"] = torch.from_numpy(
                philox_keep_flat(B * T * D, seed, site_id(d, l, SITE_SUB_OUT), step, p).reshape(B, T, D))
            out[f"{pre}.{l}.block"] = torch.from_numpy(
                philox_keep_flat(B * T * D, seed, site_id(d, l, SITE_BLOCK), step, p).reshape(B, T, D))
    return out


def _drop(x: torch.Tensor, masks: Masks, name: str, p: float) -> torch.Tensor:
    """Inverted dropout with an explicit keep mask (torch F.dropout semantics:
    kept values scaled by 1/(1-p)).  masks=None means eval mode."""
    if masks is None:
        return x
    return x * masks[name].to(x.dtype) * (1.0 / (1.0 - p))


# --------------------------------------------------------------------------
# a1  embItemLayerEnhance.forward            model_seq.py:27-29
# --------------------------------------------------------------------------
def gather_rows(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """W[idx]; no padding_idx, pad id is an ordinary row (model_seq.py:25)."""
    return table.index_select(0, idx.reshape(-1).long()).reshape(*idx.shape, table.shape[1])


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    """torch.nn.LayerNorm: biased variance, eps inside the sqrt."""
    mean = x.mean(-1, keepdim=True)
    xc = x - mean
    var = (xc * xc).mean(-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * w + b


# --------------------------------------------------------------------------
# a2/a3/a4  Log2feats.forward                 model_seq.py:359-387
#           nn.MultiheadAttention as called at :374 (packed in-proj, q from the
#           normed Q, k/v from the un-normed seqs, need_weights=True explicit
#           softmax path, dropout on the probabilities)
#           PointWiseFeedForward.forward      model_seq.py:322-326
# --------------------------------------------------------------------------
def sasrec_encoder(x: torch.Tensor, P: Params, pre: str, masks: Masks = None,
                   heads: int = SASREC_HEADS, taps: Optional[dict] = None, relu_keep: Optional[dict] = None) -> torch.Tensor:
    """x: gathered item rows [B,T,D] (not modified).  Returns log_feats [B,T,D].
    relu_keep (tests only): {f"{pre}.relu{l}": [B,T,D] of 1 / 0 / -1} replaces relu(h) by h * keep -- the decisions of the
    implementation under test; -1 = the oracle's own decision -- so that gradients can be compared to rounding even when some
    pre-activation sits within rounding of the kink (a flipped decision changes a gradient by O(1), not O(eps)).  taps[f"relu_flip{l}"] then records the largest |h| among the
    elements whose forced decision differs from the oracle's own: the caller asserts it is rounding-sized."""
    B, T, D = x.shape
    hd = D // heads
    p = SASREC_DROPOUT
    x = x + P[f"{pre}.pos_emb.weight"][:T].unsqueeze(0)                 # :361-362
    tm = (x == 0)                                                        # :365 (on the pos-added, pre-dropout values)
    x = _drop(x, masks, f"{pre}.emb", p)                                 # :363
    keep = (~tm).to(x.dtype)
    x = x * keep                                                         # :366
    causal = torch.triu(torch.ones(T, T, dtype=torch.bool), diagonal=1)  # :369  True = masked
    if taps is not None:
        taps["x0"] = x
    for l in range(2):
        Wi = P[f"{pre}.attention_layers.{l}.in_proj_weight"]
        bi = P[f"{pre}.attention_layers.{l}.in_proj_bias"]
        Wo = P[f"{pre}.attention_layers.{l}.out_proj.weight"]
        bo = P[f"{pre}.attention_layers.{l}.out_proj.bias"]
        Q = layer_norm(x, P[f"{pre}.attention_layernorms.{l}.weight"],
                       P[f"{pre}.attention_layernorms.{l}.bias"], SASREC_LN_EPS)     # :373
        q = Q @ Wi[0:D].t() + bi[0:D]
        k = x @ Wi[D:2 * D].t() + bi[D:2 * D]
        v = x @ Wi[2 * D:].t() + bi[2 * D:]
        q = q.reshape(B, T, heads, hd).permute(0, 2, 1, 3)
        k = k.reshape(B, T, heads, hd).permute(0, 2, 1, 3)
        v = v.reshape(B, T, heads, hd).permute(0, 2, 1, 3)
        S = (q * math.sqrt(1.0 / hd)) @ k.transpose(-1, -2)             # q scaled before q.k^T
        S = S.masked_fill(causal, float("-inf"))
        S = S - S.max(-1, keepdim=True).values
        E = torch.exp(S)
        A = E / E.sum(-1, keepdim=True)
        A = _drop(A, masks, f"{pre}.attn{l}", p)
        o = (A @ v).permute(0, 2, 1, 3).reshape(B, T, D)
        o = o @ Wo.t() + bo
        x = Q + o                                                        # :378 residual on the NORMED query
        y = layer_norm(x, P[f"{pre}.forward_layernorms.{l}.weight"],
                       P[f"{pre}.forward_layernorms.{l}.bias"], SASREC_LN_EPS)       # :381
        C1 = P[f"{pre}.forward_layers.{l}.conv1.weight"][:, :, 0]
        c1 = P[f"{pre}.forward_layers.{l}.conv1.bias"]
        C2 = P[f"{pre}.forward_layers.{l}.conv2.weight"][:, :, 0]
        c2 = P[f"{pre}.forward_layers.{l}.conv2.bias"]
        h = _drop(y @ C1.t() + c1, masks, f"{pre}.ffn1_{l}", p)          # :323 conv1 -> dropout1 -> relu
        if taps is not None:     # distance of the closest live pre-activation from the relu kink (test robustness)
            live = h[h != 0]
            taps[f"relu_margin{l}"] = float(live.abs().min()) if live.numel() else float("inf")
        rk = None if relu_keep is None else relu_keep.get(f"{pre}.relu{l}")
        if rk is None:
            h = torch.relu(h)
        else:
            own = h.detach() > 0
            dec = torch.where(rk < 0, own, rk > 0)
            if taps is not None:
                flipped = dec != own
                taps[f"relu_flip{l}"] = float(h.detach()[flipped].abs().max()) if bool(flipped.any()) else 0.0
                taps[f"relu_nflip{l}"] = int(flipped.sum())
            h = h * dec.to(h.dtype)
        z = _drop(h @ C2.t() + c2, masks, f"{pre}.ffn2_{l}", p)          # conv2 -> dropout2
        x = (z + y) * keep                                               # :325 (+= inputs, the LN output), :383
        if taps is not None:
            taps[f"x{l + 1}"] = x
    return layer_norm(x, P[f"{pre}.last_layernorm.weight"], P[f"{pre}.last_layernorm.bias"], SASREC_LN_EPS)  # :385


# --------------------------------------------------------------------------
# a6  predictModule.forward                   model_seq.py:40-54
# --------------------------------------------------------------------------
def predict_module(u1: torch.Tensor, u2: torch.Tensor, items: torch.Tensor, P: Params,
                   pre: str = "predictModule") -> Tuple[torch.Tensor, torch.Tensor]:
    """u*: [B,D]; items: [B,N,D] -> two [B,N] sigmoid outputs (the reference's
    trailing .squeeze() is applied by the caller-facing wrappers)."""
    W1, b1 = P[f"{pre}.fc.0.weight"], P[f"{pre}.fc.0.bias"]
    W2, b2 = P[f"{pre}.fc.2.weight"], P[f"{pre}.fc.2.bias"]
    outs = []
    for u in (u1, u2):
        cat = torch.cat((u.unsqueeze(1).expand_as(items), items), -1)
        h = torch.relu(cat @ W1.t() + b1)
        z = h @ W2.t() + b2
        outs.append((1.0 / (1.0 + torch.exp(-z))).squeeze(-1))
    return outs[0], outs[1]


# --------------------------------------------------------------------------
# a5  SASRec.forward                          model_seq.py:416-443
# --------------------------------------------------------------------------
def inter_comp(seq_self: torch.Tensor, seq_other: torch.Tensor, P: Params, pre: str, threshold: float,
               taps: Optional[dict] = None) -> torch.Tensor:
    """InterComp.forward model_seq.py:483-497 (next-1), compute-once form: the reference repeats seq_other
    `bs` times along a new leading axis (:487) and every slice of that axis computes the same thing, so the appended block
    is ONE [T, D] token group shared by every row.  InnerComp.forward (:459-472) is inter_comp(seq, seq)."""
    s = torch.matmul(seq_self, seq_other.transpose(1, 2)).amax(dim=(1, 2))          # [B]  :488-489 max over both time axes
    sm = torch.softmax(s, dim=0)                                                    # :490 softmax over the BATCH
    gate = (sm > threshold).to(seq_self.dtype)                                      # :491 getBinaryTensor (no gradient)
    h = (seq_other * gate[:, None, None]) @ P[f"{pre}.trans_nn.weight"].t() + P[f"{pre}.trans_nn.bias"]      # :492-493
    grp = (h * P[f"{pre}.trans_bs.weight"][0][:, None, None]).sum(0) + P[f"{pre}.trans_bs.bias"]             # :494 Linear(bs, 1) over the batch
    if taps is not None:
        taps[pre] = dict(s=s.detach(), softmax=sm.detach(), gate=gate.detach(), margin=float((sm.detach() - threshold).abs().min()))
    return torch.cat((seq_self, grp.unsqueeze(0).expand(seq_self.shape[0], -1, -1)), 1)                     # :495


def inner_comp(seq: torch.Tensor, P: Params, pre: str, threshold: float, taps: Optional[dict] = None) -> torch.Tensor:
    return inter_comp(seq, seq, P, pre, threshold, taps)


def sasrec_forward(P: Params, i_node: torch.Tensor, neg_samples: torch.Tensor, seq_d1: torch.Tensor,
                   seq_d2: torch.Tensor, masks: Masks = None, taps: Optional[dict] = None, isItC: bool = False,
                   threshold2: float = 0.5, isDR: bool = False, isInC: bool = False, threshold1: float = 0.5,
                   relu_keep: Optional[dict] = None) -> Tuple[torch.Tensor, ...]:
    E = P["item_emb_layer.emb_item.weight"]
    i_feat = gather_rows(E, i_node).unsqueeze(1)                         # :418
    neg_feat = gather_rows(E, neg_samples)                               # :419
    e1, e2 = gather_rows(E, seq_d1), gather_rows(E, seq_d2)              # :420-421
    if isInC:                                                            # :422-424 on the raw gathered rows; the encoders then see 2T tokens
        e1, e2 = inner_comp(e1, P, "inc_d1", threshold1, taps), inner_comp(e2, P, "inc_d2", threshold1, taps)
    f1 = sasrec_encoder(e1, P, "sac1", masks, taps=None if taps is None else taps.setdefault("sac1", {}), relu_keep=relu_keep)
    f2 = sasrec_encoder(e2, P, "sac2", masks, taps=None if taps is None else taps.setdefault("sac2", {}), relu_keep=relu_keep)
    if isItC:                                                            # :426-431 (after the encoders, both from the un-mixed features)
        f1, f2 = inter_comp(f1, f2, P, "itc_d1", threshold2, taps), inter_comp(f2, f1, P, "itc_d2", threshold2, taps)
    u1 = f1.mean(1)                                                      # :432 mean over ALL T (pads included; 2T with isItC)
    u2 = f2.mean(1)                                                      # :434
    items = torch.cat((i_feat, neg_feat), 1)                             # :435
    if taps is not None:
        taps.update(u1=u1, u2=u2, items=items, f1=f1, f2=f2)
    if isDR:                                                             # :436-440 (next-4): three heads on the same (u, items)
        return (predict_module(u1, u2, items, P) + predict_module(u1, u2, items, P, "predict_ips")
                + predict_module(u1, u2, items, P, "predict_gfunc"))
    return predict_module(u1, u2, items, P)                              # :442


# --------------------------------------------------------------------------
# a7  BERT4Rec stack                          model_seq.py:115-309
# --------------------------------------------------------------------------
def bert_layer_norm(x: torch.Tensor, a: torch.Tensor, b: torch.Tensor, eps: float = BERT_LN_EPS) -> torch.Tensor:
    """model_seq.py:124-127: unbiased std, eps added to the std."""
    mean = x.mean(-1, keepdim=True)
    xc = x - mean
    std = torch.sqrt((xc * xc).sum(-1, keepdim=True) / (x.shape[-1] - 1))
    return a * xc / (std + eps) + b


def gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    """model_seq.py:204."""
    return 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * torch.pow(x, 3))))


def bert_block(x: torch.Tensor, key_keep: torch.Tensor, P: Params, pre: str, masks: Masks = None) -> torch.Tensor:
    """TransformerBlock.forward model_seq.py:242-245.  key_keep: bool [B,T]."""
    B, T, D = x.shape
    H, dk, p = BERT_HEADS, BERT_HIDDEN // BERT_HEADS, BERT_DROPOUT
    y = bert_layer_norm(x, P[f"{pre}.input_sublayer.norm.a_2"], P[f"{pre}.input_sublayer.norm.b_2"])
    qkv = []
    for j in range(3):                                                   # :187-188
        w = P[f"{pre}.attention.linear_layers.{j}.weight"]
        bb = P[f"{pre}.attention.linear_layers.{j}.bias"]
        qkv.append((y @ w.t() + bb).reshape(B, T, H, dk).permute(0, 2, 1, 3))
    q, k, v = qkv
    S = (q @ k.transpose(-2, -1)) / math.sqrt(dk)                        # :150-151
    S = S.masked_fill(~key_keep[:, None, None, :], -1e9)                 # :155
    S = S - S.max(-1, keepdim=True).values
    E = torch.exp(S)
    A = E / E.sum(-1, keepdim=True)                                      # :157
    A = _drop(A, masks, f"{pre}.attn", p)                                # :160
    o = (A @ v).permute(0, 2, 1, 3).reshape(B, T, D)                     # :194
    o = o @ P[f"{pre}.attention.output_linear.weight"].t() + P[f"{pre}.attention.output_linear.bias"]
    x = x + _drop(o, masks, f"{pre}.sub_in", p)                          # :142
    y = bert_layer_norm(x, P[f"{pre}.output_sublayer.norm.a_2"], P[f"{pre}.output_sublayer.norm.b_2"])
    h = gelu_tanh(y @ P[f"{pre}.feed_forward.w_1.weight"].t() + P[f"{pre}.feed_forward.w_1.bias"])
    h = _drop(h, masks, f"{pre}.ffn", p)                                 # :217
    z = h @ P[f"{pre}.feed_forward.w_2.weight"].t() + P[f"{pre}.feed_forward.w_2.bias"]
    x = x + _drop(z, masks, f"{pre}.sub_out", p)
    return _drop(x, masks, f"{pre}.block", p)                            # :245


def bert4rec_forward(P: Params, i_node: torch.Tensor, neg_samples: torch.Tensor, seq_d1: torch.Tensor,
                     seq_d2: torch.Tensor, masks: Masks = None, isDR: bool = False, isInC: bool = False, isItC: bool = False,
                     threshold1: float = 0.5, threshold2: float = 0.5, taps: Optional[dict] = None) -> Tuple[torch.Tensor, ...]:
    """BERT4Rec.forward model_seq.py:277-309; isDR: the three heads of :301-305.  isInC (:283-286) / isItC (:289-294): the comp
    module runs on the gathered rows BEFORE the encoders (unlike SASRec's InterComp), which then see 2T tokens; the key mask is the
    T-token mask tiled twice along the key axis (`.repeat(1, 2T, 2)`).  Both flags together make the reference fail (the mask
    keeps 2T keys while the sequences have 4T tokens, :291-294), so that combination is refused here as well."""
    if isInC and isItC:
        raise ValueError("BERT4Rec(isInC=True, isItC=True): the reference's own mask has 2T keys for 4T tokens (model_seq.py:294)")
    E = P["item_emb_layer.emb_item.weight"]
    i_feat = gather_rows(E, i_node).unsqueeze(1)
    neg_feat = gather_rows(E, neg_samples)
    x1 = gather_rows(E, seq_d1)
    x2 = gather_rows(E, seq_d2)
    key_keep = seq_d2 > 0                                                # :288 ONE mask, from domain 2, for BOTH encoders
    if isInC:
        x1 = inner_comp(x1, P, "inc_d1", threshold1, taps)               # :284
        x2 = inner_comp(x2, P, "inc_d2", threshold1, taps)               # :285
        key_keep = torch.cat((key_keep, key_keep), 1)                    # :286
    if isItC:
        e1, e2 = x1, x2
        x1 = inter_comp(e1, e2, P, "itc_d1", threshold2, taps)           # :292
        x2 = inter_comp(e2, e1, P, "itc_d2", threshold2, taps)           # :293
        key_keep = torch.cat((key_keep, key_keep), 1)                    # :294
    for l in range(2):
        x1 = bert_block(x1, key_keep, P, f"transform1.{l}", masks)       # :295-296
    for l in range(2):
        x2 = bert_block(x2, key_keep, P, f"transform2.{l}", masks)       # :297-298
    u1, u2 = x1.mean(1), x2.mean(1)                                      # :299-300
    items = torch.cat((i_feat, neg_feat), 1)
    if isDR:
        return (predict_module(u1, u2, items, P) + predict_module(u1, u2, items, P, "predict_ips")
                + predict_module(u1, u2, items, P, "predict_gfunc"))
    return predict_module(u1, u2, items, P)


# --------------------------------------------------------------------------
# a8  loss                                     train_sr.py:184, :203-212
# --------------------------------------------------------------------------
def bce_elementwise(p: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """nn.BCELoss(reduce=False): log terms clamped at -100 (torch semantics)."""
    lp = torch.clamp(torch.log(p), min=-100.0)
    l1p = torch.clamp(torch.log(1.0 - p), min=-100.0)
    return -(y * lp + (1.0 - y) * l1p)


def masked_bce_loss(p1: torch.Tensor, p2: torch.Tensor, labels: torch.Tensor, domain_id: torch.Tensor) -> torch.Tensor:
    """mean over B*(1+neg) of BCE(p1)*(1-domain) + BCE(p2)*domain (train_sr.py:205-211)."""
    m2 = domain_id.to(p1.dtype).unsqueeze(1)
    m1 = 1.0 - m2
    return (bce_elementwise(p1, labels) * m1 + bce_elementwise(p2, labels) * m2).mean()


def dr_losses(outs, labels: torch.Tensor, domain_id: torch.Tensor, ob_label: Optional[torch.Tensor] = None, dr_e_w: float = 0.1):
    """The doubly-robust trainer's losses (next-4; train_sr_dr.py:216-221 and :392-394) on the six outputs of
    sasrec_forward(isDR=True).  Returns (loss_cls, loss_dr_e, loss_cls + dr_e_w * loss_dr_e, loss_dr_r); loss_dr_r is None
    without ob_label.  The first total drives optimizer (lr), loss_dr_r drives optimizer2 (lr * lr2)."""
    p1, p2, i1, i2, g1, g2 = outs
    m2 = domain_id.to(p1.dtype).unsqueeze(1)
    m1 = 1.0 - m2
    b1, b2 = bce_elementwise(p1, labels), bce_elementwise(p2, labels)
    loss_cls = (b1 * m1 + b2 * m2).mean()
    loss_dr_e = ((b1 - g1) ** 2 / i1 * m1 + (b2 - g2) ** 2 / i2 * m2).mean()
    loss_dr_r = None
    if ob_label is not None:
        ob = ob_label.to(p1.dtype).unsqueeze(1).expand_as(p1)
        loss_dr_r = ((g1 ** 2 + ob * (b1 ** 2 - g1 ** 2) ** 2 / i1) * m1 + (g2 ** 2 + ob * (b2 ** 2 - g2 ** 2) ** 2 / i2) * m2).mean()
    return loss_cls, loss_dr_e, loss_cls + dr_e_w * loss_dr_e, loss_dr_r


def dr_loss_and_grads(P: Params, batch: Dict[str, torch.Tensor], which: str, masks: Masks = None, dr_e_w: float = 0.1,
                      model: str = "sasrec", **fwd_kw):
    """which = "e": gradient of loss_cls + dr_e_w * loss_dr_e (first loop); "r": gradient of loss_dr_r (second loop)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    fwd = sasrec_forward if model == "sasrec" else bert4rec_forward
    outs = fwd(leaves, batch["i_node"], batch["neg_samples"], batch["seq_d1"], batch["seq_d2"], masks, isDR=True, **fwd_kw)
    lc, le, tot, lr_ = dr_losses(outs, batch["label"], batch["domain_id"], batch.get("ob_label"), dr_e_w)
    loss = tot if which == "e" else lr_
    names = list(leaves)
    gs = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    grads = {n: (g if g is not None else torch.zeros_like(leaves[n])) for n, g in zip(names, gs)}
    return dict(loss_cls=lc.detach(), loss_dr_e=le.detach(), loss=loss.detach()), tuple(o.detach() for o in outs), grads


# --------------------------------------------------------------------------
# a10 optimizer: torch.optim.Adam(lr) dense over every parameter, table
#     included (train_sr.py:480); op order of torch's single-tensor CPU Adam.
# --------------------------------------------------------------------------
class DenseAdam:
    def __init__(self, params: Params, lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, betas[0], betas[1], eps
        self.t = 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    @torch.no_grad()
    def step(self, params: Params, grads: Dict[str, torch.Tensor]) -> None:
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        step_size = self.lr / bc1
        bc2_sqrt = bc2 ** 0.5
        for k, p in params.items():
            g = grads.get(k)
            if g is None:
                continue
            m, v = self.m[k], self.v[k]
            m.lerp_(g, 1.0 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            denom = (v.sqrt() / bc2_sqrt).add_(self.eps)
            p.addcdiv_(m, denom, value=-step_size)


# --------------------------------------------------------------------------
# a7+a8+a9+a10  one training step (train_sr.py:190-217)
# --------------------------------------------------------------------------
def loss_and_grads(model: str, P: Params, batch: Dict[str, torch.Tensor], masks: Masks = None, **fwd_kw
                   ) -> Tuple[torch.Tensor, Tuple[torch.Tensor, torch.Tensor], Dict[str, torch.Tensor]]:
    """Forward + masked BCE + autograd backward.  Embedding grads are DENSE
    (nn.Embedding(sparse=False), model_seq.py:25) exactly as in the reference."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    fwd = sasrec_forward if model == "sasrec" else bert4rec_forward
    p1, p2 = fwd(leaves, batch["i_node"], batch["neg_samples"], batch["seq_d1"], batch["seq_d2"], masks, **fwd_kw)
    loss = masked_bce_loss(p1, p2, batch["label"], batch["domain_id"])
    names = list(leaves)
    gs = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    grads = {n: (g if g is not None else torch.zeros_like(leaves[n])) for n, g in zip(names, gs)}
    return loss.detach(), (p1.detach(), p2.detach()), grads


def train_step(model: str, P: Params, opt: DenseAdam, batch: Dict[str, torch.Tensor], masks: Masks = None, **fwd_kw) -> float:
    loss, _, grads = loss_and_grads(model, P, batch, masks, **fwd_kw)
    opt.step(P, grads)
    return float(loss)


# --------------------------------------------------------------------------
# a11 batch marshal                            dataset_seq.py:12-22, :252-274 ; train_sr.py:191-200
# --------------------------------------------------------------------------
def seq_padding(seq: Sequence[int], length_enc: int, long_length: int, pad_id: int):
    """dataset_seq.py:12-22 (called with length_enc = seq_len + 1)."""
    long_mask = 1 if len(seq) >= long_length else 0
    if len(seq) >= length_enc:
        enc_in = list(seq[-length_enc + 1:])
    else:
        enc_in = [pad_id] * (length_enc - len(seq) - 1) + list(seq)
    return enc_in, long_mask


def wire_to_long(x: torch.Tensor) -> torch.Tensor:
    """ids arrive as float32 (dataset_seq.py:253-262) and are cast with .long()
    (train_sr.py:191-199); exact below 2**24."""
    return x.long()


# --------------------------------------------------------------------------
# metrics (next-2)                             utils.py:296-312
# --------------------------------------------------------------------------
def get_sample_scores(pred: np.ndarray):
    rank = (-pred).argsort().argsort()[:, 0]
    out = []
    mrr = 0.0
    for topk in (1, 5, 10):
        ndcg = hit = mrr = 0.0
        for r in rank:
            mrr += 1.0 / (r + 1.0)
            if r < topk:
                ndcg += 1.0 / np.log2(r + 2.0)
                hit += 1.0
        out += [hit / len(rank), ndcg / len(rank)]
    return tuple(out) + (mrr / len(rank),)


# --------------------------------------------------------------------------
# parameter construction helpers for tests / bench (shapes as SURVEY 8(b))
# --------------------------------------------------------------------------
def sasrec_param_shapes(item_length: int, D: int, T: int, hid: int, itc_bs: int = 0, dr: bool = False,
                        inc_bs: int = 0) -> Dict[str, Tuple[int, ...]]:
    """itc_bs > 0: also the InterComp parameters of SASRec(isItC=True, bs=itc_bs) (model_seq.py:403-405, :478-480);
    dr: also the predict_ips / predict_gfunc heads of isDR=True (:411-414); inc_bs > 0: the InnerComp parameters of
    SASRec(isInC=True, bs=inc_bs) and pos_emb tables of 2T rows (:398-401: seq_len *= 2)."""
    s: Dict[str, Tuple[int, ...]] = {"item_emb_layer.emb_item.weight": (item_length, D)}
    if inc_bs:
        T = 2 * T
        for d in (1, 2):
            s[f"inc_d{d}.trans_nn.weight"] = (D, D)
            s[f"inc_d{d}.trans_nn.bias"] = (D,)
            s[f"inc_d{d}.trans_bs.weight"] = (1, inc_bs)
            s[f"inc_d{d}.trans_bs.bias"] = (1,)
    if itc_bs:
        for d in (1, 2):
            s[f"itc_d{d}.trans_nn.weight"] = (D, D)
            s[f"itc_d{d}.trans_nn.bias"] = (D,)
            s[f"itc_d{d}.trans_bs.weight"] = (1, itc_bs)
            s[f"itc_d{d}.trans_bs.bias"] = (1,)
    for d in (1, 2):
        pre = f"sac{d}"
        s[f"{pre}.pos_emb.weight"] = (T, D)
        s[f"{pre}.last_layernorm.weight"] = (D,)
        s[f"{pre}.last_layernorm.bias"] = (D,)
        for l in (0, 1):
            s[f"{pre}.attention_layernorms.{l}.weight"] = (D,)
            s[f"{pre}.attention_layernorms.{l}.bias"] = (D,)
            s[f"{pre}.attention_layers.{l}.in_proj_weight"] = (3 * D, D)
            s[f"{pre}.attention_layers.{l}.in_proj_bias"] = (3 * D,)
            s[f"{pre}.attention_layers.{l}.out_proj.weight"] = (D, D)
            s[f"{pre}.attention_layers.{l}.out_proj.bias"] = (D,)
            s[f"{pre}.forward_layernorms.{l}.weight"] = (D,)
            s[f"{pre}.forward_layernorms.{l}.bias"] = (D,)
            s[f"{pre}.forward_layers.{l}.conv1.weight"] = (D, D, 1)
            s[f"{pre}.forward_layers.{l}.conv1.bias"] = (D,)
            s[f"{pre}.forward_layers.{l}.conv2.weight"] = (D, D, 1)
            s[f"{pre}.forward_layers.{l}.conv2.bias"] = (D,)
    for head in ("predictModule",) + (("predict_ips", "predict_gfunc") if dr else ()):
        s[f"{head}.fc.0.weight"] = (hid, 2 * D)
        s[f"{head}.fc.0.bias"] = (hid,)
        s[f"{head}.fc.2.weight"] = (1, hid)
        s[f"{head}.fc.2.bias"] = (1,)
    return s


def bert4rec_param_shapes(item_length: int, hid: int, dr: bool = False, inc_bs: int = 0, itc_bs: int = 0) -> Dict[str, Tuple[int, ...]]:
    """dr: also the predict_ips / predict_gfunc heads of isDR=True (model_seq.py:268-271); inc_bs / itc_bs > 0: the InnerComp /
    InterComp modules of isInC / isItC with bs rows (:257-263)."""
    D, F = BERT_HIDDEN, BERT_FF
    s: Dict[str, Tuple[int, ...]] = {"item_emb_layer.emb_item.weight": (item_length, D)}
    for kind, bs in (("inc", inc_bs), ("itc", itc_bs)):
        if bs:
            for d in (1, 2):
                s[f"{kind}_d{d}.trans_nn.weight"] = (D, D)
                s[f"{kind}_d{d}.trans_nn.bias"] = (D,)
                s[f"{kind}_d{d}.trans_bs.weight"] = (1, bs)
                s[f"{kind}_d{d}.trans_bs.bias"] = (1,)
    for d in (1, 2):
        for l in (0, 1):
            pre = f"transform{d}.{l}"
            for j in range(3):
                s[f"{pre}.attention.linear_layers.{j}.weight"] = (D, D)
                s[f"{pre}.attention.linear_layers.{j}.bias"] = (D,)
            s[f"{pre}.attention.output_linear.weight"] = (D, D)
            s[f"{pre}.attention.output_linear.bias"] = (D,)
            s[f"{pre}.feed_forward.w_1.weight"] = (F, D)
            s[f"{pre}.feed_forward.w_1.bias"] = (F,)
            s[f"{pre}.feed_forward.w_2.weight"] = (D, F)
            s[f"{pre}.feed_forward.w_2.bias"] = (D,)
            s[f"{pre}.input_sublayer.norm.a_2"] = (D,)
            s[f"{pre}.input_sublayer.norm.b_2"] = (D,)
            s[f"{pre}.output_sublayer.norm.a_2"] = (D,)
            s[f"{pre}.output_sublayer.norm.b_2"] = (D,)
    for head in ("predictModule",) + (("predict_ips", "predict_gfunc") if dr else ()):
        s[f"{head}.fc.0.weight"] = (hid, 2 * D)
        s[f"{head}.fc.0.bias"] = (hid,)
        s[f"{head}.fc.2.weight"] = (1, hid)
        s[f"{head}.fc.2.bias"] = (1,)
    return s


def random_params(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, scale: float = 0.1) -> Params:
    """Synthetic random-init parameters (norm gains near 1) for parity tests."""
    g = torch.Generator().manual_seed(seed)
    P: Params = {}
    for k, shp in shapes.items():
        t = torch.randn(*shp, generator=g) * scale
        if k.endswith("layernorm.weight") or k.endswith("layernorms.0.weight") or k.endswith("layernorms.1.weight") \
                or k.endswith(".a_2"):
            t = 1.0 + t
        if k == "item_emb_layer.emb_item.weight" or k.endswith("pos_emb.weight"):
            t = torch.randn(*shp, generator=g)      # nn.Embedding default init N(0,1)
        P[k] = t.contiguous()
    return P


def synthetic_batch(B: int, T: int, n_items: int, pad_id: int, neg: int = 1, seed: int = 0,
                    mean_len: float = 5.0) -> Dict[str, torch.Tensor]:
    """Batch with the reference's layout: left-padded sequences,
    short real lengths, labels [1,0...], domain ids ~ Bernoulli(.5)."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.clamp((torch.rand(2, B, generator=g) * 2 * mean_len).long(), 0, T)
    seqs = []
    for d in range(2):
        s = torch.full((B, T), pad_id, dtype=torch.long)
        ids = torch.randint(1, n_items, (B, T), generator=g)
        col = torch.arange(T).unsqueeze(0)
        real = col >= (T - lens[d]).unsqueeze(1)
        s[real] = ids[real]
        seqs.append(s)
    label = torch.zeros(B, 1 + neg)
    label[:, 0] = 1.0
    return {
        "i_node": torch.randint(1, n_items, (B,), generator=g),
        "neg_samples": torch.randint(1, n_items, (B, neg), generator=g),
        "seq_d1": seqs[0], "seq_d2": seqs[1],
        "domain_id": (torch.rand(B, generator=g) < 0.5).long(),
        "label": label,
    }
