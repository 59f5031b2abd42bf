#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, read-only).  It imports
the reference's own ``model_seq`` / ``dataset_seq`` / ``utils`` modules with the
three-line ``.cuda()`` no-op shim from SURVEY.md section 8(c) and stores inputs +
outputs as small ``.npz`` fixtures.  Nothing of the reference's source travels:
the fixtures are data (parameters, index tensors, outputs, gradients, masks).

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

Fixture inventory:
  g1_gather.npz          embItemLayerEnhance forward (bit-exact rows)
  g2_log2feats_*.npz     Log2feats eval forward, D in {32,128}
  g3_sasrec_eval.npz     SASRec eval logits from a saved state_dict
  g3_bert4rec_eval.npz   BERT4Rec eval logits
  g4_sasrec_grads.npz    loss + grads of every parameter (dropout off)
  g4_bert4rec_grads.npz
  g5_sasrec_train.npz    train mode, dropout masks recorded -> logits/loss/grads
  g5_bert4rec_train.npz
  g6_adam_traj.npz       dense torch.optim.Adam trajectory, rows touched then idle
  g7_marshal.npz         seq_padding / __getitem__ / collate_fn_enhance layout
  g8_metrics.npz         get_sample_scores on a fixed prediction matrix
  g9_comp.npz            InterComp / InnerComp forward (next-1)
  g10_sasrec_itc.npz     SASRec(isItC=True) logits, loss, all grads -- written by make_golden_itc.py
"""
import os
import random
import sys
import tempfile

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

# ---- shim (reference files untouched) ------------------------------------
torch.Tensor.cuda = lambda s, *a, **k: s
torch.nn.Module.cuda = lambda s, *a, **k: s
_ones = torch.ones


def _ones_nodev(*a, **k):
    k.pop("device", None)
    return _ones(*a, **k)


torch.ones = _ones_nodev
sys.path.insert(0, REF)
import model_seq  # noqa: E402
import dataset_seq  # noqa: E402
import utils as ref_utils  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def sd_np(m):
    return {"P/" + k: v.detach().numpy().copy() for k, v in m.state_dict().items()}


def rand_batch(B, T, n_items, neg, seed, pad_id=None, mean_len=6):
    g = torch.Generator().manual_seed(seed)
    if pad_id is None:
        s1 = torch.randint(0, n_items, (B, T), generator=g)
        s2 = torch.randint(0, n_items, (B, T), generator=g)
    else:
        out = []
        for _ in range(2):
            s = torch.full((B, T), pad_id, dtype=torch.long)
            for b in range(B):
                L = int(torch.randint(0, 2 * mean_len, (1,), generator=g))
                L = min(L, T)
                if L:
                    s[b, T - L:] = torch.randint(1, n_items - 2, (L,), generator=g)
            out.append(s)
        s1, s2 = out
    return dict(
        i_node=torch.randint(1, n_items - 2, (B,), generator=g),
        neg_samples=torch.randint(1, n_items - 2, (B, neg), generator=g),
        seq_d1=s1, seq_d2=s2,
        domain_id=(torch.rand(B, generator=g) < 0.5).long(),
    )


def labels_for(B, neg):
    y = torch.zeros(B, 1 + neg)
    y[:, 0] = 1.0
    return y


def ref_loss(p1, p2, labels, domain_id):
    # train_sr.py:203-212 evaluated verbatim through torch's BCELoss
    crit = torch.nn.BCELoss(reduction="none")
    m1 = (1 - domain_id).unsqueeze(1)
    m2 = domain_id.unsqueeze(1)
    return torch.mean(crit(p1, labels) * m1 + crit(p2, labels) * m2)


def batch_np(b):
    return {"B/" + k: v.numpy().copy() for k, v in b.items()}


class MaskRecorder:
    """Replaces torch.nn.functional.dropout: draws its own keep mask, records it."""

    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.masks = []

    def __call__(self, input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return input
        keep = (torch.rand(input.shape, generator=self.g) >= p).to(input.dtype)
        self.masks.append(keep)
        return input * keep / (1.0 - p)


def run_recorded(model, batch, seed):
    rec = MaskRecorder(seed)
    orig = F.dropout
    F.dropout = rec
    torch.nn.functional.dropout = rec
    try:
        model.train()
        out = model(None, batch["i_node"], batch["neg_samples"], batch["seq_d1"].clone(), batch["seq_d2"].clone(), None, None)
    finally:
        F.dropout = orig
        torch.nn.functional.dropout = orig
    return out, rec.masks


def grads_np(model):
    return {"G/" + k: (p.grad.detach().numpy().copy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32))
            for k, p in model.named_parameters()}


def main():
    torch.set_num_threads(4)
    sys.path.insert(0, os.path.join(OUT, "..", ".."))

    # ---- G1 gather -------------------------------------------------------
    torch.manual_seed(1)
    emb = model_seq.embItemLayerEnhance(300, 48)
    idx = torch.randint(0, 300, (5, 7))
    np.savez_compressed(os.path.join(OUT, "g1_gather.npz"), table=emb.emb_item.weight.detach().numpy(),
                        idx=idx.numpy(), rows=emb(idx).detach().numpy())

    # ---- G2 Log2feats eval forward --------------------------------------
    for D in (32, 128):
        torch.manual_seed(2 + D)
        enc = model_seq.Log2feats(10, D, 100, D, 50, 16)
        with torch.no_grad():  # make norms / biases non-trivial
            for n, p in enc.named_parameters():
                if "layernorm" in n:
                    p.add_(0.1 * torch.randn_like(p))
                if n.endswith("bias"):
                    p.add_(0.05 * torch.randn_like(p))
        enc.eval()
        x = torch.randn(3 if D == 128 else 6, 50, D)
        x[0, :10] = 0.0     # rows of zeros: after pos-add they are non-zero; exercises the ==0 feature mask path weakly
        xin = x.clone()
        with torch.no_grad():
            y = enc(x)       # NB: modifies x in place (pos add), xin is the pristine input
        np.savez_compressed(os.path.join(OUT, f"g2_log2feats_d{D}.npz"), x=xin.numpy(), y=y.numpy(),
                            **{"P/sac1." + k: v.detach().numpy() for k, v in enc.state_dict().items()})

    # ---- G3/G4/G5 SASRec -------------------------------------------------
    n_items, D, T, hid, B, neg = 600, 64, 50, 16, 6, 1
    torch.manual_seed(3)
    m = model_seq.SASRec(10, D, n_items, D, T, hid, B, False, False, 0.5, 0.5)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "layernorm" in n or n.endswith("bias"):
                p.add_(0.1 * torch.randn_like(p))
    batch = rand_batch(B, T, n_items, neg, seed=30, pad_id=n_items - 1)
    labels = labels_for(B, neg)
    m.eval()
    with torch.no_grad():
        p1, p2 = m(None, batch["i_node"], batch["neg_samples"], batch["seq_d1"].clone(), batch["seq_d2"].clone(), None, None, False)
    np.savez_compressed(os.path.join(OUT, "g3_sasrec_eval.npz"), p1=p1.numpy(), p2=p2.numpy(), **sd_np(m), **batch_np(batch))
    # eval with many negatives (test() path, train_sr.py:55-56)
    batch_e = rand_batch(B, T, n_items, 9, seed=31, pad_id=n_items - 1)
    with torch.no_grad():
        q1, q2 = m(None, batch_e["i_node"], batch_e["neg_samples"], batch_e["seq_d1"].clone(), batch_e["seq_d2"].clone(), None, None, False)
    np.savez_compressed(os.path.join(OUT, "g3_sasrec_eval_neg9.npz"), p1=q1.numpy(), p2=q2.numpy(), **sd_np(m), **batch_np(batch_e))

    m.zero_grad()
    p1, p2 = m(None, batch["i_node"], batch["neg_samples"], batch["seq_d1"].clone(), batch["seq_d2"].clone(), None, None)
    loss = ref_loss(p1.squeeze(), p2.squeeze(), labels, batch["domain_id"])
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "g4_sasrec_grads.npz"), p1=p1.detach().numpy(), p2=p2.detach().numpy(),
                        loss=loss.detach().numpy(), labels=labels.numpy(), **sd_np(m), **batch_np(batch), **grads_np(m))

    # pick a mask seed whose live relu pre-activations all stay clear of the kink at 0:
    # an fp32 rounding flip there changes relu' discontinuously and would make the
    # fixture depend on summation order (seen: 6e-3 relative on conv1.weight.grad)
    from oracle import amid_oracle as orc
    for mask_seed in range(55, 200):
        m.zero_grad()
        (p1, p2), masks = run_recorded(m, batch, seed=mask_seed)
        assert len(masks) == 14, len(masks)
        om = {}
        for d in range(2):
            ms = masks[7 * d: 7 * d + 7]
            om[f"sac{d + 1}.emb"] = ms[0]
            for l in range(2):
                a, f1, f2 = ms[1 + 3 * l: 4 + 3 * l]
                om[f"sac{d + 1}.attn{l}"] = a.reshape(B, 8, T, T)
                om[f"sac{d + 1}.ffn1_{l}"] = f1.transpose(1, 2)
                om[f"sac{d + 1}.ffn2_{l}"] = f2.transpose(1, 2)
        taps = {}
        orc.sasrec_forward({k: v.detach().double() for k, v in m.state_dict().items()}, batch["i_node"], batch["neg_samples"],
                           batch["seq_d1"], batch["seq_d2"], {k: v.double() for k, v in om.items()}, taps)
        margin = min(taps[s][f"relu_margin{l}"] for s in ("sac1", "sac2") for l in (0, 1))
        if margin > 3e-5:
            break
    print("g5 sasrec mask seed", mask_seed, "relu margin", margin)
    loss = ref_loss(p1.squeeze(), p2.squeeze(), labels, batch["domain_id"])
    loss.backward()
    mk = {}
    for d in range(2):
        ms = masks[7 * d: 7 * d + 7]
        pre = f"M/sac{d + 1}"
        mk[f"{pre}.emb"] = ms[0].numpy().astype(np.uint8)                                       # [B,T,D]
        for l in range(2):
            a, f1, f2 = ms[1 + 3 * l: 4 + 3 * l]
            mk[f"{pre}.attn{l}"] = a.reshape(B, 8, T, T).numpy().astype(np.uint8)             # [B*H,T,T] -> [B,H,T,T]
            mk[f"{pre}.ffn1_{l}"] = f1.transpose(1, 2).contiguous().numpy().astype(np.uint8)  # [B,D,T] -> [B,T,D]
            mk[f"{pre}.ffn2_{l}"] = f2.transpose(1, 2).contiguous().numpy().astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "g5_sasrec_train.npz"), p1=p1.detach().numpy(), p2=p2.detach().numpy(),
                        loss=loss.detach().numpy(), labels=labels.numpy(), **sd_np(m), **batch_np(batch), **grads_np(m), **mk)

    # ---- G3/G4/G5 BERT4Rec ----------------------------------------------
    # Parameters are NOT stored (3 MB per copy): they are regenerated from
    # oracle.random_params(seed) on both sides and loaded into the reference
    # model here; a checksum guards against generator drift.
    n_items, T, hid, B, neg = 400, 20, 16, 4, 1
    mb = model_seq.BERT4Rec(10, 128, n_items, 128, T, hid, B, False, False, 0.5, 0.5)
    Pb = orc.random_params(orc.bert4rec_param_shapes(n_items, hid), seed=404)
    mb.load_state_dict(Pb, strict=True)
    psum = np.float64(sum(float(v.double().sum()) for v in Pb.values()))
    meta = dict(n_items=n_items, T=T, hid=hid, param_seed=404, param_sum=psum)
    batch = rand_batch(B, T, n_items, neg, seed=40, pad_id=n_items - 1)
    batch["seq_d2"][0, 5] = 0       # literal id 0 is the only key the mask removes (model_seq.py:288)
    batch["seq_d2"][2, T - 1] = 0
    labels = labels_for(B, neg)
    mb.eval()
    with torch.no_grad():
        p1, p2 = mb(None, batch["i_node"], batch["neg_samples"], batch["seq_d1"], batch["seq_d2"], None, None, False)
    np.savez_compressed(os.path.join(OUT, "g3_bert4rec_eval.npz"), p1=p1.numpy(), p2=p2.numpy(), **meta, **batch_np(batch))
    mb.zero_grad()
    p1, p2 = mb(None, batch["i_node"], batch["neg_samples"], batch["seq_d1"], batch["seq_d2"], None, None)
    loss = ref_loss(p1.squeeze(), p2.squeeze(), labels, batch["domain_id"])
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "g4_bert4rec_grads.npz"), p1=p1.detach().numpy(), p2=p2.detach().numpy(),
                        loss=loss.detach().numpy(), labels=labels.numpy(), **meta, **batch_np(batch), **grads_np(mb))
    mb.zero_grad()
    (p1, p2), masks = run_recorded(mb, batch, seed=66)
    loss = ref_loss(p1.squeeze(), p2.squeeze(), labels, batch["domain_id"])
    loss.backward()
    assert len(masks) == 20, len(masks)
    mk = {}
    for d in range(2):
        for l in range(2):
            a, si, ff, so, bl = masks[10 * d + 5 * l: 10 * d + 5 * l + 5]
            pre = f"M/transform{d + 1}.{l}"
            mk[f"{pre}.attn"] = np.packbits(a.numpy().astype(np.uint8))
            mk[f"{pre}.sub_in"] = np.packbits(si.numpy().astype(np.uint8))
            mk[f"{pre}.ffn"] = np.packbits(ff.numpy().astype(np.uint8))
            mk[f"{pre}.sub_out"] = np.packbits(so.numpy().astype(np.uint8))
            mk[f"{pre}.block"] = np.packbits(bl.numpy().astype(np.uint8))
    # train-mode grads: everything except the 16 large projection matrices (kept whole in g4)
    gsmall = {k: v for k, v in grads_np(mb).items() if v.size < 16384 or "emb_item" in k}
    np.savez_compressed(os.path.join(OUT, "g5_bert4rec_train.npz"), p1=p1.detach().numpy(), p2=p2.detach().numpy(),
                        loss=loss.detach().numpy(), labels=labels.numpy(), **meta, **batch_np(batch), **gsmall, **mk)

    # ---- G6 dense Adam trajectory (train_sr.py:480, :213-215) -------------
    n_items, D, T, hid, B, neg = 200, 16, 12, 8, 4, 1
    torch.manual_seed(6)
    m = model_seq.SASRec(10, D, n_items, D, T, hid, B, False, False, 0.5, 0.5)
    m.eval()                                   # dropout off => deterministic trajectory
    opt = torch.optim.Adam(m.parameters(), lr=5e-3)
    labels = labels_for(B, neg)
    traj = {}
    batches = {}
    init = sd_np(m)
    for step in range(1, 21):
        # items 1..40 appear only in steps 1-3 and again at step 15: long idle gaps
        lo, hi = (1, 40) if step in (1, 2, 3, 15) else (60, 190)
        g = torch.Generator().manual_seed(600 + step)
        b = dict(
            i_node=torch.randint(lo, hi, (B,), generator=g),
            neg_samples=torch.randint(lo, hi, (B, neg), generator=g),
            seq_d1=torch.randint(lo, hi, (B, T), generator=g),
            seq_d2=torch.randint(lo, hi, (B, T), generator=g),
            domain_id=(torch.rand(B, generator=g) < 0.5).long(),
        )
        b["seq_d1"][:, : T // 2] = n_items - 1          # pad row touched every step
        b["seq_d2"][:, : T // 3] = n_items - 1
        for k, v in b.items():
            batches[f"S{step}/{k}"] = v.numpy().copy()
        p1, p2 = m(None, b["i_node"], b["neg_samples"], b["seq_d1"].clone(), b["seq_d2"].clone(), None, None)
        loss = ref_loss(p1.squeeze(), p2.squeeze(), labels, b["domain_id"])
        opt.zero_grad()
        loss.backward()
        opt.step()
        traj[f"L/{step}"] = loss.detach().numpy().copy()
        if step in (1, 2, 5, 20):
            for k, v in m.state_dict().items():
                traj[f"T{step}/{k}"] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g6_adam_traj.npz"), labels=labels.numpy(), lr=np.float64(5e-3), **init, **batches, **traj)

    # ---- G7 batch marshal (dataset_seq.py:12-22, :177-274) ----------------
    import pandas as pd
    df = pd.read_csv(os.path.join(REF, "amazon_dataset", "cloth_sport_train25.csv")).head(24)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "rows.csv")
        df.to_csv(path, index=False)
        pad_id, seq_len = 447411, 50
        ds = dataset_seq.DualDomainSeqDataset(seq_len=seq_len, isTrain=True, neg_nums=199, long_length=7, pad_id=pad_id, csv_path=path)
        random.seed(0)
        samples = [ds[i] for i in range(8)]
        coll = dataset_seq.collate_fn_enhance(samples)
        g7 = {"C/" + k: v.numpy() for k, v in coll.items()}
        g7["C_dtype"] = np.array([str(v.dtype) for v in coll.values()])
        g7["C_keys"] = np.array(list(coll.keys()))
        pads = []
        for L in (0, 1, 5, 49, 50, 51, 80):
            enc, lm = dataset_seq.seq_padding(list(range(1, L + 1)), seq_len + 1, 7, pad_id)
            pads.append(np.array(enc + [lm]))
        g7["pad_cases"] = np.stack(pads)
        g7["pad_lens"] = np.array([0, 1, 5, 49, 50, 51, 80])
        g7["rows_user_id"] = df["user_id"].values[:8]
        g7["rows_seq_d1"] = df["seq_d1"].values[:8].astype(str)
        g7["rows_seq_d2"] = df["seq_d2"].values[:8].astype(str)
        g7["rows_domain_id"] = df["domain_id"].values[:8]
        np.savez_compressed(os.path.join(OUT, "g7_marshal.npz"), pad_id=pad_id, seq_len=seq_len, **g7)

    # ---- G8 metrics (utils.py:296-312) ------------------------------------
    rng = np.random.RandomState(8)
    pred = rng.rand(64, 30).astype(np.float32)
    pred[:, 0] += rng.rand(64).astype(np.float32) * 0.5
    pred[3, 1] = pred[3, 0]     # tie
    scores = np.array(ref_utils.get_sample_scores(pred.copy()), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "g8_metrics.npz"), pred=pred, scores=scores)

    # ---- G9 InterComp / InnerComp forward (model_seq.py:450-497) ----------
    torch.manual_seed(9)
    bs, T, D = 5, 7, 16
    itc = model_seq.InterComp(D, bs, 0.15)
    inc = model_seq.InnerComp(D, bs, 0.15)
    a = torch.randn(bs, T, D) * 0.3
    b = torch.randn(bs, T, D) * 0.3
    with torch.no_grad():
        yo = itc(a, b)
        yi = inc(a)
    np.savez_compressed(os.path.join(OUT, "g9_comp.npz"), a=a.numpy(), b=b.numpy(), inter=yo.numpy(), inner=yi.numpy(),
                        **{"P/itc." + k: v.numpy() for k, v in itc.state_dict().items()},
                        **{"P/inc." + k: v.numpy() for k, v in inc.state_dict().items()})
    print("golden vectors written to", OUT)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f"  {f:28s} {os.path.getsize(os.path.join(OUT, f)) / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
