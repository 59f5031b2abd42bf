"""Pins oracle/amid_oracle.py against the golden vectors produced by running the
reference itself (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import amid_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    P = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("P/")}
    B = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("B/")}
    G = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("G/")}
    M = {k[2:]: z[k] for k in z.files if k.startswith("M/")}
    return z, P, B, G, M


def rel_err(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_g1_gather_bit_exact():
    z = np.load(os.path.join(GOLDEN, "g1_gather.npz"))
    rows = orc.gather_rows(torch.from_numpy(z["table"]), torch.from_numpy(z["idx"]))
    assert np.array_equal(rows.numpy(), z["rows"])


@pytest.mark.parametrize("D", [32, 128])
def test_g2_log2feats_eval(D):
    z, P, *_ = load(f"g2_log2feats_d{D}.npz")
    y = orc.sasrec_encoder(torch.from_numpy(z["x"]), P, "sac1")
    assert float((y - torch.from_numpy(z["y"])).abs().max()) < 2e-6


@pytest.mark.parametrize("name", ["g3_sasrec_eval.npz", "g3_sasrec_eval_neg9.npz"])
def test_g3_sasrec_eval(name):
    z, P, B, *_ = load(name)
    p1, p2 = orc.sasrec_forward(P, B["i_node"], B["neg_samples"], B["seq_d1"], B["seq_d2"])
    assert rel_err(p1, z["p1"]) < 1e-6 and rel_err(p2, z["p2"]) < 1e-6


def bert_params(z, dr=False, inc_bs=0, itc_bs=0):
    P = orc.random_params(orc.bert4rec_param_shapes(int(z["n_items"]), int(z["hid"]), dr=dr, inc_bs=inc_bs, itc_bs=itc_bs),
                          seed=int(z["param_seed"]))
    if "table_scale" in z.files:
        P["item_emb_layer.emb_item.weight"] = P["item_emb_layer.emb_item.weight"] * float(z["table_scale"])
    s = sum(float(v.double().sum()) for v in P.values())
    assert abs(s - float(z["param_sum"])) < 1e-9 * max(1.0, abs(s)), "random_params drifted from the fixture generator"
    return P


def test_g3_bert4rec_eval():
    z, _, B, *_ = load("g3_bert4rec_eval.npz")
    P = bert_params(z)
    p1, p2 = orc.bert4rec_forward(P, B["i_node"], B["neg_samples"], B["seq_d1"], B["seq_d2"])
    assert rel_err(p1, z["p1"]) < 1e-6 and rel_err(p2, z["p2"]) < 1e-6


def check_grads(model, z, P, B, G, masks, tol=2e-5, **fwd_kw):
    batch = dict(B)
    batch["label"] = torch.from_numpy(z["labels"])
    loss, (p1, p2), grads = orc.loss_and_grads(model, P, batch, masks, **fwd_kw)
    assert rel_err(p1, z["p1"]) < 1e-6 and rel_err(p2, z["p2"]) < 1e-6
    assert abs(float(loss) - float(z["loss"])) < 1e-6 * max(1.0, abs(float(z["loss"])))
    assert G, "fixture holds no grads"
    for k, g in G.items():
        # a few grads are analytically zero (e.g. the key bias: softmax is shift invariant) -> absolute floor
        assert rel_err(grads[k], g) < tol or float((grads[k] - g).abs().max()) < 1e-8, k


def test_g4_sasrec_grads():
    z, P, B, G, _ = load("g4_sasrec_grads.npz")
    check_grads("sasrec", z, P, B, G, None)


def test_g4_bert4rec_grads():
    z, _, B, G, _ = load("g4_bert4rec_grads.npz")
    check_grads("bert4rec", z, bert_params(z), B, G, None)


def test_g5_sasrec_train_recorded_masks():
    z, P, B, G, M = load("g5_sasrec_train.npz")
    masks = {k: torch.from_numpy(v.astype(np.float32)) for k, v in M.items()}
    check_grads("sasrec", z, P, B, G, masks)


def test_g5_bert4rec_train_recorded_masks():
    z, _, B, G, M = load("g5_bert4rec_train.npz")
    Bn, T = B["seq_d1"].shape
    shapes = {"attn": (Bn, 4, T, T), "sub_in": (Bn, T, 128), "ffn": (Bn, T, 512), "sub_out": (Bn, T, 128), "block": (Bn, T, 128)}
    masks = {}
    for k, v in M.items():
        shp = shapes[k.rsplit(".", 1)[1]]
        masks[k] = torch.from_numpy(np.unpackbits(v)[: int(np.prod(shp))].reshape(shp).astype(np.float32))
    check_grads("bert4rec", z, bert_params(z), B, G, masks)


def test_g6_dense_adam_trajectory():
    z, P, *_ = load("g6_adam_traj.npz")
    P = {k: v.clone() for k, v in P.items()}
    opt = orc.DenseAdam(P, lr=float(z["lr"]))
    labels = torch.from_numpy(z["labels"])
    for step in range(1, 21):
        batch = {k: torch.from_numpy(z[f"S{step}/{k}"]) for k in ("i_node", "neg_samples", "seq_d1", "seq_d2", "domain_id")}
        batch["label"] = labels
        loss = orc.train_step("sasrec", P, opt, batch)
        assert abs(loss - float(z[f"L/{step}"])) < 2e-6, step
        if step in (1, 2, 5, 20):
            for k, v in P.items():
                d = (v - torch.from_numpy(z[f"T{step}/{k}"])).abs()
                if k.endswith("in_proj_bias"):
                    # the key bias has an analytically zero gradient (softmax shift invariance); Adam turns its
                    # 1e-11 rounding noise into +-lr steps, so that slice is chaotic in the reference itself
                    D = v.numel() // 3
                    d = torch.cat((d[:D], d[2 * D:]))
                assert float(d.max()) < 2e-6, (step, k)


def test_g7_marshal():
    z = np.load(os.path.join(GOLDEN, "g7_marshal.npz"))
    pad_id, seq_len = int(z["pad_id"]), int(z["seq_len"])
    for L, want in zip(z["pad_lens"], z["pad_cases"]):
        enc, lm = orc.seq_padding(list(range(1, int(L) + 1)), seq_len + 1, 7, pad_id)
        assert enc + [lm] == list(want)
        assert len(enc) == seq_len
    # wire format: every collated tensor is float32; ids survive the float round trip
    assert all(d == "torch.float32" for d in z["C_dtype"])
    for i in range(8):
        for col, key in (("rows_seq_d1", "C/seq_d1"), ("rows_seq_d2", "C/seq_d2")):
            seq = json.loads(str(z[col][i]))
            dom = int(z["rows_domain_id"][i])
            own = (dom == 0 and col == "rows_seq_d1") or (dom == 1 and col == "rows_seq_d2")
            if own:      # positive = last item, removed from the sequence with all its duplicates (dataset_seq.py:189-195)
                item = seq[-1]
                assert int(z["C/i_node"][i]) == item
                seq = [s for s in seq[:-1] if s != item]
            enc, _ = orc.seq_padding(seq, seq_len + 1, 7, pad_id)
            got = orc.wire_to_long(torch.from_numpy(z[key][i]))
            assert got.tolist() == enc
    assert z["C/label"].shape == (8, 2) and z["C/neg_samples"].shape == (8, 1)


def test_g8_metrics():
    z = np.load(os.path.join(GOLDEN, "g8_metrics.npz"))
    got = np.array(orc.get_sample_scores(z["pred"].copy()))
    assert np.allclose(got, z["scores"], rtol=0, atol=1e-12)


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    out = orc.philox4x32(np.array([[0, 0, 0, 0]], dtype=np.uint32), (0, 0))
    assert [hex(int(x)) for x in out[0]] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    out = orc.philox4x32(np.array([[0xFFFFFFFF] * 4], dtype=np.uint32), (0xFFFFFFFF, 0xFFFFFFFF))
    assert [hex(int(x)) for x in out[0]] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    out = orc.philox4x32(np.array([[0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]], dtype=np.uint32), (0xA4093822, 0x299F31D0))
    assert [hex(int(x)) for x in out[0]] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]
    keep = orc.philox_keep_flat(100000, seed=7, site=3, step=1, p=0.5)
    assert 0.49 < keep.mean() < 0.51


def test_g9_inter_inner_comp_forward():
    """InterComp / InnerComp (next-1): the compute-once restatement against the reference's [bs, b, n, d] formulation."""
    z, P, *_ = load("g9_comp.npz")
    a, b = torch.from_numpy(z["a"]), torch.from_numpy(z["b"])
    assert float((orc.inter_comp(a, b, P, "itc", 0.15) - torch.from_numpy(z["inter"])).abs().max()) < 1e-6
    assert float((orc.inner_comp(a, P, "inc", 0.15) - torch.from_numpy(z["inner"])).abs().max()) < 1e-6


def test_g10_sasrec_itc_grads():
    """SASRec(isItC=True) -- the configuration run.sh trains: logits, loss and every gradient (trans_nn / trans_bs included)."""
    z, P, B, G, _ = load("g10_sasrec_itc.npz")
    taps = {}
    orc.sasrec_forward(P, B["i_node"], B["neg_samples"], B["seq_d1"], B["seq_d2"], None, taps, isItC=True, threshold2=float(z["threshold2"]))
    assert np.array_equal(taps["itc_d1"]["gate"].numpy().astype(bool), z["gate"])
    assert np.array_equal(taps["itc_d2"]["gate"].numpy().astype(bool), z["gate"])          # max over all (a, c) pairs is symmetric
    assert set(G) == set(orc.sasrec_param_shapes(P["item_emb_layer.emb_item.weight"].shape[0], 64, 20, 16, itc_bs=6))
    check_grads("sasrec", z, P, B, G, None, isItC=True, threshold2=float(z["threshold2"]))


def test_g13_bert4rec_dr_outputs_losses_grads():
    """BERT4Rec(isDR=True): six outputs, the three losses and the gradients of both objectives against the reference."""
    z, _, B, *_ = load("g13_bert4rec_dr.npz")
    P = bert_params(z, dr=True)
    batch = dict(B)
    batch["label"] = torch.from_numpy(z["labels"])
    batch["ob_label"] = torch.from_numpy(z["ob_label"])
    for which, pre in (("e", "GE/"), ("r", "GR/")):
        info, outs, grads = orc.dr_loss_and_grads(P, batch, which, dr_e_w=float(z["dr_e_w"]), model="bert4rec")
        for o, name in zip(outs, ("p1", "p2", "ips1", "ips2", "g1", "g2")):
            assert rel_err(o, z[name]) < 1e-6, name
        close = lambda a, b: abs(float(a) - float(b)) < 2e-6 * max(1.0, abs(float(b)))      # noqa: E731
        assert close(info["loss_cls"], z["loss_cls"]) and close(info["loss_dr_e"], z["loss_dr_e"])
        if which == "r":
            assert close(info["loss"], z["loss_dr_r"])
        G = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
        assert len(G) > 20 and set(G) <= set(P)
        for k, g in G.items():
            if k.endswith("linear_layers.1.bias"):      # the key bias: analytically zero (softmax shift invariance), rounding noise only
                continue
            assert rel_err(grads[k], g) < 5e-5 or float((grads[k] - g).abs().max()) < 1e-8, (which, k)


@pytest.mark.parametrize("kind", ["inc", "itc"])
def test_g14_g15_bert4rec_comp_grads(kind):
    """BERT4Rec(isInC=True) / (isItC=True): the comp module in FRONT of the encoders (model_seq.py:283-294), 2T tokens, the key
    mask tiled twice: gates, logits, loss and the stored gradients against the reference."""
    z, _, B, G, _ = load("g14_bert4rec_inc.npz" if kind == "inc" else "g15_bert4rec_itc.npz")
    bs = B["seq_d1"].shape[0]
    P = bert_params(z, inc_bs=bs if kind == "inc" else 0, itc_bs=bs if kind == "itc" else 0)
    kw = dict(isInC=True, threshold1=float(z["threshold"])) if kind == "inc" else dict(isItC=True, threshold2=float(z["threshold"]))
    taps = {}
    orc.bert4rec_forward(P, B["i_node"], B["neg_samples"], B["seq_d1"], B["seq_d2"], None, taps=taps, **kw)
    assert np.array_equal(taps[f"{kind}_d1"]["gate"].numpy().astype(bool), z["gate_d1"])
    assert np.array_equal(taps[f"{kind}_d2"]["gate"].numpy().astype(bool), z["gate_d2"])
    batch = dict(B)
    batch["label"] = torch.from_numpy(z["labels"])
    loss, (p1, p2), grads = orc.loss_and_grads("bert4rec", P, batch, None, **kw)
    assert rel_err(p1, z["p1"]) < 1e-6 and rel_err(p2, z["p2"]) < 1e-6
    assert abs(float(loss) - float(z["loss"])) < 2e-6
    assert len(G) > 20 and set(G) <= set(P) and any(k.startswith(kind + "_d") for k in G)
    for k, g in G.items():
        if k.endswith("linear_layers.1.bias"):      # the key bias: analytically zero (softmax shift invariance), rounding noise only
            continue
        assert rel_err(grads[k], g) < 5e-5 or float((grads[k] - g).abs().max()) < 1e-8, k
    with pytest.raises(ValueError):
        orc.bert4rec_forward(P, B["i_node"], B["neg_samples"], B["seq_d1"], B["seq_d2"], None, isInC=True, isItC=True)


def test_g12_sasrec_inc_grads():
    """SASRec(isInC=True): InnerComp on the gathered rows, encoders over 2T tokens (2T-row pos_emb): logits, loss, every gradient."""
    z, P, B, G, _ = load("g12_sasrec_inc.npz")
    taps = {}
    ts1 = float(z["threshold1"])
    orc.sasrec_forward(P, B["i_node"], B["neg_samples"], B["seq_d1"], B["seq_d2"], None, taps, isInC=True, threshold1=ts1)
    assert np.array_equal(taps["inc_d1"]["gate"].numpy().astype(bool), z["gate_d1"])
    assert np.array_equal(taps["inc_d2"]["gate"].numpy().astype(bool), z["gate_d2"])
    assert set(G) == set(orc.sasrec_param_shapes(P["item_emb_layer.emb_item.weight"].shape[0], 64, 20, 16, inc_bs=6))
    assert P["sac1.pos_emb.weight"].shape == (40, 64)
    check_grads("sasrec", z, P, B, G, None, isInC=True, threshold1=ts1)


def test_g11_sasrec_dr_outputs_losses_grads():
    """SASRec(isDR=True, isItC=True) -- what run.sh launches through train_sr_dr.py: six outputs, the three losses and the
    gradients of both objectives (loss_cls + dr_e_w * loss_dr_e for optimizer, loss_dr_r for optimizer2)."""
    z, P, B, *_ = load("g11_sasrec_dr.npz")
    batch = dict(B)
    batch["label"] = torch.from_numpy(z["labels"])
    batch["ob_label"] = torch.from_numpy(z["ob_label"])
    kw = dict(isItC=True, threshold2=float(z["threshold2"]))
    assert set(P) == set(orc.sasrec_param_shapes(P["item_emb_layer.emb_item.weight"].shape[0], 64, 20, 16, itc_bs=6, dr=True))
    for which, pre in (("e", "GE/"), ("r", "GR/")):
        info, outs, grads = orc.dr_loss_and_grads(P, batch, which, dr_e_w=float(z["dr_e_w"]), **kw)
        for o, name in zip(outs, ("p1", "p2", "ips1", "ips2", "g1", "g2")):
            assert rel_err(o, z[name]) < 1e-6, name
        assert abs(float(info["loss_cls"]) - float(z["loss_cls"])) < 1e-6 and abs(float(info["loss_dr_e"]) - float(z["loss_dr_e"])) < 1e-6
        if which == "r":
            assert abs(float(info["loss"]) - float(z["loss_dr_r"])) < 1e-6
        G = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
        assert set(G) == set(P)
        for k, g in G.items():
            assert rel_err(grads[k], g) < 2e-5 or float((grads[k] - g).abs().max()) < 1e-8, (which, k)
