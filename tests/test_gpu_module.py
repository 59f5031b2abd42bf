"""GPU tests of the drop-in boundary: reference class names / constructor signatures / state_dict keys,
the reference training loop with torch.optim.Adam, the fused train_step, and the train_sr.py CLI."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import amid_oracle as orc

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    P = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("P/")}
    B = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("B/")}
    return z, P, B


def relmax(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def make_model(P, T, lr=5e-4, seed=0):
    from amid_amd.model_seq import SASRec
    n_rows, D = P["item_emb_layer.emb_item.weight"].shape
    hid = P["predictModule.fc.0.weight"].shape[0]
    m = SASRec(10, D, n_rows, D, T, hid, 4, False, False, 0.5, 0.5, lr=lr, seed=seed).cuda()     # reference ctor order, .cuda() as train_sr.py:461
    m.load_state_dict(P)
    return m


def test_state_dict_keys_and_shapes_match_reference():
    z, P, B = load_golden("g3_sasrec_eval.npz")
    m = make_model(P, 50)
    sd = m.state_dict()
    assert list(sd.keys()) == list(P.keys())            # same names, same order as the reference's state_dict
    for k in P:
        assert tuple(sd[k].shape) == tuple(P[k].shape), k
        assert torch.equal(sd[k].cpu(), P[k]), k
    assert sum(p.numel() for p in m.parameters()) == sum(v.numel() for v in P.values())


def test_forward_matches_reference_golden_and_squeezes():
    z, P, B = load_golden("g3_sasrec_eval_neg9.npz")
    m = make_model(P, 50).eval()
    cu = {k: v.cuda() for k, v in B.items()}
    with torch.no_grad():
        p1, p2 = m(None, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], None, None, False)
    assert tuple(p1.shape) == z["p1"].shape
    assert relmax(p1, z["p1"]) < 1e-4 and relmax(p2, z["p2"]) < 1e-4


def test_reference_training_loop_with_torch_adam():
    """optimizer.zero_grad(); loss.backward(); optimizer.step() exactly as train_sr.py:203-215."""
    T, Bn, D, hid, n_items, seed = 20, 8, 64, 16, 150, 21
    P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=2)
    m = make_model(P, T, seed=seed)
    m.train()
    optimizer = torch.optim.Adam(m.parameters(), lr=1e-3)
    criterion_cls = torch.nn.BCELoss(reduction="none")
    Po = {k: v.clone() for k, v in P.items()}
    opt_o = orc.DenseAdam(Po, lr=1e-3)
    for t in range(1, 4):
        batch = orc.synthetic_batch(Bn, T, n_items - 1, pad_id=n_items - 1, neg=1, seed=300 + t)
        cu = {k: v.cuda() for k, v in batch.items()}
        predict_d1, predict_d2 = m(None, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], None, None)
        mask_d1 = (1 - cu["domain_id"]).unsqueeze(1)
        mask_d2 = cu["domain_id"].unsqueeze(1)
        loss = torch.mean(criterion_cls(predict_d1, cu["label"]) * mask_d1 + criterion_cls(predict_d2, cu["label"]) * mask_d2)
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        loss_o = orc.train_step("sasrec", Po, opt_o, batch, orc.philox_masks_sasrec(Bn, T, D, seed=seed, step=t))
        assert abs(loss.item() - loss_o) < 5e-5, t
    sd = m.state_dict()
    for k, v in Po.items():
        d = (sd[k].cpu() - v).abs()
        if k.endswith("in_proj_bias"):
            n = v.numel() // 3
            d = torch.cat((d[:n], d[2 * n:]))
        assert float(d.max()) < 1e-4, k


def test_fused_train_step_matches_oracle_and_graph_is_reused():
    T, Bn, D, hid, n_items, seed = 20, 8, 64, 16, 150, 5
    P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=3)
    m = make_model(P, T, lr=1e-3, seed=seed)
    m.train()
    Po = {k: v.clone() for k, v in P.items()}
    opt_o = orc.DenseAdam(Po, lr=1e-3)
    for t in range(1, 5):
        batch = orc.synthetic_batch(Bn, T, n_items - 1, pad_id=n_items - 1, neg=1, seed=400 + t)
        cu = {k: v.cuda() for k, v in batch.items()}
        loss = m.train_step(cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"])
        loss_o = orc.train_step("sasrec", Po, opt_o, batch, orc.philox_masks_sasrec(Bn, T, D, seed=seed, step=t))
        assert abs(loss.item() - loss_o) < 5e-5, t
    assert m._last_plan.graph is not None
    sd = m.state_dict()                  # flushes lazily-updated rows
    assert relmax(sd["item_emb_layer.emb_item.weight"], Po["item_emb_layer.emb_item.weight"]) < 1e-5
    assert relmax(sd["predictModule.fc.0.weight"], Po["predictModule.fc.0.weight"]) < 1e-4
    # eval forward after fused training sees the flushed table
    m.eval()
    batch = orc.synthetic_batch(Bn, T, n_items - 1, pad_id=n_items - 1, neg=3, seed=9)
    cu = {k: v.cuda() for k, v in batch.items()}
    with torch.no_grad():
        p1, _ = m(None, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], None, None, False)
    q1, _ = orc.sasrec_forward(Po, batch["i_node"], batch["neg_samples"], batch["seq_d1"], batch["seq_d2"])
    assert relmax(p1, q1) < 1e-4


def test_out_of_range_item_raises_like_nn_embedding():
    P = orc.random_params(orc.sasrec_param_shapes(50, 64, 10, 8), seed=1)
    m = make_model(P, 10).eval()
    batch = orc.synthetic_batch(4, 10, 49, pad_id=49, neg=1, seed=1)
    batch["seq_d1"][0, 3] = 50
    cu = {k: v.cuda() for k, v in batch.items()}
    with pytest.raises(IndexError):
        with torch.no_grad():
            m(None, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], None, None, False)


def test_embitemlayer_forward_backward():
    from amid_amd.model_seq import embItemLayerEnhance
    z = np.load(os.path.join(GOLDEN, "g1_gather.npz"))
    emb = embItemLayerEnhance(z["table"].shape[0], z["table"].shape[1])
    with torch.no_grad():
        emb.emb_item.weight.copy_(torch.from_numpy(z["table"]))
    idx = torch.from_numpy(z["idx"]).cuda()
    out = emb(idx)
    assert np.array_equal(out.detach().cpu().numpy(), z["rows"])
    # backward (HIP sort-unique + segment reduce) on a width the kernels are built for
    emb = embItemLayerEnhance(500, 64)
    idx = torch.randint(0, 500, (6, 9)).cuda()
    idx[:, :5] = 499
    out = emb(idx)
    g = torch.randn_like(out)
    out.backward(g)
    want = torch.zeros(500, 64, dtype=torch.float64)
    want.index_add_(0, idx.reshape(-1).cpu(), g.reshape(-1, 64).double().cpu())
    assert float((emb.emb_item.weight.grad.cpu().double() - want).abs().max()) < 1e-5


def test_log2feats_standalone_matches_golden():
    from amid_amd.model_seq import Log2feats
    z = np.load(os.path.join(GOLDEN, "g2_log2feats_d128.npz"))
    enc = Log2feats(10, 128, 100, 128, 50, 16)
    sd = {k[len("P/sac1."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("P/sac1.")}
    assert set(sd) == set(enc.state_dict())
    enc.load_state_dict(sd)
    y = enc(torch.from_numpy(z["x"]).cuda())
    assert float((y.cpu() - torch.from_numpy(z["y"])).abs().max()) < 2e-5


def test_predict_module_standalone():
    from amid_amd.model_seq import predictModule
    pm = predictModule(64, 16)
    g = torch.Generator().manual_seed(0)
    u1, u2, it = torch.randn(5, 64, generator=g), torch.randn(5, 64, generator=g), torch.randn(5, 70, 64, generator=g)
    p1, p2 = pm(u1.cuda(), u2.cuda(), it.cuda())
    P = {"predictModule." + k: v.detach().cpu() for k, v in pm.state_dict().items()}
    q1, q2 = orc.predict_module(u1, u2, it, P)
    assert relmax(p1, q1) < 1e-5 and relmax(p2, q2) < 1e-5


def test_unbuilt_models_fail_loudly():
    from amid_amd import model_seq
    with pytest.raises(NotImplementedError):
        model_seq.GRU4Rec(10, 128, 100, 128, 20, 32, 4, False, False, 0.5, 0.5)
    with pytest.raises(ValueError):                      # both comp modules on BERT4Rec: the reference itself fails (model_seq.py:294)
        model_seq.BERT4Rec(10, 128, 100, 128, 20, 32, 4, True, True, 0.5, 0.5)
    with pytest.raises(ValueError):                      # the reference hard-codes hidden size 128 (model_seq.py:264-267)
        model_seq.BERT4Rec(10, 64, 100, 64, 20, 32, 4, False, False, 0.5, 0.5)


def _write_csv(path, n, rng, lo1, hi1, lo2, hi2, ob_label=False):
    rows = ["user_id,seq_d1,seq_d2,domain_id" + (",ob_label" if ob_label else "")]
    for u in range(n):
        dom = int(rng.random() < 0.5)
        l1 = int(rng.integers(1 if dom == 0 else 0, 9))
        l2 = int(rng.integers(1 if dom == 1 else 0, 9))
        s1 = [int(x) for x in rng.integers(lo1, hi1, l1)]
        s2 = [int(x) for x in rng.integers(lo2, hi2, l2)]
        rows.append(f'{u},"{json.dumps(s1)}","{json.dumps(s2)}",{dom}' + (f",{int(rng.random() < 0.6)}" if ob_label else ""))
    with open(path, "w") as f:
        f.write("\n".join(rows) + "\n")


def test_sasrec_itc_module_surface():
    """SASRec(isItC=True): the reference's extra state_dict keys, eval forward vs the oracle, a batch of the wrong size refused."""
    from amid_amd.model_seq import SASRec
    D, T, hid, n_items, bs = 64, 20, 16, 200, 8
    m = SASRec(10, D, n_items, D, T, hid, bs, False, True, 0.5, 0.15).cuda()
    want = set(orc.sasrec_param_shapes(n_items, D, T, hid, itc_bs=bs))
    assert set(m.state_dict().keys()) == want
    assert tuple(m.itc_d2.trans_bs.weight.shape) == (1, bs)
    P = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    batch = orc.synthetic_batch(bs, T, n_items - 1, pad_id=n_items - 1, neg=3, seed=2)
    batch["seq_d1"] = torch.randint(1, n_items - 1, (bs, T), generator=g)
    batch["seq_d2"] = torch.randint(1, n_items - 1, (bs, T), generator=g)
    cu = {k: v.cuda() for k, v in batch.items()}
    m.eval()
    with torch.no_grad():
        p1, p2 = m(None, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], None, None, False)
    o1, o2 = orc.sasrec_forward(P, batch["i_node"], batch["neg_samples"], batch["seq_d1"], batch["seq_d2"], isItC=True, threshold2=0.15)
    assert relmax(p1, o1.squeeze()) < 3e-5 and relmax(p2, o2.squeeze()) < 3e-5
    with pytest.raises(ValueError):
        m(None, cu["i_node"][:4], cu["neg_samples"][:4], cu["seq_d1"][:4], cu["seq_d2"][:4], None, None, False)


def test_sasrec_inc_module_surface_and_reference_loop():
    """SASRec(isInC=True): the reference's extra state_dict keys (inc_d*, 2T-row pos_emb), eval forward vs the oracle, then the
    reference's own loop shape (loss.backward(), torch.optim.Adam) for one step against the oracle's dense Adam."""
    from amid_amd.model_seq import SASRec
    D, T, hid, n_items, bs, ts1 = 64, 20, 16, 200, 8, 0.13
    m = SASRec(10, D, n_items, D, T, hid, bs, True, False, ts1, 0.5).cuda()
    assert set(m.state_dict().keys()) == set(orc.sasrec_param_shapes(n_items, D, T, hid, inc_bs=bs))
    assert tuple(m.inc_d1.trans_bs.weight.shape) == (1, bs) and tuple(m.sac2.pos_emb.weight.shape) == (2 * T, D)
    with torch.no_grad():
        m.item_emb_layer.emb_item.weight.mul_(0.3)            # self pair-max scores close enough for a non-trivial gate
    P = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    batch = orc.synthetic_batch(bs, T, n_items - 1, pad_id=n_items - 1, neg=3, seed=2)
    batch["seq_d1"] = torch.randint(1, n_items - 1, (bs, T), generator=g)
    batch["seq_d2"] = torch.randint(1, n_items - 1, (bs, T), generator=g)
    cu = {k: v.cuda() for k, v in batch.items()}
    m.eval()
    with torch.no_grad():
        p1, p2 = m(None, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], None, None, False)
    taps = {}
    o1, o2 = orc.sasrec_forward(P, batch["i_node"], batch["neg_samples"], batch["seq_d1"], batch["seq_d2"], None, taps, isInC=True, threshold1=ts1)
    assert 0 < int(taps["inc_d1"]["gate"].sum()) < bs and taps["inc_d1"]["margin"] > 1e-3 and taps["inc_d2"]["margin"] > 1e-3
    assert relmax(p1, o1.squeeze()) < 3e-5 and relmax(p2, o2.squeeze()) < 3e-5
    with pytest.raises(ValueError):
        m(None, cu["i_node"][:4], cu["neg_samples"][:4], cu["seq_d1"][:4], cu["seq_d2"][:4], None, None, False)
    # reference loop, dropout off on both sides
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    p1, p2 = m(None, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], None, None)
    crit = torch.nn.BCELoss(reduction="none")
    y = cu["label"]
    loss = torch.mean(crit(p1, y) * (1 - cu["domain_id"]).unsqueeze(1).float() + crit(p2, y) * cu["domain_id"].unsqueeze(1).float())
    loss.backward()
    opt.step()
    Po = {k: v.clone() for k, v in P.items()}
    oo = orc.DenseAdam(Po, lr=1e-3)
    lo = orc.train_step("sasrec", Po, oo, batch, None, isInC=True, threshold1=ts1)
    assert abs(float(loss.detach()) - lo) < 1e-5
    sd = m.state_dict()
    for k, v in Po.items():
        d = (sd[k].cpu() - v).abs()
        if k.endswith("in_proj_bias"):
            Dd = v.numel() // 3
            d = torch.cat((d[:Dd], d[2 * Dd:]))
        assert float(d.max()) < 2e-4, (k, float(d.max()))


@pytest.mark.parametrize("model,emb,extra", [("sasrec", "64", []), ("bert4rec", "128", []), ("sasrec", "64", ["--isItC", "True", "--ts2", "0.4"]),
                                             ("sasrec", "128", ["--dtype", "bf16"]), ("sasrec", "64", ["--isInC", "True", "--ts1", "0.02"]),
                                             ("sasrec", "64", ["--isInC", "True", "--ts1", "0.02", "--isItC", "True", "--ts2", "0.02"]),
                                             ("bert4rec", "128", ["--isItC", "True", "--ts2", "0.02"]),
                                             ("bert4rec", "128", ["--isInC", "True", "--ts1", "0.02"])])
def test_train_sr_cli_end_to_end(tmp_path, model, emb, extra):
    """The reference's command line on a synthetic CSV pair with the reference's column layout."""
    from amid_amd.train_sr import main
    rng = np.random.default_rng(0)
    root = tmp_path / "amazon_dataset"
    root.mkdir()
    _write_csv(root / "toy_train75.csv", 300, rng, 1, 400, 400, 900)
    _write_csv(root / "toy_test.csv", 80, rng, 1, 400, 400, 900)
    summary = main(["--data_root", str(tmp_path), "-ds", "amazon", "-dm", "toy", "--overlap_ratio", "0.75", "--model", model,
                    "--bs", "32", "--seq_len", "20", "--emb_dim", emb, "--hid_dim", "16", "--epoch", "2", "--neg_nums", "19",
                    "--seeds", "1", "-md", str(tmp_path / "model")] + extra)
    assert len(summary) == 1
    best = summary[0]
    assert ("d1", "HR@10") in best and ("d2", "MRR") in best
    assert all(0.0 <= v <= 1.0 for v in best.values())
    assert (tmp_path / "model" / "log0.txt").exists()


@pytest.mark.parametrize("variant", ["sasrec", "dr"])
def test_epoch_pool_equals_per_step_batches(tmp_path, variant):
    """Two epochs fed from the HBM-resident epoch pool (DeviceBatches.epoch_tensors -> begin_epoch_pool / pool_step; the second
    epoch refills the pool under the captured graph) leave exactly the parameters that per-step batches leave; isDR: both loops,
    each with its own pool, Adam state and objective."""
    from amid_amd.dataset_seq import DeviceBatches, DualDomainSeqDataset
    from amid_amd.model_seq import SASRec
    rng = np.random.default_rng(7)
    root = tmp_path / "amazon_dataset"
    root.mkdir()
    dr = variant == "dr"
    _write_csv(root / "toy_train75.csv", 150, rng, 1, 300, 300, 700, ob_label=dr)
    _write_csv(root / "toy_b.csv", 110, rng, 1, 300, 300, 700, ob_label=dr)

    def run(pooled):
        dss = [DualDomainSeqDataset(seq_len=20, isTrain=True, neg_nums=9, long_length=7, pad_id=1001, seed=3, csv_path=str(root / f))
               for f in (("toy_train75.csv", "toy_b.csv") if dr else ("toy_train75.csv",))]
        loaders = [DeviceBatches(d, 16, shuffle=True, device="cuda:0", seed=3 + k) for k, d in enumerate(dss)]
        m = SASRec(10, 64, 2000, 64, 20, 16, 16, False, False, 0.5, 0.5, isDR=dr, lr=1e-3, seed=1).cuda()
        losses = []
        for _ in range(2):
            for k, ld in enumerate(loaders):
                if dr:
                    m.engine.select_optimizer(k, lr=1e-3 * (1.0 if k == 0 else 0.5))
                if pooled:
                    n = m.begin_epoch_pool(ld.epoch_tensors(), dr_objective=k)
                    assert n == len(ld)
                    for _i in range(n):
                        out = m.pool_step(dr_objective=k)
                    m.end_epoch_pool()
                else:
                    for b in ld:
                        out = m.train_step(b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"], b["label"], b["domain_id"],
                                           ob_label=b["ob_label"] if dr else None, dr_objective=k)
                m.engine.sync()
                losses.append(out.detach().cpu().clone())
        return losses, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}

    l0, s0 = run(False)
    l1, s1 = run(True)
    for a, b in zip(l0, l1):
        assert torch.equal(a, b)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def test_device_negative_sampling(tmp_path):
    """next-3: negatives drawn on the device obey the reference's rule (dataset_seq.py:188/:198): k distinct items of the row's
    own domain pool, none of them in the row's own sequence; fresh draws every epoch; roughly uniform."""
    from amid_amd.dataset_seq import DeviceBatches, DualDomainSeqDataset
    rng = np.random.default_rng(5)
    root = tmp_path / "amazon_dataset"
    root.mkdir()
    _write_csv(root / "toy_test.csv", 120, rng, 1, 300, 300, 700)
    for is_train, k in ((True, 1), (False, 99)):
        ds = DualDomainSeqDataset(seq_len=20, isTrain=is_train, neg_nums=99, long_length=7, pad_id=1001, seed=3, csv_path=str(root / "toy_test.csv"))
        db = DeviceBatches(ds, 16, shuffle=False, device="cuda:0", seed=3)
        a, b = db.sample_negatives().cpu().numpy(), db.sample_negatives().cpu().numpy()
        assert a.shape == (len(ds), k) and not np.array_equal(a, b)
        for r in range(len(ds)):
            pool = ds.pool[int(ds.domain_id[r] != 0)]
            assert np.isin(a[r], pool).all() and not np.isin(a[r], ds.own_items[r]).any() and len(set(a[r].tolist())) == k
    # uniformity: over many epochs every eligible item of row 0 shows up about equally often
    ds = DualDomainSeqDataset(seq_len=20, isTrain=False, neg_nums=50, long_length=7, pad_id=1001, seed=3, csv_path=str(root / "toy_test.csv"))
    db = DeviceBatches(ds, 16, shuffle=False, device="cuda:0", seed=9)
    cnt = {}
    for _ in range(200):
        for v in db.sample_negatives()[0].tolist():
            cnt[v] = cnt.get(v, 0) + 1
    pool = ds.pool[int(ds.domain_id[0] != 0)]
    elig = len(pool) - len(ds.own_items[0])
    freq = np.array([cnt.get(int(v), 0) for v in pool if v not in set(ds.own_items[0].tolist())], dtype=np.float64)
    assert len(freq) == elig and abs(freq.sum() - 200 * 50) < 1e-9
    expect = 200 * 50 / elig
    assert freq.min() > 0.3 * expect and freq.max() < 2.0 * expect


def test_train_sr_dr_cli_end_to_end(tmp_path):
    """run.sh's command line (train_sr_dr.py, --isItC True, doubly-robust heads, two optimizers) on synthetic CSVs with the
    reference's column layout (the second loader's CSV carries ob_label)."""
    from amid_amd.train_sr_dr import main
    rng = np.random.default_rng(2)
    root = tmp_path / "mybank_dataset"
    root.mkdir()
    _write_csv(root / "toy_train25.csv", 200, rng, 1, 400, 400, 900)
    _write_csv(root / "toy_train25_DR.csv", 160, rng, 1, 400, 400, 900, ob_label=True)
    _write_csv(root / "toy_test.csv", 64, rng, 1, 400, 400, 900)
    summary = main(["--data_root", str(tmp_path), "-ds", "mybank", "-dm", "toy", "--overlap_ratio", "0.25", "--model", "sasrec", "--overlap", "True",
                    "--isItC", "True", "--ts2", "0.4", "--neg_nums", "19", "--lr2", "0.01", "--dr_e_w", "0.01", "--bs", "32", "--seq_len", "20",
                    "--emb_dim", "64", "--hid_dim", "16", "--epoch", "2", "--seeds", "1", "-md", str(tmp_path / "model")])
    best = summary[0]
    assert ("d1", "HR@10") in best and ("d2_ov", "MRR") in best
    assert all((0.0 <= v <= 1.0) or np.isnan(v) for v in best.values())
    log = (tmp_path / "model" / "log0.txt").read_text()
    assert "train loss_dr_r" in log and "dr_e loss" in log


def test_train_sr_dr_cli_bert4rec(tmp_path):
    """train_sr_dr.py with --model bert4rec: the doubly-robust heads on the BERT4Rec encoders (model_seq.py:268-271)."""
    from amid_amd.train_sr_dr import main
    rng = np.random.default_rng(3)
    root = tmp_path / "mybank_dataset"
    root.mkdir()
    _write_csv(root / "toy_train25.csv", 160, rng, 1, 400, 400, 900)
    _write_csv(root / "toy_train25_DR.csv", 128, rng, 1, 400, 400, 900, ob_label=True)
    _write_csv(root / "toy_test.csv", 64, rng, 1, 400, 400, 900)
    summary = main(["--data_root", str(tmp_path), "-ds", "mybank", "-dm", "toy", "--overlap_ratio", "0.25", "--model", "bert4rec", "--overlap", "True",
                    "--neg_nums", "19", "--lr2", "0.01", "--dr_e_w", "0.01", "--bs", "32", "--seq_len", "20", "--emb_dim", "128",
                    "--hid_dim", "16", "--epoch", "2", "--seeds", "1", "-md", str(tmp_path / "model")])
    best = summary[0]
    assert ("d1", "HR@10") in best and all((0.0 <= v <= 1.0) or np.isnan(v) for v in best.values())
    log = (tmp_path / "model" / "log0.txt").read_text()
    assert "train loss_dr_r" in log and "dr_e loss" in log


def test_sasrec_dr_module_reference_loop():
    """SASRec(isDR=True) through the reference's own loop shape: six outputs, loss from train_sr_dr.py:216-221 written with torch
    ops, loss.backward(), torch.optim.Adam -- one step must match the oracle's dense Adam."""
    from amid_amd.model_seq import SASRec
    D, T, hid, n_items, bs, w = 64, 20, 16, 200, 8, 0.1
    m = SASRec(10, D, n_items, D, T, hid, bs, False, False, 0.5, 0.5, isDR=True).cuda()
    assert set(m.state_dict().keys()) == set(orc.sasrec_param_shapes(n_items, D, T, hid, dr=True))
    P = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    batch = orc.synthetic_batch(bs, T, n_items - 1, pad_id=n_items - 1, neg=1, seed=4)
    cu = {k: v.cuda() for k, v in batch.items()}
    m.eval()                                   # dropout off on both sides
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    p1, p2, i1, i2, g1, g2 = m(None, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], None, None)
    crit = torch.nn.BCELoss(reduction="none")
    y = cu["label"]
    m1 = (1 - cu["domain_id"]).unsqueeze(1).float()
    m2 = cu["domain_id"].unsqueeze(1).float()
    loss_cls = torch.mean(crit(p1, y) * m1 + crit(p2, y) * m2)
    loss_dr_e = torch.mean((crit(p1, y) - g1) ** 2 / i1 * m1 + (crit(p2, y) - g2) ** 2 / i2 * m2)
    (loss_cls + w * loss_dr_e).backward()
    opt.step()
    Po = {k: v.clone() for k, v in P.items()}
    info, outs, grads = orc.dr_loss_and_grads(Po, batch, "e", None, dr_e_w=w)
    orc.DenseAdam(Po, lr=1e-3).step(Po, grads)
    assert abs(float(loss_cls.detach()) - float(info["loss_cls"])) < 5e-5 and abs(float(loss_dr_e.detach()) - float(info["loss_dr_e"])) < 5e-5
    sd = m.state_dict()
    for k, v in Po.items():
        dlt = (sd[k].cpu() - v).abs()
        if k.endswith("in_proj_bias"):
            Dd = v.numel() // 3
            dlt = torch.cat((dlt[:Dd], dlt[2 * Dd:]))
        assert float(dlt.max()) < 2e-4, k


def test_standalone_comp_modules_golden_and_autograd():
    """model_seq.InnerComp / InterComp as modules of their own (model_seq.py:450-497): forward against the reference's outputs
    (g9, its [bs, b, n, d] formulation), backward against the oracle's autograd, the batch-size contract."""
    from amid_amd import model_seq as ms
    from oracle import amid_oracle as orc
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "g9_comp.npz"))
    P = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("P/")}
    a, b = torch.from_numpy(z["a"]), torch.from_numpy(z["b"])
    bs, T, D = a.shape
    for kind, cls, want in (("itc", ms.InterComp, z["inter"]), ("inc", ms.InnerComp, z["inner"])):
        m = cls(D, bs, 0.15).cuda()
        m.load_state_dict({k[len(kind) + 1:]: v for k, v in P.items() if k.startswith(kind + ".")})
        xa, xb = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        out = m(xa, xb) if kind == "itc" else m(xa)
        assert out.shape == (bs, 2 * T, D)
        assert float((out.detach().cpu() - torch.from_numpy(want)).abs().max()) < 1e-5
        w = torch.randn(bs, 2 * T, D, generator=torch.Generator().manual_seed(5))
        (out * w.cuda()).sum().backward()
        leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        ca, cb = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        ref = orc.inter_comp(ca, cb, leaves, "itc", 0.15) if kind == "itc" else orc.inner_comp(ca, leaves, "inc", 0.15)
        (ref * w).sum().backward()
        rel = lambda g, r: float((g.cpu() - r).abs().max() / (r.abs().max() + 1e-30))      # noqa: E731
        assert rel(xa.grad, ca.grad) < 1e-4
        if kind == "itc":
            assert rel(xb.grad, cb.grad) < 1e-4
        for n in ("trans_nn.weight", "trans_nn.bias", "trans_bs.weight", "trans_bs.bias"):
            assert rel(m.get_parameter(n).grad, leaves[f"{kind}.{n}"].grad) < 1e-4, (kind, n)
        with pytest.raises(ValueError):
            m(xa[:2], xb[:2]) if kind == "itc" else m(xa[:2])
