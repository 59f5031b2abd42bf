"""The evaluation batch in four launches (SasrecEngine.enqueue_eval: test(), /root/reference/train_sr.py:31-128) against the launches it
replaces (enqueue_forward over both domains + amid_positive_rank_f32), bit for bit, against the oracle, and through train_sr.test()."""
import json

import numpy as np
import pytest
import torch

from oracle import amid_oracle as orc

pytestmark = pytest.mark.gpu
FIX = 1e-7


def make_engine(P, n_items, D, T, hid, compute="f32"):
    from amid_amd.engine import SasrecEngine
    eng = SasrecEngine(n_items, D, T, hid, device="cuda:0", lr=1e-3, seed=5, compute=compute)
    eng.load_state_dict(P)
    return eng


def eval_batch(B, T, n_items, NI, seed, dup_positive=False):
    b = orc.synthetic_batch(B, T, n_items - 1, pad_id=n_items - 1, neg=NI - 1, seed=seed)
    if dup_positive:                    # the positive's own id among the negatives: an exact tie, counted against the positive only with fix_value
        b["neg_samples"][::2, 3] = b["i_node"][::2]
    b["label"] = torch.zeros(B, NI)
    b["label"][:, 0] = 1.0
    return b


def old_path(eng, pl, cu):
    """What bench.py / test() ran before round 6: the eval forward over every sequence with the candidates gathered by K1, then the rank kernel."""
    from amid_amd.utils import device_positive_ranks
    eng.load_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"])
    eng.enqueue_prepare(pl, sparse=False)
    eng.enqueue_forward(pl, train=False, with_loss=False)
    eng.sync()
    torch.cuda.current_stream().wait_stream(eng.stream)
    p1, p2 = pl.p1.clone(), pl.p2.clone()
    r = device_positive_ranks(p1, p2, cu["domain_id"], FIX)
    r0 = device_positive_ranks(p1, p2, cu["domain_id"], 0.0)
    torch.cuda.synchronize()
    own = torch.where(cu["domain_id"][:, None] != 0, p2, p1)
    return own, r, r0


@pytest.mark.parametrize("D,hid,T,B,NI,compute", [(128, 32, 50, 48, 1000, "f32"), (128, 32, 50, 256, 200, "f32"), (128, 32, 33, 7, 5, "f32"),
                                                  (64, 16, 20, 16, 100, "f32"), (64, 32, 50, 9, 1000, "f32"), (128, 64, 50, 8, 130, "f32"),
                                                  (128, 32, 50, 24, 100, "bf16"), (128, 32, 64, 3, 2, "f32")])
def test_eval_launches_are_bit_identical_to_the_forward_and_rank_kernels(D, hid, T, B, NI, compute):
    n_items = 3000
    P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=3 + D + T)
    eng = make_engine(P, n_items, D, T, hid, compute)
    pl = eng.plan(B, T, NI, need_grad=False)
    assert eng.eval_fused_ok(pl)
    for seed in range(4):
        cu = {k: v.cuda() for k, v in eval_batch(B, T, n_items, NI, 40 + seed, dup_positive=NI > 4).items()}
        torch.cuda.synchronize()          # (the engine's stream does not wait for torch's)
        own, r, r0 = old_path(eng, pl, cu)
        eng.load_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"])
        eng.enqueue_eval(pl, FIX, with_loss=True, want_scores=True)
        eng.sync()
        eng.check_index_error(pl)
        assert torch.equal(pl.ev_p, own), float((pl.ev_p - own).abs().max())
        assert torch.equal(pl.ev_rank, r) and torch.equal(pl.ev_rank_raw, r0)
        if NI > 4:                      # the tie rule: a row whose positive repeats among the negatives loses (at least) one more rank with fix_value
            assert bool((pl.ev_rank[::2] >= pl.ev_rank_raw[::2] + 1).all()) and bool((pl.ev_rank >= pl.ev_rank_raw).all())
            ties = (own[:, 1:] == own[:, :1]).sum(1).int()           # (the draw can repeat the positive's id, or another row of equal score)
            assert bool((pl.ev_rank - pl.ev_rank_raw >= ties).all())
        y = cu["label"]
        want = torch.nn.functional.binary_cross_entropy(own.double(), y.double(), reduction="none").sum(1) / (B * NI)      # train_sr.py:63-64
        assert float((pl.ev_loss_part.double() - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-9


def test_eval_ranks_over_32_random_batches_and_against_the_oracle():
    """The headline evaluation shape (B 256, T 50, D 128, 999 negatives) over 32 batches through the captured graph (eval_epoch): every rank
    and every score equal to the launches it replaces; the first batch's scores within 2e-6 of the oracle's (CPU restatement of
    model_seq.py:416-443 in eval mode)."""
    n_items, D, T, hid, B, NI = 20000, 128, 50, 32, 256, 1000
    P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=77)
    eng = make_engine(P, n_items, D, T, hid)
    pl = eng.plan(B, T, NI, need_grad=False)
    batches = [eval_batch(B, T, n_items, NI, 900 + i) for i in range(32)]
    cus = [{k: v.cuda() for k, v in b.items()} for b in batches]
    torch.cuda.synchronize()
    packed = torch.stack([eng.pack_batch(pl, c["i_node"], c["neg_samples"], c["seq_d1"], c["seq_d2"], c["label"], c["domain_id"]) for c in cus])
    out = eng.eval_epoch(pl, packed, FIX, with_loss=True, use_graph=True)
    eng.sync()
    assert (FIX, True) in pl.eval_graphs
    for i, c in enumerate(cus):
        own, r, r0 = old_path(eng, pl, c)
        assert torch.equal(out[i, :B], r) and torch.equal(out[i, B:2 * B], r0), i
        want = torch.nn.functional.binary_cross_entropy(own.double(), c["label"].double(), reduction="none").sum(1) / (B * NI)
        got = out[i, 2 * B:].view(torch.float32).double()
        assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-9
    b = batches[0]
    with torch.no_grad():
        p1, p2 = orc.sasrec_forward(P, b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"])
    want = torch.where(b["domain_id"][:, None] != 0, p2.reshape(B, -1), p1.reshape(B, -1))
    eng.load_batch(pl, *(cus[0][k] for k in ("i_node", "neg_samples", "seq_d1", "seq_d2", "label", "domain_id")))
    eng.enqueue_eval(pl, FIX, want_scores=True)
    eng.sync()
    assert float((pl.ev_p.cpu() - want).abs().max()) < 2e-6


def _write_csv(path, n, rng, lo1, hi1, lo2, hi2):
    rows = ["user_id,seq_d1,seq_d2,domain_id"]
    for u in range(n):
        dom = int(rng.random() < 0.5)
        l1 = int(rng.integers(1 if dom == 0 else 0, 9))
        l2 = int(rng.integers(1 if dom == 1 else 0, 9))
        s1 = [int(x) for x in rng.integers(lo1, hi1, l1)]
        s2 = [int(x) for x in rng.integers(lo2, hi2, l2)]
        rows.append(f'{u},"{json.dumps(s1)}","{json.dumps(s2)}",{dom}')
    with open(path, "w") as f:
        f.write("\n".join(rows) + "\n")


@pytest.mark.parametrize("emb,overlap", [(128, False), (64, True)])
def test_train_sr_test_gives_the_same_metrics_either_way(tmp_path, emb, overlap):
    """train_sr.test() through SASRec.eval_ranks (four launches per batch, a graph) and through model.forward + the rank kernel on the same
    evaluation set and negatives: the same seven metrics per split, the same loss to rounding."""
    import argparse
    from amid_amd import model_seq
    from amid_amd.dataset_seq import DeviceBatches, DualDomainSeqDataset
    from amid_amd.train_sr import test
    rng = np.random.default_rng(5)
    _write_csv(tmp_path / "toy_test.csv", 200, rng, 1, 400, 400, 900)
    ds = DualDomainSeqDataset(seq_len=20, isTrain=False, neg_nums=99, long_length=7, pad_id=1001, seed=3, csv_path=str(tmp_path / "toy_test.csv"))
    model = model_seq.SASRec(10, emb, 1100, emb, 20, 32, 32, False, False, 0.5, 0.5, seed=2)
    args = argparse.Namespace(overlap=overlap)
    res = {}
    for fused in (True, False):
        model.engine.EVAL_FUSED = fused
        vb = DeviceBatches(ds, 32, shuffle=False, device="cuda:0", seed=9)
        res[fused] = test(model, args, vb)
    assert set(res[True]) == set(res[False])
    for k, v in res[False].items():
        if k == "loss":
            assert abs(res[True][k] - v) <= 1e-6 * abs(v)
        else:
            assert res[True][k] == v or all(np.isnan(a) and np.isnan(b) or a == b for a, b in zip(res[True][k], v)), k
