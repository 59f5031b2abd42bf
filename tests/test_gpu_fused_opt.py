"""SasrecEngine.FUSED_OPT outside the folded step (round 6): every single-GPU train step's gradient tail applies Adam itself
(amid_grad_tail_opt_f32 without its position-row role; the parameters whose gradients kernels write straight into dense.grad -- the comp
modules' -- as its `left` range).  On / off, model variant by model variant: the same parameters after K steps, bit for bit, and one
launch fewer.  Reference step: optimizer.zero_grad(); loss.backward(); optimizer.step(), /root/reference/train_sr.py:213-215."""
import pytest
import torch

from oracle import amid_oracle as orc

pytestmark = pytest.mark.gpu


def build(kind, P, n_items, D, T, hid, B):
    from amid_amd.engine import SasrecEngine
    from amid_amd.engine_bert import Bert4recEngine
    kw = dict(lr=1e-3, seed=41)
    if kind == "bert":
        eng = Bert4recEngine(n_items, 128, T, hid, **kw)
    elif kind == "bert_itc":
        eng = Bert4recEngine(n_items, 128, T, hid, comp="itc", comp_bs=B, comp_threshold=0.02, **kw)
    elif kind == "itc":
        eng = SasrecEngine(n_items, D, T, hid, itc_bs=B, itc_threshold=0.02, **kw)
    elif kind == "inc":
        eng = SasrecEngine(n_items, D, T, hid, inc_bs=B, inc_threshold=0.02, **kw)
    elif kind == "dr":
        eng = SasrecEngine(n_items, D, T, hid, dr=True, **kw)
    elif kind == "bf16":
        eng = SasrecEngine(n_items, D, T, hid, compute="bf16", **kw)
    else:
        eng = SasrecEngine(n_items, D, T, hid, **kw)
    eng.load_state_dict(P)
    return eng


def shapes(kind, n_items, D, T, hid, B):
    if kind == "bert":
        return orc.bert4rec_param_shapes(n_items, hid)
    if kind == "bert_itc":
        return orc.bert4rec_param_shapes(n_items, hid, itc_bs=B)
    return orc.sasrec_param_shapes(n_items, D, T, hid, itc_bs=B if kind == "itc" else 0, dr=kind == "dr", inc_bs=B if kind == "inc" else 0)


@pytest.mark.parametrize("kind,D,T,B,pool", [("plain", 64, 50, 64, True), ("plain", 128, 20, 96, True), ("plain", 128, 50, 40, False), ("bert", 128, 50, 96, True),
                                              ("itc", 128, 50, 48, True), ("inc", 64, 20, 32, False), ("dr", 128, 20, 64, False), ("bf16", 128, 50, 96, True),
                                              ("bert_itc", 128, 20, 32, False)])
def test_optimizer_in_the_tail_of_every_single_gpu_step_is_bit_identical(kind, D, T, B, pool):
    from amid_amd._lib import lib
    n_items, hid, K = 900, 32, 7
    P = orc.random_params(shapes(kind, n_items, D, T, hid, B), seed=5)
    batches = [orc.synthetic_batch(B, T, n_items - 1, pad_id=n_items - 1 if not kind.startswith("bert") else 0, neg=1, seed=60 + t) for t in range(3)]
    out = {}
    for on in (False, True):
        eng = build(kind, P, n_items, D, T, hid, B)
        eng.FUSED_OPT = on
        pl = eng.plan(B, T, 2, need_grad=True)
        cus = [{k: v.cuda() for k, v in b.items()} for b in batches]
        torch.cuda.synchronize()          # (the engine's stream does not wait for torch's: the batches must have landed before load_batch reads them)
        if pool:
            eng.set_input_pool(pl, torch.stack([eng.pack_batch(pl, c["i_node"], c["neg_samples"], c["seq_d1"], c["seq_d2"], c["label"], c["domain_id"]) for c in cus]))
        names = []
        L = lib()
        orig = L.call
        L.call = lambda name, *a: (names.append(name), orig(name, *a))[1]
        losses = []
        try:
            for t in range(K):
                if not pool:
                    c = cus[t % 3]
                    eng.load_batch(pl, c["i_node"], c["neg_samples"], c["seq_d1"], c["seq_d2"], c["label"], c["domain_id"])
                if t == 1:
                    names.clear()
                eng.enqueue_train_step(pl)
                eng.sync()
                losses.append(pl.dr_losses.clone() if kind == "dr" else float(pl.loss.item()))
        finally:
            del L.call
        eng.check_index_error(pl)
        eng.flush_table()
        eng.sync()
        out[on] = dict(n=len(names) // (K - 1), names=set(names), losses=losses, sd={k: v.cpu().clone() for k, v in eng.state_dict().items()},
                       m=eng.dense.m.cpu().clone(), v=eng.dense.v.cpu().clone())
    a, b = out[False], out[True]
    assert "amid_grad_tail_opt_f32" in b["names"] and "amid_grad_tail_opt_f32" not in a["names"], (kind, sorted(b["names"]))
    assert b["n"] == a["n"] - 1, (a["n"], b["n"])
    for x, y in zip(a["losses"], b["losses"]):
        assert torch.equal(x, y) if torch.is_tensor(x) else x == y, (a["losses"], b["losses"])
    assert torch.equal(a["m"], b["m"]) and torch.equal(a["v"], b["v"])
    for k, want in a["sd"].items():
        assert torch.equal(b["sd"][k], want), (kind, k, float((b["sd"][k] - want).abs().max()))
