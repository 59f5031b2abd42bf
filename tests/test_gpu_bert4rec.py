"""GPU parity of the BERT4Rec path (forward, loss, backward, optimizer, graph replay, nn.Module surface) against the CPU
oracle and the reference-generated golden vectors.  Everything runs through libamid_hip.so."""
import os

import numpy as np
import pytest
import torch

from oracle import amid_oracle as orc
from tests.test_gpu_sasrec import GOLDEN, dense_table_grad, load_golden, log, rel_l2, relmax

pytestmark = pytest.mark.gpu
D = orc.BERT_HIDDEN


def make_engine(P, T, lr=5e-4, seed=0):
    from amid_amd.engine_bert import Bert4recEngine
    n_rows = P["item_emb_layer.emb_item.weight"].shape[0]
    hid = P["predictModule.fc.0.weight"].shape[0]
    eng = Bert4recEngine(n_rows, D, T, hid, lr=lr, seed=seed)
    eng.load_state_dict(P)
    return eng


def batch_with_masked_keys(Bn, T, n_items, seed, neg=1):
    """Synthetic batch whose seq_d2 holds zeros (masked keys, model_seq.py:288), one row entirely zero (all keys masked:
    the -1e9 fill makes the softmax uniform) and one row without any."""
    b = orc.synthetic_batch(Bn, T, n_items - 1, pad_id=0, neg=neg, seed=seed)
    g = torch.Generator().manual_seed(seed)
    z = torch.rand(Bn, T, generator=g) < 0.3
    b["seq_d2"] = torch.where(z, torch.zeros_like(b["seq_d2"]), b["seq_d2"].clamp(min=1))
    b["seq_d2"][0] = 0
    b["seq_d2"][1] = b["seq_d2"][1].clamp(min=1)
    return b


def run_forward(eng, batch, train, with_loss, step=None, seed=None):
    Bn, T = batch["seq_d1"].shape
    NI = 1 + batch["neg_samples"].reshape(Bn, -1).shape[1]
    pl = eng.plan(Bn, T, NI, need_grad=True)
    if step is not None:
        eng.set_step(step, seed)
    cu = {k: v.cuda() for k, v in batch.items()}
    eng.load_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu.get("label"), cu.get("domain_id"))
    eng.enqueue_prepare(pl, sparse=True)
    eng.enqueue_forward(pl, train=train, with_loss=with_loss)
    eng.sync()
    eng.check_index_error(pl)
    return pl


def grads_check(tag, eng, pl, grads, tol, l2tol):
    worst, bad = 0.0, []
    for name in eng.dense.slots:
        got = eng.dense.view(name, eng.dense.grad)
        if name.endswith("linear_layers.1.bias"):
            # the key bias shifts every score of a query row by the same amount: its gradient is analytically zero and
            # what either side computes is rounding noise -- bound it against the query-bias gradient instead
            ref = grads[name.replace("linear_layers.1", "linear_layers.0")].abs().max()
            assert float(got.abs().max()) < 1e-4 * float(ref) + 1e-9, name
            continue
        e, e2 = relmax(got, grads[name]), rel_l2(got, grads[name])
        log(f"{tag} grad {name:55s} relmax {e:.3e} l2 {e2:.3e}")
        worst = max(worst, e)
        if not (e < tol and e2 < l2tol):
            bad.append((name, e, e2))
    assert not bad, bad
    tg = dense_table_grad(eng, pl)
    e, e2 = relmax(tg, grads["item_emb_layer.emb_item.weight"]), rel_l2(tg, grads["item_emb_layer.emb_item.weight"])
    log(f"{tag} grad table relmax {e:.3e} l2 {e2:.3e}; worst dense {worst:.3e}")
    assert e < tol and e2 < l2tol


@pytest.mark.parametrize("T", [50, 17, 64])
@pytest.mark.parametrize("train", [False, True])
def test_forward_logits_vs_oracle(T, train):
    Bn, hid, n_items = 9, 32, 500
    P = orc.random_params(orc.bert4rec_param_shapes(n_items, hid), seed=T)
    batch = batch_with_masked_keys(Bn, T, n_items, seed=3)
    eng = make_engine(P, T, seed=77)
    masks = orc.philox_masks_bert4rec(Bn, T, seed=77, step=5) if train else None
    pl = run_forward(eng, batch, train=train, with_loss=False, step=5, seed=77)
    p1, p2 = orc.bert4rec_forward(P, batch["i_node"], batch["neg_samples"], batch["seq_d1"], batch["seq_d2"], masks)
    e1, e2 = relmax(pl.p1, p1), relmax(pl.p2, p2)
    log(f"bert fwd T={T} train={train}: logits relmax {e1:.3e} {e2:.3e}")
    assert e1 < 1e-4 and e2 < 1e-4          # north-star tolerance on fp32 logits
    assert e1 < 3e-5 and e2 < 3e-5          # what the kernels actually deliver


def golden_params(z):
    """The BERT goldens carry the parameter seed + a checksum instead of 2.5 MB of weights (make_golden.py)."""
    P = orc.random_params(orc.bert4rec_param_shapes(int(z["n_items"]), int(z["hid"])), seed=int(z["param_seed"]))
    assert abs(sum(float(v.double().sum()) for v in P.values()) - float(z["param_sum"])) < 1e-6 * max(1.0, abs(float(z["param_sum"])))
    return P


# g5_bert4rec_train.npz (the reference's train-mode step under ITS torch-RNG dropout masks) pins the oracle's dropout placement in
# tests/test_oracle_golden.py; the HIP path draws its masks from the counter RNG instead, which the oracle restates bit for bit
# (philox_masks_bert4rec), so train-mode GPU parity is the oracle comparison above.
def test_forward_golden_bert4rec_eval():
    z, _, B, _ = load_golden("g3_bert4rec_eval.npz")
    P = golden_params(z)
    eng = make_engine(P, B["seq_d1"].shape[1])
    pl = run_forward(eng, B, train=False, with_loss=False)
    e1, e2 = relmax(pl.p1, z["p1"]), relmax(pl.p2, z["p2"])
    log(f"golden g3_bert4rec_eval: {e1:.3e} {e2:.3e}")
    assert e1 < 1e-4 and e2 < 1e-4


@pytest.mark.parametrize("train", [False, True])
def test_backward_grads_vs_oracle(train):
    T, Bn, hid, n_items = 50, 7, 32, 400
    P = orc.random_params(orc.bert4rec_param_shapes(n_items, hid), seed=21)
    batch = batch_with_masked_keys(Bn, T, n_items, seed=4)
    seed, step = 99, 2
    masks = orc.philox_masks_bert4rec(Bn, T, seed=seed, step=step) if train else None
    loss, (p1, p2), grads = orc.loss_and_grads("bert4rec", P, batch, masks)
    eng = make_engine(P, T, seed=seed)
    pl = run_forward(eng, batch, train=train, with_loss=True, step=step, seed=seed)
    eng.enqueue_backward(pl, train=train)
    eng.sync()
    assert abs(float(pl.loss.item()) - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    assert relmax(pl.p1, p1) < 3e-5
    grads_check(f"bert bwd train={train}", eng, pl, grads, 2e-4, 2e-4)          # GELU is smooth: no knife edge here


@pytest.mark.parametrize("Bn,T", [(64, 50), (256, 20), (9, 13)])
def test_strips_on_bf16_pieces_have_fp32_accuracy(Bn, T):
    """Bert4recEngine.STRIP_P3 (csrc/bert_strip.hip MODE 3: every strip product as six bf16 piece pairs over three-plane tile images,
    amid_bert_weight_images_f32) against the fp32 matrix instructions on the same step, dropout on: every saved activation and every
    gradient within 4e-6 of its tensor's largest entry (the bar of SASRec's strips on pieces, tests/test_gpu_seqn.py); operands ROUNDED to
    bf16 would sit at 3e-3.  The three-plane images sum to the weight tiles exactly."""
    hid, n_items = 32, 900
    P = orc.random_params(orc.bert4rec_param_shapes(n_items, hid), seed=70 + T)
    batch = batch_with_masked_keys(Bn, T, n_items, seed=12)
    seed, step = 17, 6
    out = {}
    for p3 in (False, True):
        eng = make_engine(P, T, seed=seed)
        eng.STRIP_P3 = p3
        pl = run_forward(eng, batch, train=True, with_loss=True, step=step, seed=seed)
        assert pl.strip and bool(getattr(pl, "p3_fwd", False)) == p3
        eng.enqueue_backward(pl, train=True)
        eng.sync()
        rec = {f"{k}{l}": getattr(pl, k)[l].clone() for k in ("q", "k", "v", "x1", "y2", "pre", "h") for l in (0, 1)}
        rec.update(x1_=pl.x[1].clone(), x2_=pl.x[2].clone(), loss=pl.loss.clone(), rows=dense_table_grad(eng, pl))
        rec.update({"g:" + n: eng.dense.view(n, eng.dense.grad).clone() for n in eng.dense.slots})
        out[p3] = rec
        if p3:      # hi + mid + lo of every tile image = the fp32 tile, bit for bit
            src, ld, trn, n, buf = eng._tile_images()
            img = buf.float().sum(3).cpu()                  # [2][2][24][D D] in fragment order: compare as multisets per tile
            w = eng.dense.view("transform1.0.attention.linear_layers.0.weight", eng.dense.data).float().cpu().reshape(-1)
            assert torch.equal(torch.sort(img[0, 0, 0]).values, torch.sort(w).values)
            assert torch.equal(torch.sort(img[0, 0, 12]).values, torch.sort(w).values)
    worst = 0.0
    for name, want in out[False].items():
        got = out[True][name]
        if name.endswith("linear_layers.1.bias"):
            continue                                         # (analytically zero: rounding noise on both sides)
        live = want.abs().max()
        e = float((got - want).abs().max() / live.clamp(min=1e-30))
        worst = max(worst, e)
        assert e < 4e-6, (name, e)
    log(f"bert strips on pieces B={Bn} T={T}: worst deviation from the fp32 instructions {worst:.2e} of a tensor's largest entry")


def test_tile_images_by_the_gather_riders_equal_the_standalone_launch():
    """The step's 96 weight-tile images written by extra workgroups of the gather K1 (amid_embed_fwd_tiles_f32) are, bit for bit, what
    amid_bert_weight_images_f32 writes in a launch of its own; the gather's rows are unchanged by the riders; and every image's three planes
    sum to the fp32 tile (or its transpose) exactly -- checked on w_2's column blocks, the tiles with a source stride of 512."""
    from amid_amd._lib import lib
    hid, n_items, T, Bn = 32, 700, 20, 16
    P = orc.random_params(orc.bert4rec_param_shapes(n_items, hid), seed=5)
    eng = make_engine(P, T, seed=1)
    batch = batch_with_masked_keys(Bn, T, n_items, seed=3)
    pl = run_forward(eng, batch, train=False, with_loss=False)            # (K1 of this forward wrote the images: strips on pieces by default)
    assert pl.strip and eng.STRIP_P3
    src, ld, trn, n, buf = eng._tile_images()
    by_riders, rows = buf.clone(), pl.xg.clone()
    buf.zero_()
    lib().call("amid_bert_weight_images_f32", src, ld, trn, n, buf.data_ptr(), eng.s)
    eng.sync()
    assert torch.equal(by_riders.view(torch.int16), buf.view(torch.int16))
    want_rows = eng.table[pl.idx_all.long()]
    assert torch.equal(rows[: want_rows.shape[0]], want_rows)
    w2 = eng.dense.view("transform1.0.feed_forward.w_2.weight", eng.dense.data).float().cpu()       # [128][512]
    for c in range(4):
        tile = w2[:, 128 * c: 128 * (c + 1)]
        for tr, i in ((0, 8 + c), (1, 20 + c)):
            img = buf[0, 0, i].float().sum(0).cpu()          # fragment order: compare as multisets
            assert torch.equal(torch.sort(img).values, torch.sort((tile.t() if tr else tile).reshape(-1)).values), (c, tr)


@pytest.mark.parametrize("Bn,T,build", [(256, 50, ""), (128, 50, "_rt5"), (256, 20, "_rt3")])
def test_tile_builds_forward_backward_vs_oracle(Bn, T, build):
    """The three builds of the row-tile kernels (112 / 80 / 48 rows per workgroup: the headline batch, half of it, the mybank
    sequence length) forward + backward against the oracle, dropout on."""
    hid, n_items = 32, 2000
    P = orc.random_params(orc.bert4rec_param_shapes(n_items, hid), seed=60 + T)
    batch = batch_with_masked_keys(Bn, T, n_items, seed=8)
    seed, step = 5, 3
    masks = orc.philox_masks_bert4rec(Bn, T, seed=seed, step=step)
    loss, (p1, p2), grads = orc.loss_and_grads("bert4rec", P, batch, masks)
    eng = make_engine(P, T, seed=seed)
    pl = run_forward(eng, batch, train=True, with_loss=True, step=step, seed=seed)
    assert pl.rt_suffix == build
    eng.enqueue_backward(pl, train=True)
    eng.sync()
    assert abs(float(pl.loss.item()) - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    assert relmax(pl.p1, p1) < 1e-4
    for name in eng.dense.slots:
        if name.endswith("linear_layers.1.bias"):
            continue
        assert rel_l2(eng.dense.view(name, eng.dense.grad), grads[name]) < 1e-3, name
    assert rel_l2(dense_table_grad(eng, pl), grads["item_emb_layer.emb_item.weight"]) < 1e-3


def test_backward_golden_bert4rec_grads():
    z, _, B, G = load_golden("g4_bert4rec_grads.npz")
    P = golden_params(z)
    B = dict(B)
    B["label"] = torch.from_numpy(z["labels"])
    eng = make_engine(P, B["seq_d1"].shape[1])
    pl = run_forward(eng, B, train=False, with_loss=True)
    eng.enqueue_backward(pl, train=False)
    eng.sync()
    assert abs(float(pl.loss.item()) - float(z["loss"])) < 1e-5
    grads_check("golden g4 bert", eng, pl, G, 5e-4, 5e-4)


@pytest.mark.parametrize("T,Bn", [(20, 16), (50, 12), (13, 9)])
def test_train_steps_track_dense_adam_reference(T, Bn):
    """K full steps (dropout on, lazy table Adam) against the oracle's dense-Adam trajectory."""
    hid, n_items, K = 32, 300, 5
    P = orc.random_params(orc.bert4rec_param_shapes(n_items, hid), seed=31)
    _bert_traj(P, T, Bn, hid, n_items, K, 4242)


def _bert_traj(P, T, Bn, hid, n_items, K, seed):
    eng = make_engine(P, T, lr=1e-3, seed=seed)
    Po = {k: v.clone() for k, v in P.items()}
    opt = orc.DenseAdam(Po, lr=1e-3)
    pl = eng.plan(Bn, T, 2, need_grad=True)
    for t in range(1, K + 1):
        batch = batch_with_masked_keys(Bn, T, 60 if t in (1, 2, 5) else n_items, seed=100 + t)
        masks = orc.philox_masks_bert4rec(Bn, T, seed=seed, step=t)
        loss_o = orc.train_step("bert4rec", Po, opt, batch, masks)
        cu = {k: v.cuda() for k, v in batch.items()}
        eng.load_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"])
        eng.enqueue_train_step(pl)
        eng.sync()
        assert eng.step == t
        log(f"bert traj step {t}: loss gpu {float(pl.loss.item()):.7f} oracle {loss_o:.7f}")
        assert abs(float(pl.loss.item()) - loss_o) < 5e-5
    eng.flush_table()
    eng.sync()
    sd = eng.state_dict()
    for k, v in Po.items():
        if k.endswith("linear_layers.1.bias"):
            continue                 # zero-gradient key bias: Adam normalises pure rounding noise to +-lr per step on either side
        d = float((sd[k].cpu() - v).abs().max())
        log(f"bert traj param {k:55s} maxabs {d:.3e}")
        assert d < 2e-4, k


def test_graph_replay_equals_eager():
    T, Bn, hid, n_items = 20, 8, 16, 200
    P = orc.random_params(orc.bert4rec_param_shapes(n_items, hid), seed=5)
    batches = [batch_with_masked_keys(Bn, T, n_items, seed=200 + t) for t in range(4)]

    def run(use_graph):
        eng = make_engine(P, T, lr=1e-3, seed=9)
        pl = eng.plan(Bn, T, 2, need_grad=True)
        losses = []
        if use_graph:
            cu = {k: v.cuda() for k, v in batches[0].items()}
            eng.load_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"])
            eng.capture_train_step(pl)
        for b in batches:
            cu = {k: v.cuda() for k, v in b.items()}
            eng.load_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"])
            if use_graph:
                eng.replay_train_step(pl)
            else:
                eng.enqueue_train_step(pl)
            eng.sync()
            losses.append(float(pl.loss.item()))
        eng.flush_table()
        eng.sync()
        return losses, {k: v.cpu().clone() for k, v in eng.state_dict().items()}

    l0, s0 = run(False)
    l1, s1 = run(True)
    assert l0 == l1
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def test_module_surface_matches_reference():
    """BERT4Rec nn.Module: reference constructor / forward signatures, state_dict keys, autograd + torch.optim.Adam loop."""
    from amid_amd.model_seq import BERT4Rec
    T, Bn, hid, n_items = 20, 8, 16, 120
    model = BERT4Rec(10, D, n_items, D, T, hid, Bn, False, False, 0.5, 0.5).cuda()
    want = set(orc.bert4rec_param_shapes(n_items, hid))
    assert set(model.state_dict().keys()) == want
    P = orc.random_params(orc.bert4rec_param_shapes(n_items, hid), seed=8)
    model.load_state_dict({k: v.cuda() for k, v in P.items()})
    batch = batch_with_masked_keys(Bn, T, n_items, seed=12)
    cu = {k: v.cuda() for k, v in batch.items()}
    model.eval()
    with torch.no_grad():
        p1, p2 = model(None, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], None, None, isTrain=False)
    o1, o2 = orc.bert4rec_forward(P, batch["i_node"], batch["neg_samples"], batch["seq_d1"], batch["seq_d2"])
    assert relmax(p1, o1.squeeze()) < 3e-5 and relmax(p2, o2.squeeze()) < 3e-5
    # reference loop in eval mode (no dropout) with torch's own Adam: one step must match the oracle's dense Adam
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    p1, p2 = model(None, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], None, None)
    lab, dom = cu["label"].float(), cu["domain_id"]
    m1 = (dom == 0).float().unsqueeze(1)
    bce = torch.nn.BCELoss(reduction="none")
    loss = ((bce(p1.reshape(Bn, -1), lab) * m1).sum() + (bce(p2.reshape(Bn, -1), lab) * (1 - m1)).sum()) / lab.numel()
    loss.backward()
    opt.step()
    Po = {k: v.clone() for k, v in P.items()}
    loss_o = orc.train_step("bert4rec", Po, orc.DenseAdam(Po, lr=1e-3), batch, None)
    assert abs(float(loss.detach()) - loss_o) < 5e-5
    sd = model.state_dict()
    for k, v in Po.items():
        if not k.endswith("linear_layers.1.bias"):          # zero-gradient key bias, see above
            assert float((sd[k].cpu() - v).abs().max()) < 2e-4, k


# ---------------------------------------------------------------------------- isDR (model_seq.py:268-271, :301-305)
@pytest.mark.parametrize("mode", [0, 1])
def test_dr_golden_outputs_losses_grads(mode):
    """BERT4Rec(isDR=True) against the reference's own six outputs, losses and gradients of both objectives (g13)."""
    from amid_amd.engine_bert import Bert4recEngine
    z = np.load(os.path.join(GOLDEN, "g13_bert4rec_dr.npz"))
    P = orc.random_params(orc.bert4rec_param_shapes(int(z["n_items"]), int(z["hid"]), dr=True), seed=int(z["param_seed"]))
    B = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("B/")}
    Bn, T = B["seq_d1"].shape
    eng = Bert4recEngine(int(z["n_items"]), 128, T, int(z["hid"]), lr=1e-3, seed=0, dr=True, dr_e_w=float(z["dr_e_w"]))
    eng.load_state_dict(P)
    pl = eng.plan(Bn, T, 2, need_grad=True)
    cu = {k: v.cuda() for k, v in B.items()}
    eng.dr_mode = mode
    eng.load_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], torch.from_numpy(z["labels"]).cuda(), cu["domain_id"],
                   torch.from_numpy(z["ob_label"]).cuda())
    eng.enqueue_prepare(pl, sparse=True)
    eng.enqueue_forward(pl, train=False, with_loss=True)
    eng.enqueue_backward(pl, train=False)
    eng.sync()
    for got, name in zip((pl.p1, pl.p2, pl.ips1, pl.ips2, pl.g1, pl.g2), ("p1", "p2", "ips1", "ips2", "g1", "g2")):
        assert relmax(got, z[name]) < 1e-4, name
    losses = pl.dr_losses.cpu()
    for c, name in enumerate(("loss_cls", "loss_dr_e", "loss_dr_r")):
        assert abs(float(losses[c]) - float(z[name])) < 2e-5 * max(1.0, abs(float(z[name]))), name
    pre = "GE/" if mode == 0 else "GR/"
    G = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
    bad = []
    for name, want in G.items():
        if name == "item_emb_layer.emb_item.weight" or name.endswith("linear_layers.1.bias"):
            continue
        got = eng.dense.view(name, eng.dense.grad)
        e = relmax(got, want) if float(want.abs().max()) > 1e-12 else float(got.abs().max().cpu())
        if not e < 1e-3:
            bad.append((name, e))
    assert not bad, bad
    tg = dense_table_grad(eng, pl)
    assert relmax(tg, G["item_emb_layer.emb_item.weight"]) < 1e-3


def test_dr_train_steps_fused_vs_oracle_two_optimizers():
    """BERT4Rec(isDR=True) train steps (fused three-head scorer launch, dropout on) alternating the two objectives / Adam states
    against the oracle's two dense Adams; graph replay bit-identical to eager."""
    from amid_amd.engine_bert import Bert4recEngine
    T, Bn, hid, n_items, w, lr = 12, 6, 16, 200, 0.1, 1e-3
    P = orc.random_params(orc.bert4rec_param_shapes(n_items, hid, dr=True), seed=21)
    batches = []
    for t in range(4):
        b = orc.synthetic_batch(Bn, T, n_items - 1, pad_id=0, neg=1, seed=700 + t)
        b["ob_label"] = (torch.rand(Bn, generator=torch.Generator().manual_seed(t)) < 0.6).long()
        batches.append(b)
    seed = 17
    Po = {k: v.clone() for k, v in P.items()}
    opts = [orc.DenseAdam(Po, lr=lr), orc.DenseAdam(Po, lr=lr * 0.5)]
    want = []
    for t, b in enumerate(batches):
        k = t % 2
        masks = orc.philox_masks_bert4rec(Bn, T, seed=seed + (0 if k == 0 else 0x9E3779B9), step=opts[k].t + 1)
        info, _, grads = orc.dr_loss_and_grads(Po, b, "e" if k == 0 else "r", masks, dr_e_w=w, model="bert4rec")
        opts[k].step(Po, grads)
        want.append(float(info["loss"]))

    def run(use_graph):
        eng = Bert4recEngine(n_items, 128, T, hid, lr=lr, seed=seed, dr=True, dr_e_w=w)
        eng.load_state_dict(P)
        pl = eng.plan(Bn, T, 2, need_grad=True)
        got = []
        for t, b in enumerate(batches):
            k = t % 2
            eng.select_optimizer(k, lr=lr * (1.0 if k == 0 else 0.5))
            eng.dr_mode = k
            cu = {kk: v.cuda() for kk, v in b.items()}
            eng.load_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"], cu["ob_label"])
            if use_graph:
                if not eng.has_graph(pl):
                    eng.capture_train_step(pl)
                eng.replay_train_step(pl)
            else:
                eng.enqueue_train_step(pl)
            eng.sync()
            ls = pl.dr_losses.cpu()
            got.append(float(ls[0] + w * ls[1]) if k == 0 else float(ls[2]))
        eng.flush_table(); eng.sync()
        return got, {kk: v.cpu().clone() for kk, v in eng.state_dict().items()}

    g0, s0 = run(False)
    g1, s1 = run(True)
    assert g0 == g1 and all(torch.equal(s0[k], s1[k]) for k in s0)
    for a, b_ in zip(g0, want):
        assert abs(a - b_) < 5e-5 * max(1.0, abs(b_)), (g0, want)
    for k, v in Po.items():
        d = (s0[k] - v).abs()
        if k.endswith("linear_layers.1.bias"):
            continue
        # Adam's first steps move every weight by ~lr * sign(g): where the true gradient is ~0 (rows with ob_label 0 give the
        # encoders no gradient under loss_dr_r) rounding noise picks the sign, so a few elements may sit one step apart
        assert float((d > 3e-4).float().mean()) < 2e-3 and float(d.max()) < 2.5e-3, (k, float(d.max()), float((d > 3e-4).float().mean()))


# ---------------------------------------------------------------------------- isInC / isItC (model_seq.py:283-294)
def comp_kw(kind, bs, thr):
    return dict(comp=kind, comp_bs=bs, comp_threshold=thr)


def comp_fwd_kw(kind, thr):
    return dict(isInC=True, threshold1=thr) if kind == "inc" else dict(isItC=True, threshold2=thr)


@pytest.mark.parametrize("kind", ["inc", "itc"])
def test_comp_golden_outputs_loss_grads(kind):
    """BERT4Rec(isInC=True) / (isItC=True) against the reference's own logits, loss and gradients (g14 / g15): the comp module
    in front of the encoders, 2T tokens under the T-token key mask tiled twice."""
    from amid_amd.engine_bert import Bert4recEngine
    z = np.load(os.path.join(GOLDEN, "g14_bert4rec_inc.npz" if kind == "inc" else "g15_bert4rec_itc.npz"))
    B = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("B/")}
    Bn, T = B["seq_d1"].shape
    P = orc.random_params(orc.bert4rec_param_shapes(int(z["n_items"]), int(z["hid"]), inc_bs=Bn if kind == "inc" else 0,
                                                    itc_bs=Bn if kind == "itc" else 0), seed=int(z["param_seed"]))
    P["item_emb_layer.emb_item.weight"] = P["item_emb_layer.emb_item.weight"] * float(z["table_scale"])
    thr = float(z["threshold"])
    eng = Bert4recEngine(int(z["n_items"]), 128, T, int(z["hid"]), lr=1e-3, seed=0, **comp_kw(kind, Bn, thr))
    eng.load_state_dict(P)
    pl = eng.plan(Bn, T, 2, need_grad=True)
    assert pl.shape.Tenc == 2 * T
    cu = {k: v.cuda() for k, v in B.items()}
    eng.load_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], torch.from_numpy(z["labels"]).cuda(), cu["domain_id"])
    eng.enqueue_prepare(pl, sparse=True)
    eng.enqueue_forward(pl, train=False, with_loss=True)
    eng.enqueue_backward(pl, train=False)
    eng.sync()
    assert np.array_equal(pl.inc_gate[0].cpu().numpy().astype(bool), z["gate_d1"])
    assert np.array_equal(pl.inc_gate[1].cpu().numpy().astype(bool), z["gate_d2"])
    assert relmax(pl.p1, z["p1"]) < 1e-4 and relmax(pl.p2, z["p2"]) < 1e-4
    assert abs(float(pl.loss.cpu()) - float(z["loss"])) < 2e-5
    G = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("G/")}
    bad = []
    for name, want in G.items():
        if name == "item_emb_layer.emb_item.weight" or name.endswith("linear_layers.1.bias"):
            continue
        got = eng.dense.view(name, eng.dense.grad)
        e = relmax(got, want) if float(want.abs().max()) > 1e-12 else float(got.abs().max().cpu())
        if not e < 1e-3:
            bad.append((name, e))
    assert not bad, bad
    assert any(n.startswith(kind + "_d") for n in G)
    tg = dense_table_grad(eng, pl)
    assert relmax(tg, G["item_emb_layer.emb_item.weight"]) < 1e-3


@pytest.mark.parametrize("kind,Bn,T", [("inc", 6, 12), ("itc", 6, 12), ("itc", 32, 50), ("inc", 24, 20)])
def test_comp_train_steps_vs_oracle_and_graph(kind, Bn, T):
    """Train steps with dropout on (the encoders' Philox sites indexed over 2T tokens) against the oracle's dense Adam; graph
    replay bit-identical to eager.  The table is scaled so that the batch softmax gates are mixed, and the threshold sits in
    the widest gap of the first batch's softmax so that rounding cannot flip a gate."""
    from amid_amd.engine_bert import Bert4recEngine
    hid, n_items, lr, seed = 16, 400, 1e-3, 23
    P = orc.random_params(orc.bert4rec_param_shapes(n_items, hid, inc_bs=Bn if kind == "inc" else 0, itc_bs=Bn if kind == "itc" else 0),
                          seed=31)
    P["item_emb_layer.emb_item.weight"] = P["item_emb_layer.emb_item.weight"] * 0.1
    batches = [batch_with_masked_keys(Bn, T, n_items, 900 + t) for t in range(3)]
    taps = {}
    orc.bert4rec_forward(P, batches[0]["i_node"], batches[0]["neg_samples"], batches[0]["seq_d1"], batches[0]["seq_d2"], None, taps=taps,
                         **comp_fwd_kw(kind, 0.5))
    sm = torch.sort(torch.cat([taps[f"{kind}_d{d}"]["softmax"] for d in (1, 2)])).values
    gaps = sm[1:] - sm[:-1]
    i = int(torch.argmax(gaps))
    thr = float((sm[i] + sm[i + 1]) / 2)
    Po = {k: v.clone() for k, v in P.items()}
    opt = orc.DenseAdam(Po, lr=lr)
    want, margins = [], []
    for b in batches:
        masks = orc.philox_masks_bert4rec(Bn, 2 * T, seed=seed, step=opt.t + 1)
        tp = {}
        loss, _, grads = orc.loss_and_grads("bert4rec", Po, b, masks, taps=tp, **comp_fwd_kw(kind, thr))
        margins.append(min(tp[f"{kind}_d{d}"]["margin"] for d in (1, 2)))
        opt.step(Po, grads)
        want.append(float(loss))
    log(f"bert comp {kind} B {Bn} T {T}: threshold {thr:.5f} gate margins {margins}")
    if min(margins) < 1e-5:
        pytest.skip(f"a batch-softmax value sits within {min(margins):.2e} of the threshold: the gate is rounding-dependent")

    def run(use_graph):
        eng = Bert4recEngine(n_items, 128, T, hid, lr=lr, seed=seed, **comp_kw(kind, Bn, thr))
        eng.load_state_dict(P)
        pl = eng.plan(Bn, T, 2, need_grad=True)
        got = []
        for b in batches:
            cu = {kk: v.cuda() for kk, v in b.items()}
            eng.load_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"])
            if use_graph:
                if not eng.has_graph(pl):
                    eng.capture_train_step(pl)
                eng.replay_train_step(pl)
            else:
                eng.enqueue_train_step(pl)
            eng.sync()
            got.append(float(pl.loss.cpu()))
        eng.flush_table(); eng.sync()
        return got, {kk: v.cpu().clone() for kk, v in eng.state_dict().items()}

    g0, s0 = run(False)
    g1, s1 = run(True)
    assert g0 == g1 and all(torch.equal(s0[k], s1[k]) for k in s0)
    for a, b_ in zip(g0, want):
        assert abs(a - b_) < 5e-5 * max(1.0, abs(b_)), (g0, want)
    for k, v in Po.items():
        if k.endswith("linear_layers.1.bias"):
            continue
        d = (s0[k] - v).abs()
        assert float((d > 3e-4).float().mean()) < 2e-3 and float(d.max()) < 3.5e-3, (k, float(d.max()), float((d > 3e-4).float().mean()))


def test_comp_module_surface():
    """model_seq.BERT4Rec(isItC=True): state_dict keys, forward against the oracle, the batch-size contract, and the refusal of the
    combination the reference itself cannot run."""
    from amid_amd import model_seq as ms
    Bn, T, hid, n_items = 8, 10, 16, 300
    with pytest.raises(ValueError):
        ms.BERT4Rec(10, 128, n_items, 128, T, hid, Bn, True, True, 0.5, 0.5)
    m = ms.BERT4Rec(10, 128, n_items, 128, T, hid, Bn, False, True, 0.5, 0.13)
    sd = m.state_dict()
    assert sd["itc_d2.trans_bs.weight"].shape == (1, Bn) and sd["itc_d1.trans_nn.weight"].shape == (128, 128)
    P = {k: v.detach().cpu().clone() for k, v in sd.items()}
    P["item_emb_layer.emb_item.weight"] *= 0.1
    m.load_state_dict(P)
    b = batch_with_masked_keys(Bn, T, n_items, 77)
    m.eval()
    with torch.no_grad():
        p1, p2 = m(None, b["i_node"].cuda(), b["neg_samples"].cuda(), b["seq_d1"].cuda(), b["seq_d2"].cuda(), None, None)
    taps = {}
    w1, w2 = orc.bert4rec_forward(P, b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"], None, isItC=True, threshold2=0.13, taps=taps)
    if min(taps[f"itc_d{d}"]["margin"] for d in (1, 2)) > 1e-5:
        assert relmax(p1, w1) < 1e-4 and relmax(p2, w2) < 1e-4
    with pytest.raises(ValueError):      # trans_bs is Linear(bs, 1) over the batch: other batch sizes cannot run (as in the reference)
        m(None, b["i_node"][:4].cuda(), b["neg_samples"][:4].cuda(), b["seq_d1"][:4].cuda(), b["seq_d2"][:4].cuda(), None, None)
