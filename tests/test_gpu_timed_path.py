"""Parity of the path that train_step / graph replay / bench.py actually TIME: the fused step whose encoder work covers the live
sequences only (the step's own loss multiplies the other domain's terms of every sample by zero, train_sr.py:205-211), entered the
way the product enters it -- enqueue_local_grads / enqueue_train_step / capture + replay -- and compared with the CPU oracle.

The gradient comparisons are made kink-free instead of loose: the oracle is told the implementation's own relu decisions
(relu_keep), the test asserts that every decision that differs from the oracle's own sits within rounding of the kink, and the
gradients are then held to max-abs tolerances."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import amid_oracle as orc

pytestmark = pytest.mark.gpu
LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def log(msg):
    os.makedirs(LOG, exist_ok=True)
    with open(os.path.join(LOG, "parity.log"), "a") as f:
        f.write(msg + "\n")


def relmax(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def make_engine(P, T, lr=5e-4, seed=0, compute="f32"):
    from amid_amd.engine import SasrecEngine
    n_rows, D = P["item_emb_layer.emb_item.weight"].shape
    hid = P["predictModule.fc.0.weight"].shape[0]
    eng = SasrecEngine(n_rows, D, T, hid, lr=lr, seed=seed, compute=compute)
    eng.load_state_dict(P)
    return eng


def load(eng, pl, batch):
    cu = {k: v.cuda() for k, v in batch.items()}
    eng.load_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"])


def timed_local_grads(eng, pl, batch, step, seed):
    """Step `step` up to the local gradients exactly as train_step / capture_train_step enqueue it (t += 1, lazy-Adam catch-up,
    side-stream sort, forward + loss + backward over the live sequences, gradient tail)."""
    eng.set_step(step - 1, seed)
    load(eng, pl, batch)
    eng.enqueue_local_grads(pl)
    eng.sync()
    eng.check_index_error(pl)
    assert eng.step == step


def gpu_relu_keep(eng, pl, batch):
    """The implementation's relu decisions, from the saved relu outputs, as the oracle's relu_keep; -1 (the oracle's own
    decision) for the sequences the timed path does not encode or differentiate: domain g of the samples with domain_id != g."""
    Bn, T = batch["seq_d1"].shape
    M, D = Bn * T, eng.D
    dom = batch["domain_id"]
    out = {}
    for g in (0, 1):
        live = (dom == g)
        for l in (0, 1):
            h = pl.h[l][g * M:(g + 1) * M].reshape(Bn, T, D).cpu()
            keep = (h > 0).float()
            keep[~live] = -1.0
            out[f"sac{g + 1}.relu{l}"] = keep
    return out


def dense_table_grad(eng, pl):
    U = int(pl.n_uniq.item())
    g = torch.zeros(eng.n_rows, eng.D)
    g[pl.uniq_ids[:U].cpu().long()] = pl.uniq_grad[:U].cpu()
    return g


def check_grads(tag, eng, pl, grads, tol, l2tol):
    worst = worst2 = 0.0
    for name in eng.dense.slots:
        got, want = eng.dense.view(name, eng.dense.grad).cpu().clone(), grads[name].clone()
        if name.endswith("in_proj_bias"):            # the key-bias third is analytically zero (softmax shift invariance)
            n3 = got.numel() // 3
            got[n3:2 * n3] = 0; want[n3:2 * n3] = 0
        e, e2 = relmax(got, want), rel_l2(got, want)
        worst, worst2 = max(worst, e), max(worst2, e2)
        # (a scalar -- predictModule.fc.2.bias -- is one cancelling sum over the batch: its L2 error IS its max error, held to `tol`;
        # profiles/tools/probe/fuzz_timed.py, seeds 12 / 13: 9.2e-5 at B 300, T 32, D 64)
        assert e < tol and e2 < (l2tol if got.numel() > 8 else tol), (tag, name, e, e2)
    tg = dense_table_grad(eng, pl)
    e, e2 = relmax(tg, grads["item_emb_layer.emb_item.weight"]), rel_l2(tg, grads["item_emb_layer.emb_item.weight"])
    log(f"{tag}: worst dense grad relmax {worst:.3e} l2 {worst2:.3e}; table relmax {e:.3e} l2 {e2:.3e}")
    assert e < tol and e2 < l2tol, (tag, "table", e, e2)


def split_batch(Bn, T, n_items, seed, split):
    batch = orc.synthetic_batch(Bn, T, n_items - 1, pad_id=n_items - 1, neg=1, seed=seed)
    if split == "all0":
        batch["domain_id"] = torch.zeros(Bn, dtype=torch.long)
    elif split == "all1":
        batch["domain_id"] = torch.ones(Bn, dtype=torch.long)
    elif split == "one0":                 # a single live sequence in domain 0, the rest in domain 1
        batch["domain_id"] = torch.ones(Bn, dtype=torch.long)
        batch["domain_id"][Bn // 3] = 0
    return batch


# B, T, D, expected live-row build of the row-tile kernels, domain split
TIMED_CASES = [
    (256, 50, 128, "_rt4", "mixed"),      # BASELINE.json configs[1]: the headline shape, 50 live rows per workgroup
    (256, 20, 128, "_rt3", "mixed"),      # configs[3] (mybank, seq_len 20)
    (384, 50, 128, "_rt5", "mixed"),      # 75 live rows per workgroup: the 80-row build
    (512, 50, 128, "", "mixed"),          # configs[2]'s batch: 100 live rows, the 112-row build
    (256, 50, 64, "_rt4", "mixed"),
    (256, 50, 128, "_rt4", "all0"),       # every sample in domain 0: domain 1's encoder has no live sequence at all
    (256, 50, 128, "_rt4", "all1"),
    (200, 50, 128, "_rt3", "one0"),       # one live sequence in domain 0 (a one-row-tile domain), ragged last tiles
    (37, 13, 64, "_rt3", "mixed"),        # odd everything
    (1100, 50, 128, "", "mixed"),         # B > 1024: the live-sequence windows of the tiles / splits / attention slots
    (64, 16, 128, "_rt3", "mixed"),       # T <= 16 at D 128: the one-strip builds of the fused forward (seq_fwd_kernel<128, 1>, seqn <1, 4>)
    (300, 9, 128, "_rt3", "mixed"),       # ... with a ragged strip
    (256, 32, 128, "_rt3", "mixed"),      # T = 32: two full strips
    # D = 64 (the reference's default --emb_dim: 8 heads of 8 dims, two per column tile in the matrix-core attention and in the fused
    # forward, csrc/sasrec_seqn.hip <64, ...>): the two-strip x four-part build and a full 64-token sequence (cases 4 and 8 above: <64, 4, 2> / <64, 1, 4>)
    (300, 20, 64, "_rt3", "mixed"),
    (64, 64, 64, "_rt4", "all1"),
]


@pytest.mark.parametrize("Bn,T,D,build,split", TIMED_CASES)
def test_timed_path_loss_and_grads_vs_oracle(Bn, T, D, build, split):
    _timed_vs_oracle(Bn, T, D, build, split, compact_min=None)


@pytest.mark.parametrize("Bn,T,D,build,split", [TIMED_CASES[0], TIMED_CASES[1], TIMED_CASES[5], TIMED_CASES[7], TIMED_CASES[8]])
def test_timed_path_with_compact_index_list_vs_oracle(Bn, T, D, build, split):
    """The same with the step's sparse side (sort, segment reduce, row Adam inputs) on the compact index list of the live sequences,
    which the engine only takes for long lists (SasrecEngine.COMPACT_MIN_IDX): forced here for the small shapes."""
    _timed_vs_oracle(Bn, T, D, build, split, compact_min=0)


# The encoder's backward as ONE launch per step (amid_sas_seq_bwd_f32; SasrecEngine.SEQ_BACKWARD "auto" takes it for batches of more
# live sequences than CUs and at most two rounds, i.e. cases 2 and 3 above): forced on for the headline shape, degenerate domain
# splits, three key tiles (T 40) and many rounds, and forced off for a shape "auto" would take.
@pytest.mark.parametrize("Bn,T,D,split,force", [(256, 50, 128, "mixed", "1"), (256, 50, 128, "all0", "1"), (256, 50, 128, "all1", "1"),
                                                (200, 50, 128, "one0", "1"), (300, 40, 128, "mixed", "1"), (64, 33, 128, "mixed", "1"),
                                                (1100, 50, 128, "mixed", "1"), (512, 50, 128, "mixed", "0"),
                                                # T <= 32: the N-split build's two-strip / one-strip shapes (csrc/sasrec_seqn_bwd.hip)
                                                (256, 20, 128, "mixed", "1"), (300, 32, 128, "mixed", "1"), (64, 17, 128, "one0", "1"),
                                                (64, 16, 128, "mixed", "1"), (300, 9, 128, "mixed", "1"), (5, 1, 128, "all1", "1"),
                                                # D = 64 (8 heads of 8 dims: a wave per head, two per column tile, in the attention core)
                                                (256, 50, 64, "mixed", "1"), (300, 20, 64, "mixed", "1"), (64, 64, 64, "all1", "1"),
                                                (37, 13, 64, "one0", "1")])
def test_timed_path_fused_backward_vs_oracle(Bn, T, D, split, force):
    _timed_vs_oracle(Bn, T, D, None, split, compact_min=None, seq_backward=force)


@pytest.mark.parametrize("Bn,T,split", [(256, 50, "mixed"), (300, 40, "mixed"), (64, 33, "mixed"), (200, 50, "one0"), (256, 64, "all1")])
@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_n_split_fused_backward_is_bit_identical_to_the_strip_build(Bn, T, split, compute):
    """amid_sas_seq_bwd_f32's two builds (csrc/sasrec_strip.hip seq_bwd_kernel: a wave per 16-row strip; csrc/sasrec_seqn_bwd.hip: two waves
    per strip with half the columns each, a wave per head in the attention core) sum every product and every row / column sum in the
    same order: the step's dense gradients, the saved gradient tensors the weight gradients read and the table-row gradients agree bit
    for bit (the strip build itself is held to the oracle above)."""
    from amid_amd._lib import lib
    L = lib()
    if not L.value("amid_diag_variants"):
        pytest.skip("the four-strip N-split backward is built into the diagnostic library only (profiles/tools/build_diag.sh, AMID_LIB_PATH)")
    D, hid, n_items = 128, 32, 3000
    P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=300 + D + Bn)
    batch = split_batch(Bn, T, n_items, seed=Bn + T, split=split)
    seed, step = 21, 4
    got = {}
    prev = L.value("amid_sas_seq_bwd_variant", -1)
    try:
        for v in (1, 2):
            L.value("amid_sas_seq_bwd_variant", v)
            eng = make_engine(P, T, seed=seed, compute=compute)
            eng.SEQ_BACKWARD = "1"
            pl = eng.plan(Bn, T, 2, need_grad=True)
            for ts in (pl.dq_l, pl.dk_l, pl.dv_l, pl.dpre1, pl.dpre2, pl.dr):       # rows outside the live sequences are never written
                for t_ in ts:
                    t_.zero_()
            pl.dxbuf.zero_()
            timed_local_grads(eng, pl, batch, step, seed)
            assert pl.seq_bwd_used
            n = int(pl.n_uniq.item())
            got[v] = dict(rows=pl.uniq_grad[:n].clone(), dx=pl.dxbuf.clone(), **{name: eng.dense.view(name, eng.dense.grad).clone() for name in eng.dense.slots},
                          **{f"{k}{l}": getattr(pl, k)[l].clone() for k in ("dq_l", "dk_l", "dv_l", "dpre1", "dpre2", "dr") for l in (0, 1)})
    finally:
        L.value("amid_sas_seq_bwd_variant", prev)
    bad = {name: float((got[2][name] - want).abs().max()) for name, want in got[1].items() if not torch.equal(got[2][name], want)}
    assert all(bool(torch.isfinite(t_).all()) for t_ in got[1].values())
    assert not bad, " ".join(f"{k}:{v:.1e}" for k, v in bad.items())


@pytest.mark.parametrize("Bn,T", [(512, 50), (256, 20)])
def test_timed_path_bf16_products_vs_fp32_oracle(Bn, T):
    """compute="bf16" (BASELINE.json configs[2]: batch 512, bf16 with an fp32 reference tolerance check) THROUGH THE TIMED PATH: the
    fused step over the live sequences with the forward's twelve projection products on the bf16 matrix cores
    (amid_sas_seq_fwd_bf16w_f32), the strip backward's data-gradient products and the weight gradients' products too (bf16 operands,
    fp32 accumulation; LayerNorm, the attention core, residuals, dropout, the partial sums and Adam stay fp32): own logits within 2e-2
    relative of the fp32 oracle (the bar SURVEY.md section 8(c) sets), the loss within 2e-3, every gradient tensor within 2 x the relative
    L2 error measured for its kind (BF16_GRAD_BARS) -- and not fp32-exact (the mode is on).  The relu decisions are the GPU's (bf16
    rounding flips pre-activations near zero)."""
    D, hid, n_items = 128, 32, 3000
    P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=300 + D + Bn)
    batch = split_batch(Bn, T, n_items, seed=Bn + T, split="mixed")
    seed, step = 21, 4
    masks = orc.philox_masks_sasrec(Bn, T, D, seed=seed, step=step)
    eng = make_engine(P, T, seed=seed, compute="bf16")
    pl = eng.plan(Bn, T, 2, need_grad=True)
    assert pl.strip and eng.live_forward_ok(pl)
    timed_local_grads(eng, pl, batch, step, seed)
    keep = gpu_relu_keep(eng, pl, batch)
    loss, (p1, p2), grads = orc.loss_and_grads("sasrec", P, batch, masks, relu_keep=keep)
    assert abs(float(pl.loss.item()) - float(loss)) < 2e-3
    dom = batch["domain_id"]
    own = torch.where(dom[:, None] == 0, pl.p1.cpu(), pl.p2.cpu())
    want = torch.where(dom[:, None] == 0, p1, p2)
    e = relmax(own, want)
    log(f"timed bf16 B={Bn} T={T}: own logits relmax {e:.3e}")
    assert 1e-5 < e < 2e-2
    # every gradient tensor against ITS OWN bar: 2 x the relative L2 error measured for its kind on MI355X (BF16_GRAD_BARS; the
    # measured values are logged to gpurun_out/parity.log on every run) -- a dropped bias-gradient term, a wrong fragment layout in one of
    # the six transposed bf16 weight images or a skipped k-step of the bf16 weight-gradient kernel moves a tensor by O(1), far outside
    measured = {}
    for name in eng.dense.slots:
        if name.endswith("in_proj_bias"):      # (the key third of it is analytically zero: checked by kind below on the q and v thirds)
            n3 = grads[name].numel() // 3
            got3, want3 = eng.dense.view(name, eng.dense.grad).cpu(), grads[name]
            sel = torch.cat([torch.arange(0, n3), torch.arange(2 * n3, 3 * n3)])
            e2 = rel_l2(got3[sel], want3[sel])
        else:
            e2 = rel_l2(eng.dense.view(name, eng.dense.grad), grads[name])
        kind = bf16_grad_kind(name)
        measured[kind] = max(measured.get(kind, 0.0), e2)
        assert e2 < BF16_GRAD_BARS[kind], (name, kind, e2, BF16_GRAD_BARS[kind])
    tg = dense_table_grad(eng, pl)
    e2 = rel_l2(tg, grads["item_emb_layer.emb_item.weight"])
    measured["table"] = e2
    log(f"timed bf16 B={Bn} T={T}: grad rel L2 by kind " + " ".join(f"{k}:{v:.2e}" for k, v in sorted(measured.items())))
    assert e2 < BF16_GRAD_BARS["table"]


def bf16_grad_kind(name: str) -> str:
    if name.startswith("predictModule"):
        return "scorer"
    for key, kind in (("pos_emb", "pos_emb"), ("in_proj_weight", "in_proj_w"), ("in_proj_bias", "in_proj_b"), ("out_proj.weight", "out_proj_w"),
                      ("out_proj.bias", "out_proj_b"), ("conv1.weight", "conv1_w"), ("conv1.bias", "conv1_b"), ("conv2.weight", "conv2_w"),
                      ("conv2.bias", "conv2_b"), ("attention_layernorms", "ln1"), ("forward_layernorms", "ln2"), ("last_layernorm", "ln_last")):
        if key in name:
            return kind
    raise KeyError(name)


# 2 x the largest relative L2 error of a gradient tensor of the kind against the fp32 oracle, measured on MI355X over both shapes of
# test_timed_path_bf16_products_vs_fp32_oracle (gpurun_out/parity.log, round 4: B 512 x T 50 is the larger -- conv1_b 2.8e-2, conv1_w 4.0e-2,
# conv2_b 2.1e-2, conv2_w 2.9e-2, in_proj_b 3.5e-2, in_proj_w 4.2e-2, ln1 5.6e-2, ln2 4.0e-2, ln_last 2.0e-2, out_proj_b 2.4e-2,
# out_proj_w 3.3e-2, pos_emb 6.2e-2, scorer 3.2e-2, table 3.7e-2; twelve bf16 products forward and twelve backward under a dropout
# scale of 2).  A term missing from a sum or a wrong operand image moves its tensor by tens of percent.
BF16_GRAD_BARS = {"conv1_b": 5.6e-2, "conv1_w": 7.9e-2, "conv2_b": 4.2e-2, "conv2_w": 5.8e-2, "in_proj_b": 6.9e-2, "in_proj_w": 8.4e-2,
                  "ln1": 1.1e-1, "ln2": 8.0e-2, "ln_last": 4.0e-2, "out_proj_b": 4.9e-2, "out_proj_w": 6.7e-2, "pos_emb": 1.2e-1,
                  "scorer": 6.3e-2, "table": 7.3e-2}


# the folded bf16 step (test_bf16_step_on_a_pool_takes_the_folded_launches): 2 x the relative L2 error of a gradient tensor of the kind against the fp32
# oracle at ITS shapes -- B 256 x T 50 is the larger: half the samples of BF16_GRAD_BARS' B 512, so sqrt(2) more rounding noise per tensor; the
# unfolded bf16 step measures the same values there to three digits (conv1_b 6.91e-2 against 6.92e-2, pos_emb 7.83e-2 both, ...: the folded
# step multiplies the same bf16-rounded operands), gpurun_out/parity.log "folded bf16" / "timed bf16 B=256 T=50"
BF16_FOLD_GRAD_BARS = {"conv1_b": 1.4e-1, "conv1_w": 1.5e-1, "conv2_b": 1.15e-1, "conv2_w": 1.33e-1, "in_proj_b": 1.27e-1, "in_proj_w": 1.45e-1,
                       "ln1": 1.34e-1, "ln2": 1.25e-1, "ln_last": 1.0e-1, "out_proj_b": 1.18e-1, "out_proj_w": 1.32e-1, "pos_emb": 1.57e-1,
                       "scorer": 8.4e-2, "table": 1.0e-1}


def timed_pool_step(eng, pl, batch, step, seed):
    """Step `step` as bench.py runs it: the batch resident in an input pool, the WHOLE step (enqueue_train_step: with the pool the
    step's head and tail are folded -- SasrecEngine.FUSED_TAIL -- and the segment reduce is finished inside the optimizer launch).  The
    gradients of the step stay readable afterwards (dense.grad, uniq_grad); the oracle differentiates at the parameters before the step."""
    eng.set_step(step - 1, seed)
    cu = {k: v.cuda() for k, v in batch.items()}
    packed = eng.pack_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"])
    eng.set_input_pool(pl, torch.stack([packed, packed]))
    eng.enqueue_train_step(pl)
    eng.sync()
    eng.check_index_error(pl)
    assert eng.step == step


def _timed_vs_oracle(Bn, T, D, build, split, compact_min, seq_backward=None, pool=False):
    hid, n_items = 32, 3000
    P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=300 + D + Bn)
    batch = split_batch(Bn, T, n_items, seed=Bn + T, split=split)
    seed, step = 21, 4
    masks = orc.philox_masks_sasrec(Bn, T, D, seed=seed, step=step)
    eng = make_engine(P, T, seed=seed)
    if compact_min is not None:
        eng.COMPACT_MIN_IDX = compact_min
    if seq_backward is not None:
        eng.SEQ_BACKWARD = seq_backward
    pl = eng.plan(Bn, T, 2, need_grad=True)
    if pool:
        timed_pool_step(eng, pl, batch, step, seed)
        assert pl.tail2, "the shape was chosen to take the folded step"
    else:
        timed_local_grads(eng, pl, batch, step, seed)
    if seq_backward is not None:
        assert pl.seq_bwd_used == (seq_backward == "1")
    assert pool or (pl.compact == eng.compact_ok(pl) and (compact_min != 0 or D != 128 or pl.compact))
    # the encoder's GEMM chains of the fused step run as strip kernels over the live sequences (csrc/sasrec_strip.hip); engines
    # built without them fall back to the live-row builds of the row-tile kernels named in the case
    assert pl.strip or build is None or pl.rt_suffix_v == build
    keep = gpu_relu_keep(eng, pl, batch)
    taps = {}
    loss, (p1, p2), grads = orc.loss_and_grads("sasrec", P, batch, masks, relu_keep=keep, taps=taps)
    flips = [(taps[s].get(f"relu_flip{l}", 0.0), taps[s].get(f"relu_nflip{l}", 0)) for s in ("sac1", "sac2") for l in (0, 1)]
    log(f"timed B={Bn} T={T} D={D} {split}: relu decisions that differ from the oracle's own (|h|, count): {flips}")
    assert max(f[0] for f in flips) < 2e-5            # only pre-activations within rounding of the kink may be decided differently
    assert abs(float(pl.loss.item()) - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    dom = batch["domain_id"]
    own = torch.where(dom[:, None] == 0, pl.p1.cpu(), pl.p2.cpu())          # the logits the loss reads
    want = torch.where(dom[:, None] == 0, p1, p2)
    assert relmax(own, want) < 1e-4
    check_grads(f"timed B={Bn} T={T} D={D} {split}", eng, pl, grads, 2e-4, 5e-5)


# The step as bench.py times it (input pool, D 128, T > 32, strips): twelve launches -- amid_step_head_f32, the embedding backward on the last
# strip launch, the position rows' gradients in the gradient tail, the segment reduce finished inside the optimizer launch.  The domain
# splits leave a domain without any live sequence (all0 / all1) or with one (one0); B 1100 = the pad run spans ~800 chunks; B 5 = a single
# chunk-crossing run at most.
@pytest.mark.parametrize("Bn,T,split", [(256, 50, "mixed"), (256, 50, "all0"), (256, 50, "all1"), (200, 50, "one0"), (250, 40, "mixed"), (300, 40, "mixed"),
                                        (512, 50, "mixed"), (64, 33, "mixed"), (1100, 50, "mixed"), (37, 47, "mixed"), (5, 64, "all1"), (130, 64, "one0"),
                                        (256, 20, "mixed"), (200, 32, "one0"), (64, 17, "mixed"), (300, 24, "all1")])      # (16 < T <= 32: folded since round 6)
def test_timed_path_folded_step_vs_oracle(Bn, T, split):
    _timed_vs_oracle(Bn, T, 128, None, split, compact_min=None, pool=True)


@pytest.mark.parametrize("Bn,T,split,n_items", [(256, 50, "mixed", 3000), (200, 50, "one0", 3000), (64, 33, "all0", 3000),
                                                (256, 50, "mixed", 1_200_000), (700, 40, "mixed", 4_300_000)])
@pytest.mark.parametrize("use_graph", [False, True])
def test_folded_step_matches_the_fifteen_launch_step(Bn, T, split, n_items, use_graph):
    """SasrecEngine.FUSED_TAIL on / off over the same pool (tables of 2^20 rows and more: the riders' sort in 4 096 bins, round 6): the step's inputs come out bit-identical (mirrored batch image, index list, live
    list), the first step's loss too (same forward), its gradients to rounding (the compact list's chunks cut the runs elsewhere, the
    position rows are summed in another order), and five steps -- rows that lag, rows that come back -- leave the same parameters to
    rounding."""
    D, hid, K = 128, 32, 5
    P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=11 + Bn)
    batches = [split_batch(Bn, T, n_items, seed=900 + t, split=split) for t in range(3)]
    for t in (1, 2):                  # rows 1..40 only appear in the first batch: they lag and are caught up when the pool wraps
        batches[t]["seq_d1"] = torch.where(batches[t]["seq_d1"] == n_items - 1, batches[t]["seq_d1"], batches[t]["seq_d1"].clamp(min=41))
        batches[t]["seq_d2"] = torch.where(batches[t]["seq_d2"] == n_items - 1, batches[t]["seq_d2"], batches[t]["seq_d2"].clamp(min=41))
    out = {}
    for fused in (False, True):
        eng = make_engine(P, T, lr=1e-3, seed=77)
        eng.FUSED_TAIL = fused
        pl = eng.plan(Bn, T, 2, need_grad=True)
        packed = []
        for b in batches:
            cu = {k: v.cuda() for k, v in b.items()}
            packed.append(eng.pack_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"]))
        eng.set_input_pool(pl, torch.stack(packed))
        if use_graph:
            eng.capture_train_step(pl)
        rec = dict(loss=[], first={})
        for t in range(K):
            if use_graph:
                eng.replay_train_step(pl)
            else:
                eng.enqueue_train_step(pl)
            eng.sync()
            assert pl.tail2 == fused
            assert torch.equal(pl.in_pack, packed[t % 3])
            rec["loss"].append(float(pl.loss.item()))
            if t == 0:
                U = int(pl.n_uniq.item())                # the step's sort: the distinct rows of the list it reduces over, ascending
                src = pl.idx_c[:pl.n_compact] if (pl.tail2 or getattr(pl, "compact", False)) else pl.idx_all
                assert torch.equal(pl.uniq_ids[:U].long(), torch.unique(src.long()))
                rec["first"] = dict(idx=pl.idx_all.clone(), live=pl.live.clone(), table=dense_table_grad(eng, pl),
                                    **{name: eng.dense.view(name, eng.dense.grad).clone() for name in eng.dense.slots})
        eng.check_index_error(pl)
        eng.flush_table()
        eng.sync()
        rec["params"] = {k: v.cpu().clone() for k, v in eng.state_dict().items()}
        out[fused] = rec
    a, b = out[False], out[True]
    assert torch.equal(a["first"]["idx"], b["first"]["idx"]) and torch.equal(a["first"]["live"], b["first"]["live"])
    assert a["loss"][0] == b["loss"][0]
    for name, want in a["first"].items():
        if name in ("idx", "live"):
            continue
        e = relmax(b["first"][name], want)
        assert e < 2e-6, (name, e)
    for t in range(K):
        assert abs(a["loss"][t] - b["loss"][t]) < 2e-5, (t, a["loss"], b["loss"])
    # Adam divides by sqrt(v): an element whose gradient sits within rounding of zero (a cancelling sum over the pad row's ~10 k positions)
    # takes a step of +lr in one summation order and -lr in the other -- in the reference itself -- and Adam's per-element normalisation
    # turns the 2e-6 (of the tensor's largest entry) the first step's gradients differ by into percents of lr on every element whose
    # gradient is small: a max-abs bar over free-running parameters is ill-posed (see test_timed_path_trajectory_graph_replay_vs_oracle,
    # which holds each path to the oracle's Adam driven by the path's own gradients).  Held here: the L2 distance; the count of elements
    # more than 2e-5 apart is logged (measured: 0.1 % of the table, 7 % of the position rows after five steps at lr 1e-3).
    for k, want in a["params"].items():
        got = b["params"][k]
        far = int(((got - want).abs() > 2e-5).sum())
        log(f"folded vs fifteen-launch step B={Bn} T={T} {split} graph={use_graph} {k}: rel l2 {rel_l2(got, want):.2e}, {far} of {want.numel()} elements apart")
        assert rel_l2(got, want) < 1e-3, (k, rel_l2(got, want), far)


@pytest.mark.parametrize("Bn,T,split", [(256, 50, "mixed"), (200, 64, "one0"), (37, 33, "all0"), (300, 40, "mixed"), (256, 20, "mixed"), (130, 32, "one0"),
                                        (64, 17, "all1")])
@pytest.mark.parametrize("use_graph", [False, True])
def test_head_on_the_forward_workgroups_is_bit_identical_to_its_own_launch(Bn, T, split, use_graph):
    """SasrecEngine.HEAD_ON_FWD on / off over the same pool: amid_sas_seq_fwd_split_lnstat_head_f32 runs the head's code on the rows the
    forward's workgroup still holds -- the same operations in the same order as amid_head_fwd_bwd_own_vec_f32 on the stored rows -- so the
    losses, logits, every gradient of the first step and the parameters after five steps agree BIT FOR BIT, and the step is one launch
    shorter."""
    from amid_amd._lib import lib
    D, hid, n_items, K = 128, 32, 3000, 5
    P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=5 + Bn)
    batches = [split_batch(Bn, T, n_items, seed=400 + t, split=split) for t in range(3)]
    out = {}
    for on in (False, True):
        eng = make_engine(P, T, lr=1e-3, seed=78)
        eng.HEAD_ON_FWD = on
        pl = eng.plan(Bn, T, 2, need_grad=True)
        packed = []
        for b in batches:
            cu = {k: v.cuda() for k, v in b.items()}
            packed.append(eng.pack_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"]))
        eng.set_input_pool(pl, torch.stack(packed))
        names = []
        L = lib()
        orig = L.call
        L.call = lambda name, *a: (names.append(name), orig(name, *a))[1]        # (a spy on the C-ABI calls of this one step)
        try:
            eng.enqueue_train_step(pl)
            eng.sync()
        finally:
            del L.call
        assert pl.tail2
        with_head = {"amid_sas_seq_fwd_split_lnstat_head_f32", "amid_sas_seq_fwd_gather_head_f32"}      # (the latter: the gather as the forward's prologue too)
        assert bool(with_head & set(names)) == on and ("amid_head_fwd_bwd_own_vec_f32" in names) == (not on), names
        rec = dict(n_calls=len(names), loss=[float(pl.loss.item())], p1=pl.p1.clone(), p2=pl.p2.clone(), u=pl.u.clone(),
                   table=dense_table_grad(eng, pl), **{name: eng.dense.view(name, eng.dense.grad).clone() for name in eng.dense.slots})
        if use_graph:
            eng.capture_train_step(pl)
        for t in range(1, K):
            if use_graph:
                eng.replay_train_step(pl)
            else:
                eng.enqueue_train_step(pl)
            eng.sync()
            rec["loss"].append(float(pl.loss.item()))
        eng.check_index_error(pl)
        eng.flush_table()
        eng.sync()
        rec["params"] = {k: v.cpu().clone() for k, v in eng.state_dict().items()}
        out[on] = rec
    a, b = out[False], out[True]
    assert b["n_calls"] == a["n_calls"] - 1
    assert a["loss"] == b["loss"], (a["loss"], b["loss"])
    for name, want in a.items():
        if name in ("n_calls", "loss", "params"):
            continue
        assert torch.equal(b[name], want), (name, relmax(b[name], want))
    for k, want in a["params"].items():
        assert torch.equal(b["params"][k], want), (k, rel_l2(b["params"][k], want))


@pytest.mark.parametrize("Bn,T,split,crowd", [(256, 50, "mixed", False), (200, 64, "one0", False), (37, 33, "all0", False), (300, 40, "mixed", True),
                                               (1100, 50, "mixed", True)])
@pytest.mark.parametrize("use_graph", [False, True])
def test_optimizer_in_the_gradient_tail_is_bit_identical_to_its_own_launch(Bn, T, split, crowd, use_graph):
    """SasrecEngine.FUSED_OPT on / off over the same pool: amid_grad_tail_opt_f32 makes the sums of amid_grad_tail_live_f32 and applies the
    Adam steps of amid_optimizer_step_spans_f32 -- the same additions in the same order, the same update arithmetic -- in ONE launch (the
    chunk-crossing runs by the chunk block that takes the last ticket, their pieces read with agent scope), so every gradient of the first
    step and the parameters after K steps (rows that lag and come back: a three-batch pool walked K = 12 times round under graph replay)
    agree BIT FOR BIT, step after step.  crowd: ids drawn from a few rows, so that many runs cross chunk borders (the last block's owner list
    is long), beside the pad row's."""
    from amid_amd._lib import lib
    D, hid, n_items, K = 128, 32, 3000, 12
    P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=15 + Bn)
    batches = [split_batch(Bn, T, n_items, seed=700 + t, split=split) for t in range(3)]
    if crowd:
        for b in batches:
            for k in ("seq_d1", "seq_d2"):
                b[k] = torch.where(b[k] == n_items - 1, b[k], 1 + b[k] % 23)
    out = {}
    for on in (False, True):
        eng = make_engine(P, T, lr=1e-3, seed=79)
        eng.FUSED_OPT = on
        pl = eng.plan(Bn, T, 2, need_grad=True)
        packed = []
        for b in batches:
            cu = {k: v.cuda() for k, v in b.items()}
            packed.append(eng.pack_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"]))
        eng.set_input_pool(pl, torch.stack(packed))
        names = []
        L = lib()
        orig = L.call
        L.call = lambda name, *a: (names.append(name), orig(name, *a))[1]        # (a spy on the C-ABI calls of this one step)
        try:
            eng.enqueue_train_step(pl)
            eng.sync()
        finally:
            del L.call
        assert pl.tail2
        assert ("amid_grad_tail_opt_f32" in names) == on and ("amid_optimizer_step_spans_f32" in names) == (not on), names
        rec = dict(n_calls=len(names), loss=[float(pl.loss.item())], table=dense_table_grad(eng, pl),
                   **{name: eng.dense.view(name, eng.dense.grad).clone() for name in eng.dense.slots})
        rec["after1"] = {k: v.clone() for k, v in eng.state_dict().items()}
        if use_graph:
            eng.capture_train_step(pl)
        for t in range(1, K):
            if use_graph:
                eng.replay_train_step(pl)
            else:
                eng.enqueue_train_step(pl)
            eng.sync()
            rec["loss"].append(float(pl.loss.item()))
        eng.check_index_error(pl)
        assert int(pl.tail_ticket[0].item()) == 0 if on else True
        eng.flush_table()
        eng.sync()
        rec["params"] = {k: v.cpu().clone() for k, v in eng.state_dict().items()}
        out[on] = rec
    a, b = out[False], out[True]
    assert b["n_calls"] == a["n_calls"] - 1
    assert a["loss"] == b["loss"], (a["loss"], b["loss"])
    for name, want in a.items():
        if name in ("n_calls", "loss", "params", "after1"):
            continue
        assert torch.equal(b[name], want), (name, relmax(b[name], want))
    for k, want in a["after1"].items():
        assert torch.equal(b["after1"][k], want), ("after one step", k)
    for k, want in a["params"].items():
        assert torch.equal(b["params"][k], want), (k, rel_l2(b["params"][k], want))


@pytest.mark.parametrize("Bn,T,split,head_on_fwd", [(256, 50, "mixed", True), (200, 64, "one0", True), (37, 33, "all0", True), (300, 40, "mixed", False),
                                                    (256, 20, "mixed", True), (100, 29, "one0", True)])
@pytest.mark.parametrize("use_graph", [False, True])
def test_gather_in_the_forward_prologue_is_bit_identical_to_its_own_launch(Bn, T, split, head_on_fwd, use_graph):
    """SasrecEngine.GATHER_ON_FWD on / off over the same pool: the forward's workgroups build their own input rows -- table[id] + pos, the row's
    Philox keep bits, the == 0 mask: the gather K1's arithmetic -- and the step head's last workgroups write the weight images K1's riders
    wrote.  The gathered rows, the mask bytes, the losses, every gradient of the first step and the parameters after six steps agree BIT FOR
    BIT, and the step is one launch shorter.  (head_on_fwd False: the folded step's forward without the head on its tail.)"""
    from amid_amd._lib import lib
    D, hid, n_items, K = 128, 32, 3000, 6
    P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=25 + Bn)
    for d in (1, 2):          # x = table + pos exactly zero on a whole row of either encoder: the == 0 mask fires
        P[f"sac{d}.pos_emb.weight"][3] = -P["item_emb_layer.emb_item.weight"][7]
    batches = [split_batch(Bn, T, n_items, seed=300 + t, split=split) for t in range(3)]
    for b in batches:
        b["seq_d1"][:, 3] = 7
        b["seq_d2"][:, 3] = 7
    out = {}
    for on in (False, True):
        eng = make_engine(P, T, lr=1e-3, seed=80)
        eng.GATHER_ON_FWD = on
        eng.HEAD_ON_FWD = head_on_fwd
        pl = eng.plan(Bn, T, 2, need_grad=True)
        packed = []
        for b in batches:
            cu = {k: v.cuda() for k, v in b.items()}
            packed.append(eng.pack_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"]))
        eng.set_input_pool(pl, torch.stack(packed))
        names = []
        L = lib()
        orig = L.call
        L.call = lambda name, *a: (names.append(name), orig(name, *a))[1]        # (a spy on the C-ABI calls of this one step)
        try:
            eng.enqueue_train_step(pl)
            eng.sync()
        finally:
            del L.call
        assert pl.tail2
        assert any(n.startswith("amid_sas_seq_fwd_gather") for n in names) == on and any(n.startswith("amid_embed_fwd") for n in names) == (not on), names
        live = pl.live[:Bn].long()
        n0 = int(pl.live[Bn].item())
        M = Bn * T
        rows = torch.cat([(g * M + live[(0 if g == 0 else n0):(n0 if g == 0 else Bn)][:, None] * T + torch.arange(T, device="cuda")[None, :]).reshape(-1) for g in (0, 1)])
        rec = dict(n_calls=len(names), loss=[float(pl.loss.item())], x0=pl.xg[rows].clone(), tm=pl.tmq[rows].clone(),
                   items=pl.xg[2 * M:2 * M + 2 * Bn].clone(), table=dense_table_grad(eng, pl),
                   **{name: eng.dense.view(name, eng.dense.grad).clone() for name in eng.dense.slots})
        if use_graph:
            eng.capture_train_step(pl)
        for t in range(1, K):
            if use_graph:
                eng.replay_train_step(pl)
            else:
                eng.enqueue_train_step(pl)
            eng.sync()
            rec["loss"].append(float(pl.loss.item()))
        eng.check_index_error(pl)
        eng.flush_table()
        eng.sync()
        rec["params"] = {k: v.cpu().clone() for k, v in eng.state_dict().items()}
        out[on] = rec
    a, b = out[False], out[True]
    assert b["n_calls"] == a["n_calls"] - 1
    assert int(a["tm"].sum()) > 0, "the fixture's zero row did not reach the mask"
    assert a["loss"] == b["loss"], (a["loss"], b["loss"])
    for name, want in a.items():
        if name in ("n_calls", "loss", "params"):
            continue
        assert torch.equal(b[name], want), (name, relmax(b[name].float(), want.float()))
    for k, want in a["params"].items():
        assert torch.equal(b["params"][k], want), (k, rel_l2(b["params"][k], want))


def test_timed_path_real_tokenised_batches_vs_oracle():
    """BASELINE.json configs[1] on the DATA bench.py times, not only its shape: the first two batches of cloth_sport_train75 as the
    reference's own DualDomainSeqDataset tokenised them (tests/golden/tok_cloth_sport_train75.npz: its left-padding, its pad id 447 411,
    its negative draw), batch 256, seq 50, dim 128, the reference's 894 820-row table, fed through the input pool and the folded step --
    each step's loss, own logits, every dense gradient and the table-row gradients against the oracle at the parameters the step started
    from.  The oracle runs on the table remapped to the rows the batch touches (a dense gradient of the whole table is 458 MB of zeros)."""
    from amid_amd.dataset_seq import DeviceBatches, DualDomainSeqDataset
    from amid_amd.engine import SasrecEngine
    Bn, T, D, hid, n_rows = 256, 50, 128, 32, 2 * 447410
    seed, lr = 1234, 5e-4
    root = os.path.dirname(os.path.abspath(__file__))
    ds = DualDomainSeqDataset.from_tokenised(os.path.join(root, "golden", "tok_cloth_sport_train75.npz"))
    assert ds.seq_len == T and ds.pad_id == 447411
    ep = DeviceBatches(ds, Bn, shuffle=False, device="cuda", seed=0, negatives="fixture").epoch_tensors()
    Pd = orc.random_params(orc.sasrec_param_shapes(8, D, T, hid), seed=31)
    eng = SasrecEngine(n_rows, D, T, hid, lr=lr, seed=seed)
    g = torch.Generator(device="cuda").manual_seed(2)
    eng.table.copy_(torch.randn(n_rows, D, generator=g, device="cuda"))
    with torch.no_grad():
        for name in eng.dense.slots:
            eng.dense.view(name).copy_(Pd[name].cuda())
    pl = eng.plan(Bn, T, 2, need_grad=True)
    eng.set_input_pool(pl, eng.pack_epoch(pl, ep["i_node"][:2], ep["neg_samples"][:2], ep["seq_d1"][:2], ep["seq_d2"][:2], ep["label"],
                                          ep["domain_id"][:2]))
    keys = ("i_node", "neg_samples", "seq_d1", "seq_d2")
    for t in (1, 2):
        batch = {k: ep[k][t - 1].cpu() for k in keys + ("domain_id",)}
        batch["neg_samples"] = batch["neg_samples"].reshape(Bn, -1)
        batch["label"] = ep["label"].cpu().reshape(Bn, -1).float()
        pad_share = float((batch["seq_d1"] == ds.pad_id).float().mean() + (batch["seq_d2"] == ds.pad_id).float().mean()) / 2
        assert pad_share > 0.8                          # the real data's skew: ~89 % of the positions are the pad row
        ids = torch.cat([batch[k].reshape(-1) for k in keys])
        uniq, inv = torch.unique(ids, return_inverse=True)
        eng.flush_table()
        eng.sync()
        Ps = {name: eng.dense.view(name).cpu().clone() for name in eng.dense.slots}
        Ps["item_emb_layer.emb_item.weight"] = eng.table[uniq.cuda()].cpu()
        sub, o = dict(batch), 0
        for k in keys:
            n = batch[k].numel()
            sub[k] = inv[o:o + n].reshape(batch[k].shape); o += n
        eng.enqueue_train_step(pl)
        eng.sync()
        eng.check_index_error(pl)
        assert pl.tail2 and eng.step == t
        masks = orc.philox_masks_sasrec(Bn, T, D, seed=seed, step=t)
        taps = {}
        loss, (p1, p2), grads = orc.loss_and_grads("sasrec", Ps, sub, masks, relu_keep=gpu_relu_keep(eng, pl, batch), taps=taps)
        assert max(taps[s_].get(f"relu_flip{l}", 0.0) for s_ in ("sac1", "sac2") for l in (0, 1)) < 2e-5
        assert abs(float(pl.loss.item()) - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
        dom = batch["domain_id"]
        own = torch.where(dom[:, None] == 0, pl.p1.cpu(), pl.p2.cpu())
        assert relmax(own, torch.where(dom[:, None] == 0, p1, p2)) < 1e-4
        worst = worst2 = 0.0
        for name in eng.dense.slots:
            got, want = eng.dense.view(name, eng.dense.grad).cpu().clone(), grads[name].clone()
            if name.endswith("in_proj_bias"):
                n3 = got.numel() // 3
                got[n3:2 * n3] = 0; want[n3:2 * n3] = 0
            e, e2 = relmax(got, want), rel_l2(got, want)
            worst, worst2 = max(worst, e), max(worst2, e2)
            assert e < 2e-4 and e2 < (5e-5 if got.numel() > 8 else 2e-4), (t, name, e, e2)
        U = int(pl.n_uniq.item())
        got_ids, got_rows = pl.uniq_ids[:U].cpu().long(), pl.uniq_grad[:U].cpu()
        want_tab = grads["item_emb_layer.emb_item.weight"]
        where = torch.searchsorted(uniq, got_ids)
        assert bool((uniq[where] == got_ids).all())
        tab = torch.zeros_like(want_tab)
        tab[where] = got_rows
        live_ids = torch.cat([torch.where(dom[:, None] == 0, sub["seq_d1"], sub["seq_d2"]).reshape(-1), sub["i_node"].reshape(-1), sub["neg_samples"].reshape(-1)])
        assert set(where.tolist()) == set(live_ids.tolist())             # the step's list = the ids of the own-domain sequences and the items
        e, e2 = relmax(tab, want_tab), rel_l2(tab, want_tab)
        log(f"real batch {t} of cloth_sport_train75 (pads {pad_share:.3f}, {U} rows in the step's list): loss {float(pl.loss.item()):.6f} "
            f"oracle {float(loss):.6f}; worst dense grad relmax {worst:.3e} l2 {worst2:.3e}; table relmax {e:.3e} l2 {e2:.3e}")
        assert e < 2e-4 and e2 < 5e-5


def _fuzz_cases(seed: int, n: int):
    """The draw of profiles/tools/probe/fuzz_timed.py: (B, T, D, domain split, fused backward forced on / off / auto)."""
    import random
    rng = random.Random(seed)
    cases = []
    for _ in range(n):
        D = rng.choice([64, 128])
        T = rng.choice([1, 2, 7, 15, 16, 17, 20, 31, 32, 33, 40, 47, 48, 49, 50, 63, 64])
        B = rng.choice([1, 3, 37, 64, 130, 256, 300])
        split = rng.choice(["mixed", "mixed", "all0", "all1", "one0"])
        force = rng.choice([None, "1", "0"])
        cases.append((B, T, D, split, force))
    return cases


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("case", _fuzz_cases(4, 24), ids=lambda c: "-".join(str(x) for x in c))
def test_timed_path_randomised_shapes_vs_oracle(case):
    """A seeded 24-draw subset of the randomised sweep of profiles/tools/probe/fuzz_timed.py (round 3 ran 160 draws by hand): the timed
    step -- loss, own logits, every dense gradient, the table-row gradients -- against the oracle at random (B, T, D, domain split) with
    the fused per-sequence backward forced on, off or left to the engine.  T = 1 .. 64 crosses every build boundary of the one-launch
    kernels (16 / 32 / 48 rows), B = 1 .. 300 both sides of the CU count, D 64 the head-pair attention."""
    B, T, D, split, force = case
    _timed_vs_oracle(B, T, D, None, split, compact_min=None, seq_backward=force)


@pytest.mark.parametrize("Bn,T,D", [(256, 50, 128), (256, 20, 128), (320, 50, 128)])     # (320: the fused per-sequence backward, side-stream sort)
@pytest.mark.parametrize("pool", [False, True])
def test_timed_path_trajectory_graph_replay_vs_oracle(Bn, T, D, pool):
    """Three whole steps at the headline shape (and the mybank shape) through capture_train_step + replay_train_step -- the thing
    bench.py times -- against the oracle, step by step: the loss and every gradient of the step against the oracle's at the same
    parameters, and the optimizer against the oracle's dense Adam.

    Adam divides by sqrt(v), so an element whose gradient is within ~1e-7 of zero (1 % of the touched table elements at this size)
    turns rounding noise into a visible fraction of lr -- in the reference itself --, and the perturbation then feeds the next
    step: a free-running max-abs comparison of parameters is ill-posed here (the key-bias slice of test_oracle_golden is the same
    effect).  So the oracle's Adam is driven by the gradients the GPU produced: parameters must then agree to rounding after every
    step -- which checks the lazy row updates, their catch-up and the flush exactly --, while the gradients themselves are checked
    against the oracle's own at those parameters."""
    hid, n_items, K, lr, seed = 32, 3000, 3, 1e-3, 4242
    P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=20 + T)
    eng = make_engine(P, T, lr=lr, seed=seed)
    Po = {k: v.clone() for k, v in P.items()}
    opt = orc.DenseAdam(Po, lr=lr)
    pl = eng.plan(Bn, T, 2, need_grad=True)
    batches = [orc.synthetic_batch(Bn, T, n_items - 1, pad_id=n_items - 1, neg=1, seed=500 + t) for t in range(K)]
    for t in (1, 2):                  # rows 1..40 are only touched at step 1: idle rows must keep moving through their momentum
        batches[t]["seq_d1"] = torch.where(batches[t]["seq_d1"] == n_items - 1, batches[t]["seq_d1"], batches[t]["seq_d1"].clamp(min=41))
    if pool:
        packed = []
        for b in batches:
            cu = {k: v.cuda() for k, v in b.items()}
            packed.append(eng.pack_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"]))
        eng.set_input_pool(pl, torch.stack(packed))
    else:
        load(eng, pl, batches[0])
    eng.capture_train_step(pl)
    for t in range(1, K + 1):
        if not pool:
            load(eng, pl, batches[t - 1])
        eng.replay_train_step(pl)
        eng.sync()
        masks = orc.philox_masks_sasrec(Bn, T, D, seed=seed, step=t)
        taps = {}
        loss_o, _, grads = orc.loss_and_grads("sasrec", Po, batches[t - 1], masks, relu_keep=gpu_relu_keep(eng, pl, batches[t - 1]), taps=taps)
        assert max(taps[s_].get(f"relu_flip{l}", 0.0) for s_ in ("sac1", "sac2") for l in (0, 1)) < 2e-5
        log(f"timed traj B={Bn} T={T} pool={pool} step {t}: loss gpu {float(pl.loss.item()):.7f} oracle {float(loss_o):.7f}")
        assert abs(float(pl.loss.item()) - float(loss_o)) < 5e-5
        check_grads(f"timed traj B={Bn} T={T} pool={pool} step {t}", eng, pl, grads, 2e-4, 5e-5)
        g_gpu = {name: eng.dense.view(name, eng.dense.grad).cpu().clone() for name in eng.dense.slots}
        g_gpu["item_emb_layer.emb_item.weight"] = dense_table_grad(eng, pl)
        opt.step(Po, g_gpu)
    eng.check_index_error(pl)
    eng.flush_table()
    eng.sync()
    sd = eng.state_dict()
    worst = 0.0
    for k, v in Po.items():
        d = float((sd[k].cpu() - v).abs().max())
        worst = max(worst, d)
        assert d < 5e-6, (k, d)
    log(f"timed traj B={Bn} T={T} pool={pool}: worst |param diff| after {K} steps {worst:.3e}")


def plain_local_grads(eng, pl, batch, step, seed):
    """The same step through the PLAIN kernels (every sequence of both domains encoded and differentiated)."""
    eng.set_step(step, seed)
    load(eng, pl, batch)
    eng.enqueue_prepare(pl, sparse=True)
    eng.enqueue_forward(pl, train=True, with_loss=True)
    eng.enqueue_backward(pl, train=True)
    eng.sync()


@pytest.mark.parametrize("Bn,T,n_items", [(4096, 50, 10_000_000), (256, 50, 3000), (300, 20, 3000)])
def test_timed_path_equals_plain_path(Bn, T, n_items):
    """BASELINE.json configs[4]'s per-GPU step (B 4096 on a 10 M-row table) and two small shapes: the timed path against the plain
    path of the same engine -- the encoder-input gradient rows of the live sequences to rounding, exact zeros for the others, the reduced table-row and dense gradients to rounding --, and its loss-side logits against
    the oracle on a 64-sample slice (a sample's logits depend on that sample only)."""
    D, hid = 128, 32
    seed, step = 5, 3
    shapes = orc.sasrec_param_shapes(8, D, T, hid)
    P = orc.random_params(shapes, seed=77)
    from amid_amd.engine import SasrecEngine
    eng = SasrecEngine(n_items, D, T, hid, seed=seed)
    g = torch.Generator(device="cuda").manual_seed(1)
    eng.table.copy_(torch.randn(n_items, D, generator=g, device="cuda"))
    with torch.no_grad():
        for name in eng.dense.slots:
            eng.dense.view(name).copy_(P[name].cuda())
    torch.cuda.synchronize()
    gb = torch.Generator().manual_seed(Bn)
    batch = orc.synthetic_batch(Bn, T, n_items - 2, pad_id=n_items - 1, neg=1, seed=9)
    if n_items > 100000:                  # S-uniform: every position a uniform id, no pads
        batch["seq_d1"] = torch.randint(0, n_items, (Bn, T), generator=gb)
        batch["seq_d2"] = torch.randint(0, n_items, (Bn, T), generator=gb)
    pl = eng.plan(Bn, T, 2, need_grad=True)
    timed_local_grads(eng, pl, batch, step, seed)
    M = Bn * T
    dom = batch["domain_id"]
    live = torch.cat((dom == 0, dom == 1)).repeat_interleave(T)              # [2M]
    t_dx = pl.dxg[: 2 * M].cpu().clone()
    t_loss, t_ids_n = float(pl.loss.item()), int(pl.n_uniq.item())
    t_tab = (pl.uniq_ids[:t_ids_n].cpu().clone(), pl.uniq_grad[:t_ids_n].cpu().clone())
    t_dense = eng.dense.grad.cpu().clone()
    t_own = torch.where(dom[:, None] == 0, pl.p1.cpu(), pl.p2.cpu())
    plain_local_grads(eng, pl, batch, step, seed)
    p_dx = pl.dxg[: 2 * M].cpu()
    # (not bit for bit: the timed path's head sums a sequence's T LayerNorm rows in 16 groups, the plain head in 8)
    assert relmax(t_dx[live], p_dx[live]) < 1e-5
    assert float(p_dx[~live].abs().max()) == 0.0 and float(t_dx[~live].abs().max()) == 0.0
    p_n = int(pl.n_uniq.item())
    p_ids, p_rows = pl.uniq_ids[:p_n].cpu(), pl.uniq_grad[:p_n].cpu()
    nz = p_rows.abs().amax(1) > 0                                            # rows only dead positions touch reduce to exact zeros
    tz = t_tab[1].abs().amax(1) > 0
    assert torch.equal(p_ids[nz], t_tab[0][tz])
    assert relmax(t_tab[1][tz], p_rows[nz]) < 2e-6
    assert abs(t_loss - float(pl.loss.item())) < 1e-6
    assert relmax(t_dense, eng.dense.grad) < 2e-5
    assert relmax(t_own, torch.where(dom[:, None] == 0, pl.p1.cpu(), pl.p2.cpu())) < 2e-6
    # oracle on the first 64 samples (ids remapped onto the rows they touch; dropout counters are indexed b-major, so the slice's
    # masks are the first entries of the whole batch's)
    S = 64
    sub = {k: v[:S].clone() for k, v in batch.items()}
    ids = torch.cat([sub[k].reshape(-1) for k in ("i_node", "neg_samples", "seq_d1", "seq_d2")])
    uniq, inv = torch.unique(ids, return_inverse=True)
    Ps = dict(P)
    Ps["item_emb_layer.emb_item.weight"] = eng.table[uniq.cuda()].cpu()
    o = 0
    for k in ("i_node", "neg_samples", "seq_d1", "seq_d2"):
        n = sub[k].numel()
        sub[k] = inv[o:o + n].reshape(sub[k].shape); o += n
    p1, p2 = orc.sasrec_forward(Ps, sub["i_node"], sub["neg_samples"], sub["seq_d1"], sub["seq_d2"],
                                orc.philox_masks_sasrec(S, T, D, seed=seed, step=step))
    want = torch.where(dom[:S, None] == 0, p1, p2)
    assert relmax(t_own[:S], want) < 1e-4


@pytest.mark.timeout(1800)
def test_timed_path_cfg5_batch_every_gradient_vs_oracle():
    """BASELINE.json configs[4]'s per-GPU step at FULL size -- batch 4096, seq 50, dim 128, a 10 M-row table, S-uniform ids (no pads: ~417 k
    index positions, ~210 k distinct rows) -- through the timed path (compact index list, side-stream sort, live sequences) against the
    oracle: loss, own logits, EVERY dense gradient and the table-row gradients.  The oracle runs on the table remapped to the rows the
    batch touches (a dense 10 M-row embedding gradient is 5 GB of zeros); dropout counters are indexed by (row, position), not by id."""
    Bn, T, D, hid, n_items = 4096, 50, 128, 32, 10_000_000
    seed, step = 5, 3
    P = orc.random_params(orc.sasrec_param_shapes(8, D, T, hid), seed=77)
    from amid_amd.engine import SasrecEngine
    eng = SasrecEngine(n_items, D, T, hid, seed=seed)
    g = torch.Generator(device="cuda").manual_seed(1)
    eng.table.copy_(torch.randn(n_items, D, generator=g, device="cuda"))
    with torch.no_grad():
        for name in eng.dense.slots:
            eng.dense.view(name).copy_(P[name].cuda())
    torch.cuda.synchronize()
    gb = torch.Generator().manual_seed(Bn)
    batch = orc.synthetic_batch(Bn, T, n_items - 2, pad_id=n_items - 1, neg=1, seed=9)
    batch["seq_d1"] = torch.randint(0, n_items, (Bn, T), generator=gb)
    batch["seq_d2"] = torch.randint(0, n_items, (Bn, T), generator=gb)
    pl = eng.plan(Bn, T, 2, need_grad=True)
    timed_local_grads(eng, pl, batch, step, seed)
    assert pl.compact                                  # the long-list form of the sparse side
    keys = ("i_node", "neg_samples", "seq_d1", "seq_d2")
    ids = torch.cat([batch[k].reshape(-1) for k in keys])
    uniq, inv = torch.unique(ids, return_inverse=True)
    Ps = dict(P)
    Ps["item_emb_layer.emb_item.weight"] = eng.table[uniq.cuda()].cpu()
    sub = dict(batch)
    o = 0
    for k in keys:
        n = batch[k].numel()
        sub[k] = inv[o:o + n].reshape(batch[k].shape); o += n
    masks = orc.philox_masks_sasrec(Bn, T, D, seed=seed, step=step)
    keep = gpu_relu_keep(eng, pl, batch)
    loss, (p1, p2), grads = orc.loss_and_grads("sasrec", Ps, sub, masks, relu_keep=keep)
    assert abs(float(pl.loss.item()) - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    dom = batch["domain_id"]
    own = torch.where(dom[:, None] == 0, pl.p1.cpu(), pl.p2.cpu())
    assert relmax(own, torch.where(dom[:, None] == 0, p1, p2)) < 1e-4
    worst = worst2 = 0.0
    for name in eng.dense.slots:
        got, want = eng.dense.view(name, eng.dense.grad).cpu().clone(), grads[name].clone()
        if name.endswith("in_proj_bias"):
            n3 = got.numel() // 3
            got[n3:2 * n3] = 0; want[n3:2 * n3] = 0
        e, e2 = relmax(got, want), rel_l2(got, want)
        worst, worst2 = max(worst, e), max(worst2, e2)
        assert e < 2e-4 and e2 < 5e-5, (name, e, e2)
    U = int(pl.n_uniq.item())
    got_ids, got_rows = pl.uniq_ids[:U].cpu().long(), pl.uniq_grad[:U].cpu()
    want_tab = grads["item_emb_layer.emb_item.weight"]                       # [n unique, D], row j <-> id uniq[j]
    tab = torch.zeros_like(want_tab)
    where = torch.searchsorted(uniq, got_ids)
    assert bool((uniq[where] == got_ids).all())          # the step's unique list holds ids of the batch only
    tab[where] = got_rows
    e, e2 = relmax(tab, want_tab), rel_l2(tab, want_tab)
    log(f"cfg5 batch (B 4096, {uniq.numel()} distinct rows): worst dense grad relmax {worst:.3e} l2 {worst2:.3e}; table relmax {e:.3e} l2 {e2:.3e}")
    assert e < 2e-4 and e2 < 5e-5


# ---------------------------------------------------------------------------- kernel level: the three row-tile kernels at several tile heights
def _rand(g, *shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).cuda()


@pytest.mark.parametrize("suf,rpt", [("", 100), ("", 40), ("", 50), ("", 75)])
@pytest.mark.parametrize("Bn,T", [(64, 50), (37, 20), (1100, 8)])
def test_rows_row_tile_kernels_equal_plain_in_every_build(suf, rpt, Bn, T):
    """amid_sas_{ffn_bwd,qkv_bwd,qkv_ffn_bwd}_rows_f32 (tiles over the live sequences through the LDS row map) against the plain
    entry points at the same rows per tile, on inputs whose dead rows carry zero gradients: every per-row output bit-identical on the live
    rows, the LayerNorm partial sums equal to rounding (the rows are grouped into tiles differently)."""
    from amid_amd._lib import lib, ptr_array
    L = lib()
    D, M = 128, Bn * T
    g = torch.Generator().manual_seed(Bn * 31 + T)
    dom = (torch.rand(Bn, generator=g) < 0.5).long()
    livef = torch.cat((dom == 0, dom == 1)).float().repeat_interleave(T).cuda()          # [2M]
    live = livef.bool().cpu()
    s = torch.cuda.current_stream().cuda_stream
    host = (ctypes.c_ubyte * L.value("amid_step_state_bytes"))()
    L.call("amid_step_state_pack", ctypes.addressof(host), 11, 6, 5e-4, 0.9, 0.999, 1e-8)
    st = torch.frombuffer(bytearray(host), dtype=torch.uint8).cuda()
    pa = lambda a, b: ptr_array([a.data_ptr(), b.data_ptr()])          # noqa: E731
    W = lambda: (_rand(g, D, D, scale=0.1), _rand(g, D, D, scale=0.1))  # noqa: E731  per-domain [D, D]
    lnw = (1 + _rand(g, D, scale=0.1), 1 + _rand(g, D, scale=0.1))
    tmq = (torch.rand(2 * M, D // 4, generator=g) < 0.02).to(torch.uint8).cuda() * 5
    tpg = -(-M // rpt)
    dom_d = dom.cuda()

    def inputs(kind):
        mk = lambda: _rand(g, 2 * M, D)                                # noqa: E731
        if kind == "ffn":
            return dict(dxo=mk() * livef[:, None], h=mk().relu(), r=mk(), w1T=W(), w2T=W(), woT=W())
        d = dict(dq=mk() * livef[:, None], dk=mk() * livef[:, None], dv=mk() * livef[:, None], dr=mk() * livef[:, None], x=mk(),
                 wq=W(), wk=W(), wv=W())
        if kind == "qkv_ffn":
            d.update(h=mk().relu(), r=mk(), w1T=W(), w2T=W(), woT=W())
        return d

    def run(kind, i, rows):
        nanbuf = lambda: torch.full((2 * M, D), float("nan"), device="cuda")   # noqa: E731
        part = lambda: torch.full((2 * tpg, 2, D), float("nan"), device="cuda")  # noqa: E731
        hint = (dom_d.data_ptr(), Bn, T) if rows else ()
        sufx = ("_rows" if rows else "") + "_f32" + suf
        if kind == "ffn":
            dpre2, dpre1, dr, d_o, lp = nanbuf(), nanbuf(), nanbuf(), nanbuf(), part()
            L.call("amid_sas_ffn_bwd" + sufx, i["dxo"].data_ptr(), tmq.data_ptr(), i["h"].data_ptr(), i["r"].data_ptr(), pa(*lnw),
                   pa(*i["w1T"]), pa(*i["w2T"]), pa(*i["woT"]), 1e-8, M, D, rpt, 1, st.data_ptr(), 1, 0.5, dpre2.data_ptr(),
                   dpre1.data_ptr(), dr.data_ptr(), d_o.data_ptr(), lp.data_ptr(), 0, *hint, s)
            torch.cuda.synchronize()
            return [dpre2.cpu(), dpre1.cpu(), dr.cpu(), d_o.cpu()], [lp.cpu()]
        if kind == "qkv":
            dx, lp = nanbuf(), part()
            L.call("amid_sas_qkv_bwd" + sufx, i["dq"].data_ptr(), i["dk"].data_ptr(), i["dv"].data_ptr(), i["dr"].data_ptr(),
                   i["x"].data_ptr(), pa(*lnw), pa(*i["wq"]), pa(*i["wk"]), pa(*i["wv"]), 1e-8, M, D, rpt, dx.data_ptr(), lp.data_ptr(),
                   0, *hint, s)
            torch.cuda.synchronize()
            return [dx.cpu()], [lp.cpu()]
        dx, lp, dpre2, dpre1, fdr, d_o, flp = nanbuf(), part(), nanbuf(), nanbuf(), nanbuf(), nanbuf(), part()
        L.call("amid_sas_qkv_ffn_bwd" + sufx, i["dq"].data_ptr(), i["dk"].data_ptr(), i["dv"].data_ptr(), i["dr"].data_ptr(),
               i["x"].data_ptr(), pa(*lnw), pa(*i["wq"]), pa(*i["wk"]), pa(*i["wv"]), 1e-8, M, D, rpt, dx.data_ptr(), lp.data_ptr(),
               tmq.data_ptr(), i["h"].data_ptr(), i["r"].data_ptr(), pa(*lnw), pa(*i["w1T"]), pa(*i["w2T"]), pa(*i["woT"]), 0,
               st.data_ptr(), 1, 0.5, dpre2.data_ptr(), dpre1.data_ptr(), fdr.data_ptr(), d_o.data_ptr(), flp.data_ptr(), 0, *hint, s)
        torch.cuda.synchronize()
        return [dpre2.cpu(), dpre1.cpu(), fdr.cpu(), d_o.cpu()], [lp.cpu(), flp.cpu()]      # d x[l+1] stays in registers: not an output

    for kind in ("ffn", "qkv", "qkv_ffn"):
        inp = inputs(kind)
        rows_plain, parts_plain = run(kind, inp, False)
        rows_hint, parts_hint = run(kind, inp, True)
        for a, b in zip(rows_plain, rows_hint):
            assert torch.isfinite(a).all()
            assert torch.equal(a[live], b[live]), (kind, suf)
        for a, b in zip(parts_plain, parts_hint):
            # [2 * tpg, 2, D]: domain g's slots are [g * tpg, (g + 1) * tpg); unused live-tile slots are zero-filled
            sa = a.reshape(2, tpg, 2, D).double().sum(1)
            sb = b.reshape(2, tpg, 2, D).double().sum(1)
            assert torch.isfinite(sb).all()
            assert float((sa - sb).abs().max()) < 2e-5 * float(sa.abs().max() + 1e-30), (kind, suf)


def test_graphs_of_four_steps_equal_single_step_replays():
    """capture_train_steps: eight pooled steps as two replays of a four-step graph leave bit-identical parameters to eight replays of
    the single-step graph (every step picks its batch by the device step counter either way)."""
    Bn, T, D, hid, n_items, K = 64, 20, 128, 32, 900, 8
    P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=5)
    batches = [orc.synthetic_batch(Bn, T, n_items - 1, pad_id=n_items - 1, neg=1, seed=700 + t) for t in range(K)]
    out = []
    for chunk in (1, 4):
        eng = make_engine(P, T, seed=9)
        pl = eng.plan(Bn, T, 2, need_grad=True)
        packed = []
        for b in batches:
            cu = {k: v.cuda() for k, v in b.items()}
            packed.append(eng.pack_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"]))
        eng.set_input_pool(pl, torch.stack(packed))
        eng.capture_train_step(pl)
        if chunk > 1:
            eng.capture_train_steps(pl, chunk)
        for i in range(0, K, chunk):
            if chunk > 1:
                eng.replay_train_steps(pl, chunk)
            else:
                eng.replay_train_step(pl)
        eng.sync()
        assert eng.step == K
        eng.check_index_error(pl)
        eng.flush_table()
        eng.sync()
        out.append({k: v.cpu().clone() for k, v in eng.state_dict().items()})
    for k in out[0]:
        assert torch.equal(out[0][k], out[1][k]), k


@pytest.mark.parametrize("Bn,T,n_items", [(256, 20, 3000), (200, 20, 60_000), (64, 12, 900_000), (256, 32, 3000), (1100, 20, 3000), (2900, 20, 3000)])
@pytest.mark.parametrize("use_graph", [False, True, 4])
def test_sort_chained_in_the_catchup_launch_equals_the_side_stream_sort(Bn, T, n_items, use_graph):
    """SasrecEngine.SORT_CHAIN (round 6): where the riders have no five launches (the one-launch backward at T <= 32) the step's WHOLE index sort
    runs inside the catch-up launch -- its rider workgroups chain the five phases with a barrier of their own -- instead of a dozen launches on
    a side stream.  Same outputs (the distinct rows ascending, their runs, the positions in list order inside a run), and therefore the same
    step bit for bit: eight steps over a three-batch pool, eager and replayed."""
    D, hid, K = 128, 32, 8
    P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=5 + Bn)
    batches = [split_batch(Bn, T, n_items, seed=700 + t, split="mixed") for t in range(3)]
    out = {}
    for chain in (False, True):
        eng = make_engine(P, T, lr=1e-3, seed=31)
        eng.SORT_CHAIN = chain
        eng.SEQ_BACKWARD = "1"              # (the one-launch backward whatever the batch: B 1100 = 23 rider workgroups, B 2900 = 60)
        eng.COMPACT_MIN_IDX = 1 << 30       # (... over the full index list)
        pl = eng.plan(Bn, T, 2, need_grad=True)
        packed = []
        for b in batches:
            cu = {k: v.cuda() for k, v in b.items()}
            packed.append(eng.pack_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"]))
        eng.set_input_pool(pl, torch.stack(packed))
        if use_graph:
            eng.capture_train_step(pl)
        if use_graph == 4:
            eng.capture_train_steps(pl, 4)       # (what the train loop and bench.py replay: graphs of four steps -- checked after each graph)
        rec = dict(loss=[])
        for t in range(K if use_graph != 4 else K // 4):
            if use_graph == 4:
                eng.replay_train_steps(pl, 4)
            elif use_graph:
                eng.replay_train_step(pl)
            else:
                eng.enqueue_train_step(pl)
            eng.sync()
            assert pl.chain == chain and not pl.tail2 and not pl.riding, "the shape was chosen for the one-launch backward: no riders, no fold"
            src = pl.idx_all.long()
            U = int(pl.n_uniq.item())
            want_u, cnt = torch.unique(src, return_counts=True)
            assert U == want_u.numel() and torch.equal(pl.uniq_ids[:U].long(), want_u), t
            so = pl.seg_off[:U + 1].long()
            assert torch.equal(so[1:] - so[:-1], cnt), t
            assert torch.equal(pl.pos_sorted[:src.numel()].long(), torch.sort(src, stable=True).indices), t
            rec["loss"].append(float(pl.loss.item()))
        eng.check_index_error(pl)
        eng.flush_table()
        eng.sync()
        rec["params"] = {k: v.clone() for k, v in eng.state_dict().items()}
        out[chain] = rec
    assert out[False]["loss"] == out[True]["loss"], (out[False]["loss"], out[True]["loss"])
    for k, want in out[False]["params"].items():
        assert torch.equal(out[True]["params"][k], want), k


@pytest.mark.parametrize("Bn,T", [(256, 50), (200, 20)])
def test_bf16_step_on_a_pool_takes_the_folded_launches(Bn, T):
    """compute = "bf16" on an input pool (round 6, SasrecEngine.BF16_FOLD): the step takes the fp32 step's nine folded launches -- the forward on
    bf16 pieces multiplying ONE piece per operand (amid_sas_seq_fwd_gather_head_p1_f32), the gather in its prologue and the head on its tail -- with
    the three strip launches' data-gradient products on one bf16 piece too (mma mode 1 on the hi plane of the three-plane images).  Held: the launch
    list; loss and own logits at the bf16 mode's bars; every gradient tensor inside the bf16 bars of its kind (BF16_GRAD_BARS) and NOT fp32-exact
    for the tensors behind a strip product (the mode is on); batches beyond BF16_FOLD_MAX_B keep the unfolded bf16 launches."""
    from amid_amd._lib import lib
    D, hid, n_items = 128, 32, 3000
    P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=300 + D + Bn)
    batch = split_batch(Bn, T, n_items, seed=Bn + T, split="mixed")
    seed, step = 21, 4
    masks = orc.philox_masks_sasrec(Bn, T, D, seed=seed, step=step)
    eng = make_engine(P, T, seed=seed, compute="bf16")
    pl = eng.plan(Bn, T, 2, need_grad=True)
    L = lib()
    calls, orig = [], L.call
    def spy(name, *a):
        calls.append((name, a))
        return orig(name, *a)
    L.call = spy
    try:
        timed_pool_step(eng, pl, batch, step, seed)
    finally:
        L.call = orig
    names = [c[0] for c in calls]
    assert pl.tail2 and "amid_sas_seq_fwd_gather_head_p1_f32" in names and "amid_grad_tail_opt_f32" in names, names
    strips = [c for c in calls if c[0] in ("amid_sas_strip_ffn_bwd_sort_f32", "amid_sas_strip_qkv_bwd_sort_scorer_f32", "amid_sas_strip_qkv_bwd_emb_f32")]
    assert len(strips) == 3
    for name, a in strips:       # the precision argument: one bf16 piece
        mode = a[-2] if name != "amid_sas_strip_qkv_bwd_sort_scorer_f32" else a[-11]
        assert mode == 1, (name, mode)
    keep = gpu_relu_keep(eng, pl, batch)
    loss, (p1, p2), grads = orc.loss_and_grads("sasrec", P, batch, masks, relu_keep=keep)
    assert abs(float(pl.loss.item()) - float(loss)) < 2e-3
    dom = batch["domain_id"]
    own = torch.where(dom[:, None] == 0, pl.p1.cpu(), pl.p2.cpu())
    e = relmax(own, torch.where(dom[:, None] == 0, p1, p2))
    assert 1e-5 < e < 2e-2, e                      # (the forward multiplies bf16 operands: the mode's own bars, not fp32-exact)
    worst, measured = 0.0, {}
    for name in eng.dense.slots:
        got, want = eng.dense.view(name, eng.dense.grad).cpu().clone(), grads[name].clone()
        if name.endswith("in_proj_bias"):
            n3 = got.numel() // 3
            got[n3:2 * n3] = 0; want[n3:2 * n3] = 0
        e2 = rel_l2(got, want)
        kind = bf16_grad_kind(name)
        measured[kind] = max(measured.get(kind, 0.0), e2)
        if "attention_layers.0.in_proj_weight" in name:
            worst = max(worst, e2)
    log(f"folded bf16 B={Bn} T={T}: grad rel L2 by kind " + " ".join(f"{k}:{v:.2e}" for k, v in sorted(measured.items())))
    for kind, e2 in measured.items():
        assert e2 < BF16_FOLD_GRAD_BARS[kind], (kind, e2, BF16_FOLD_GRAD_BARS[kind])
    if True:
        assert worst > 1e-5, "layer 0's q / k / v weight gradients sit behind bf16 strip products: fp32-exact means the mode is off"
    assert rel_l2(dense_table_grad(eng, pl), grads["item_emb_layer.emb_item.weight"]) < BF16_FOLD_GRAD_BARS["table"]
    big = make_engine(P, T, seed=seed, compute="bf16")
    big.BF16_FOLD_MAX_B = Bn - 1
    plb = big.plan(Bn, T, 2, need_grad=True)
    timed_pool_step(big, plb, batch, step, seed)
    assert not plb.tail2
