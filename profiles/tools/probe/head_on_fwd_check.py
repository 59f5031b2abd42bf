"""Probe: the folded step's first loss / logits / user vectors with the head on the forward's workgroups (HEAD_ON_FWD) against the head's
own launch, over a few shapes and seeds."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import amid_oracle as orc
from test_gpu_timed_path import make_engine, split_batch

D, hid, n_items = 128, 32, 3000
for Bn, T, split, pseed, bseed in [(64, 33, "all0", 75, 900), (37, 33, "all0", 42, 400), (64, 33, "all0", 42, 400), (37, 33, "all0", 75, 900),
                                   (64, 33, "mixed", 75, 900), (256, 50, "mixed", 261, 400), (256, 50, "mixed", 75, 900)]:
    P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=pseed)
    batches = [split_batch(Bn, T, n_items, seed=bseed + t, split=split) for t in range(3)]
    ref = None
    for mode in ("off", "on"):
        eng = make_engine(P, T, lr=1e-3, seed=77)
        eng.HEAD_ON_FWD = mode == "on"
        eng.HEAD_ON_FWD_KEEPS_X = True
        pl = eng.plan(Bn, T, 2, need_grad=True)
        packed = []
        for b in batches:
            cu = {k: v.cuda() for k, v in b.items()}
            packed.append(eng.pack_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"]))
        eng.set_input_pool(pl, torch.stack(packed))
        eng.enqueue_train_step(pl)
        eng.sync()
        got = dict(loss=float(pl.loss.item()), p1=pl.p1.clone(), u=pl.u.clone(), x=pl.x[2].clone(), dx=pl.dxbuf.clone())
        if ref is None:
            ref = got
            continue
        du = (got["u"] != ref["u"]).view(2, Bn, D).sum(2)
        print(Bn, T, split, pseed, bseed, "loss", repr(ref["loss"]), repr(got["loss"]), "x ne", int((got["x"] != ref["x"]).sum()),
              "u ne", int(du.sum()), "samples with u ne", int((du > 0).sum()), "p1 ne", int((got["p1"] != ref["p1"]).sum()))
        lnw = P["sac1.last_layernorm.weight"]; lnb = P["sac1.last_layernorm.bias"]
        print("   last LN weight range", float(lnw.min()), float(lnw.max()), "bias", float(lnb.min()), float(lnb.max()))
