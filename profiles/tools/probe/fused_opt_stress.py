"""Is the fused tail + optimizer deterministic?  The itc case of tests/test_gpu_fused_opt.py N times with FUSED_OPT off and on: every run's
parameters against the first off run's, bit for bit."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
from oracle import amid_oracle as orc
import test_gpu_fused_opt as t
kind = sys.argv[1] if len(sys.argv) > 1 else "itc"
D, T, B = {"itc": (128, 50, 48), "inc": (64, 20, 32), "bert_itc": (128, 20, 32)}[kind]
n_items, hid, K = 900, 32, 7
P = orc.random_params(t.shapes(kind, n_items, D, T, hid, B), seed=5)
batches = [orc.synthetic_batch(B, T, n_items - 1, pad_id=n_items - 1 if not kind.startswith("bert") else 0, neg=1, seed=60 + i) for i in range(3)]
ref = None
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    for on in (False, True):
        eng = t.build(kind, P, n_items, D, T, hid, B)
        eng.FUSED_OPT = on
        pl = eng.plan(B, T, 2, need_grad=True)
        cus = [{k: v.cuda() for k, v in b.items()} for b in batches]
        pooled = kind == "itc"
        if pooled:
            eng.set_input_pool(pl, torch.stack([eng.pack_batch(pl, c["i_node"], c["neg_samples"], c["seq_d1"], c["seq_d2"], c["label"], c["domain_id"]) for c in cus]))
        first_bad = None
        snaps = []
        losses = []
        for step in range(K):
            if not pooled:
                c = cus[step % 3]
                eng.load_batch(pl, c["i_node"], c["neg_samples"], c["seq_d1"], c["seq_d2"], c["label"], c["domain_id"])
            eng.enqueue_train_step(pl)
            eng.sync()
            losses.append(round(float(pl.loss.item()), 5))
            snaps.append({k: v.clone() for k, v in eng.state_dict().items()})
        if ref is None:
            ref = snaps
        for step in range(K):
            bad = [k for k in ref[step] if not torch.equal(ref[step][k], snaps[step][k])]
            if bad:
                first_bad = (step + 1, bad[:4])
                if "item_emb_layer.emb_item.weight" in bad:
                    d = (ref[step]["item_emb_layer.emb_item.weight"] != snaps[step]["item_emb_layer.emb_item.weight"]).any(1).nonzero().flatten().tolist()
                    first_bad += (d[:6],)
                break
        print(rep, "fused" if on else "two launches", "same" if first_bad is None else f"DIFF at step {first_bad}", losses[:4])
