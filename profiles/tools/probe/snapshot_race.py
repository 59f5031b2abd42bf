import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from test_gpu_timed_path import *
Bn, Tn, split, n_items = int(sys.argv[1]), int(sys.argv[2]), "mixed", int(sys.argv[3])
D, hid = 128, 32
P = orc.random_params(orc.sasrec_param_shapes(n_items, D, Tn, hid), seed=11 + Bn)
batches = [split_batch(Bn, Tn, n_items, seed=900 + t, split=split) for t in range(3)]
ref = None
for it in range(5):
    eng = make_engine(P, Tn, lr=1e-3, seed=77)
    eng.FUSED_TAIL = False
    pl = eng.plan(Bn, Tn, 2, need_grad=True)
    packed = []
    for b in batches:
        cu = {k: v.cuda() for k, v in b.items()}
        packed.append(eng.pack_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"]))
    eng.set_input_pool(pl, torch.stack(packed))
    eng.capture_train_step(pl)
    eng.replay_train_step(pl); eng.sync()
    dom = batches[0]["domain_id"].cuda()
    M = Bn * Tn
    live_rows = torch.cat([(dom == 0).repeat_interleave(Tn), (dom == 1).repeat_interleave(Tn)])     # rows of [2M] that are live
    cur = dict(xg=pl.xg[:2 * M][live_rows].clone(), tmq=pl.tmq[live_rows].clone(), w16=eng.w16.clone().view(torch.int16), x1=pl.x[1][live_rows].clone(), x2=pl.x[2][live_rows].clone(),
               u=pl.u.clone(), idx=pl.idx_all.clone(), live=pl.live.clone(), items=pl.xg[2 * M:].clone())
    print("run", it, "loss %.7f" % float(pl.loss.item()), end=" ")
    if ref is None:
        ref = cur
        print()
    else:
        for k in cur:
            nd = int((cur[k] != ref[k]).sum())
            print(k, nd, end="  ")
        if int((cur["x2"] != ref["x2"]).sum()):
            bad = (cur["x2"] != ref["x2"]).any(1).nonzero().flatten()
            print("\n   x2 bad rows", bad.numel(), "of", cur["x2"].shape[0], "first", bad[:10].tolist(), "seqs", torch.unique(bad // Tn)[:20].tolist(), end="")
        print(flush=True)
