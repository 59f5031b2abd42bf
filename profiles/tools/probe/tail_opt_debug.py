"""FUSED_OPT on / off in lockstep: after every step compare table / moments / dense parameters bit for bit and name the first rows that differ."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
from oracle import amid_oracle as orc
import test_gpu_timed_path as tp
Bn, T, D, hid, n_items = 256, 50, 128, 32, 3000
P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=15 + Bn)
batches = [tp.split_batch(Bn, T, n_items, seed=700 + t, split="mixed") for t in range(3)]
engs = []
for on in (False, True):
    eng = tp.make_engine(P, T, lr=1e-3, seed=79)
    eng.FUSED_OPT = on
    pl = eng.plan(Bn, T, 2, need_grad=True)
    packed = [eng.pack_batch(pl, *[b[k].cuda() for k in ("i_node", "neg_samples", "seq_d1", "seq_d2", "label", "domain_id")]) for b in batches]
    eng.set_input_pool(pl, torch.stack(packed))
    engs.append((eng, pl))
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    for eng, pl in engs:
        eng.enqueue_train_step(pl)
        eng.sync()
    (e0, p0), (e1, p1) = engs
    bad = []
    for name, a, b in (("table", e0.table, e1.table), ("m", e0.table_m, e1.table_m), ("v", e0.table_v, e1.table_v), ("last", e0.table_last, e1.table_last)):
        if not torch.equal(a, b):
            rows = (a != b).reshape(a.shape[0], -1).any(1).nonzero().flatten().tolist() if a.dim() > 1 else (a != b).nonzero().flatten().tolist()
            bad.append((name, rows[:10], len(rows)))
    for name in e0.dense.slots:
        for buf in ("data", "grad", "m", "v"):
            a, b = e0.dense.view(name, getattr(e0.dense, buf)), e1.dense.view(name, getattr(e1.dense, buf))
            if not torch.equal(a, b):
                bad.append((name, buf, int((a != b).sum())))
    U = int(p0.n_uniq.item())
    if not torch.equal(p0.uniq_grad[:U], p1.uniq_grad[:U]):
        rows = (p0.uniq_grad[:U] != p1.uniq_grad[:U]).any(1).nonzero().flatten()
        ids = p0.uniq_ids[:U][rows].tolist()
        so = p0.seg_off[:U + 1]
        bad.append(("uniq_grad rows", rows.tolist()[:10], ids[:10], [(int(so[r]), int(so[r + 1])) for r in rows.tolist()[:10]]))
    print("step", e0.step, "loss", float(p0.loss.item()), float(p1.loss.item()), "DIFF" if bad else "same", bad[:8])
    if bad:
        break
