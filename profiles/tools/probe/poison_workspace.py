"""Does any launch read workspace it has not written?  Fill the caching allocator's free blocks with NaN (allocate, fill, free), build the engine
(its plan buffers are torch.empty: they now hold NaN), run three train steps and compare the losses with a run on zero-filled memory."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
from oracle import amid_oracle as orc
import test_gpu_fused_opt as t

CASES = [("plain", 64, 50, 64, True), ("plain", 128, 20, 96, True), ("plain", 128, 50, 40, False), ("plain", 128, 50, 256, True), ("bert", 128, 50, 96, True),
         ("itc", 128, 50, 48, True), ("inc", 64, 20, 32, False), ("dr", 128, 20, 64, False), ("bf16", 128, 50, 96, True), ("bert_itc", 128, 20, 32, False)]


def poison(value):
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    blocks = [torch.full((64 << 20,), value, device="cuda") for _ in range(12)]      # 3 GB of the value
    small = [torch.full((n,), value, device="cuda") for n in (256, 4096, 65536, 1 << 20) for _ in range(64)]
    torch.cuda.synchronize()
    del blocks, small


def run(kind, D, T, B, pool, value):
    n_items, hid = 900, 32
    P = orc.random_params(t.shapes(kind, n_items, D, T, hid, B), seed=5)
    batches = [orc.synthetic_batch(B, T, n_items - 1, pad_id=n_items - 1 if not kind.startswith("bert") else 0, neg=1, seed=60 + i) for i in range(3)]
    poison(value)
    eng = t.build(kind, P, n_items, D, T, hid, B)
    pl = eng.plan(B, T, 2, need_grad=True)
    cus = [{k: v.cuda() for k, v in b.items()} for b in batches]
    if pool:
        eng.set_input_pool(pl, torch.stack([eng.pack_batch(pl, c["i_node"], c["neg_samples"], c["seq_d1"], c["seq_d2"], c["label"], c["domain_id"]) for c in cus]))
    out = []
    for step in range(3):
        if not pool:
            c = cus[step % 3]
            eng.load_batch(pl, c["i_node"], c["neg_samples"], c["seq_d1"], c["seq_d2"], c["label"], c["domain_id"])
        eng.enqueue_train_step(pl)
        eng.sync()
        out.append(float((pl.dr_losses[0] if kind == "dr" else pl.loss).item()))
    sd = {k: v.clone() for k, v in eng.state_dict().items()}
    del eng, pl
    return out, sd


for case in CASES:
    a, sa = run(*case, 0.0)
    b, sb = run(*case, float("nan"))
    c, sc = run(*case, 1e30)
    bad = [k for k in sa if not (torch.equal(sa[k], sb[k]) and torch.equal(sa[k], sc[k]))]
    print(case, "ok" if (a == b == c and not bad) else f"DEPENDS ON UNWRITTEN MEMORY: zero {a} nan {b} big {c} params {bad[:3]}")
