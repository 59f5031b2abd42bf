"""Training driver with the reference's ``train_sr.py`` command line (train_sr.py:357-625), running the hot
loop on the MI355X engine.

Same flags as the reference (train_sr.py:360-389; the ones the reference never reads are accepted and
ignored; ``type=bool`` flags keep the reference's "any non-empty string is True" behaviour), same
constants (item_length 447 410, pad id item_length + 1, table of 2 x item_length rows, drop_last on both
loaders, five seeds 0..4, loss logged every 20 iterations, best-so-far HR/NDCG/MRR per epoch).
Additions: ``--data_root`` (the reference hard-codes /ossfs/workspace/CDSR), ``--seeds``, ``--device``,
``--no_graph``, ``--no_pool``, ``--max_steps``, ``--dtype {fp32,bf16}``; data parallel when launched by ``python -m torch.distributed.run --nproc-per-node N``
(one process per GPU, RCCL, ``--bs`` per GPU; BASELINE.json configs[3]).

    python train_sr.py --data_root /path/to/AMID -ds amazon -dm cloth_sport --overlap_ratio 0.75 \
        --model sasrec --bs 256 --seq_len 50 --emb_dim 128 --epoch 2 --seeds 1
"""
from __future__ import annotations

import argparse
import logging
import os
import random
import time

import numpy as np
import torch

from .dataset_seq import DeviceBatches, DualDomainSeqDataset, JointBatches
from .model_seq import BERT4Rec, GRU4Rec, SASRec
from .utils import AverageMeter, device_positive_ranks, init_logger, scores_from_ranks

logger = logging.getLogger()
FIX_VALUE = 1e-7          # train_sr.py:42: ties between the positive and a negative count against the positive


STEPS_PER_GRAPH = 4          # train(): consecutive pooled steps per replayed hipGraph (single GPU)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Multi-edge multi-domain training")
    p.add_argument("--epoch", type=int, default=50, help="# of epoch")
    p.add_argument("--bs", type=int, default=256, help="# images in batch")
    p.add_argument("--use_gpu", type=bool, default=True)
    p.add_argument("--lr", type=float, default=5e-4, help="initial learning rate for adam")
    p.add_argument("--emb_dim", type=int, default=128, help="embedding size")
    p.add_argument("--hid_dim", type=int, default=32, help="hidden layer dim")
    p.add_argument("--seq_len", type=int, default=20, help="the length of the sequence")
    p.add_argument("--graph_nums", type=int, default=2)
    p.add_argument("--head_nums", type=int, default=32)
    p.add_argument("--long_length", type=int, default=7, help="the length for setting long-tail node")
    p.add_argument("--m1_layers", type=int, default=3)
    p.add_argument("--m2_layers", type=int, default=3)
    p.add_argument("--m3_layers", type=int, default=4)
    p.add_argument("--m4_layers", type=int, default=2)
    p.add_argument("--alpha_l", type=int, default=3)
    p.add_argument("--neg_nums", type=int, default=199, help="sample negative numbers")
    p.add_argument("--mask_rate_enc", type=float, default=0.9)
    p.add_argument("--mask_rate_dec", type=float, default=0.9)
    p.add_argument("--overlap_ratio", type=float, default=0.5, help="overlap ratio for choose dataset")
    p.add_argument("--bs_ratio", type=float, default=0.5)
    p.add_argument("-md", "--model-dir", type=str, default="model/")
    p.add_argument("--log-file", type=str, default="log")
    p.add_argument("--model", type=str, default="model select")
    p.add_argument("-ds", "--dataset_type", type=str, default="amazon")
    p.add_argument("-dm", "--domain_type", type=str, default="movie_book")
    p.add_argument("--isInC", type=bool, default=False, help="add inc")
    p.add_argument("--isItC", type=bool, default=False, help="add itc")
    p.add_argument("--ts1", type=float, default=0.5)
    p.add_argument("--ts2", type=float, default=0.5)
    p.add_argument("--overlap", type=bool, default=False, help="split the metrics by overlapped / non-overlapped users")
    # additions
    p.add_argument("--data_root", type=str, default=".", help="directory holding {amazon,mybank}_dataset/ (reference: /ossfs/workspace/CDSR)")
    p.add_argument("--seeds", type=int, default=5, help="number of seeds 0..n-1 (reference: 5)")
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--no_graph", action="store_true", help="launch the step eagerly instead of replaying a hipGraph")
    p.add_argument("--no_pool", action="store_true", help="build and copy every batch inside the loop (the reference's way) instead of "
                                                          "keeping the epoch's packed batches resident in HBM")
    p.add_argument("--max_steps", type=int, default=0, help="stop each epoch after this many steps (0 = full epoch)")
    p.add_argument("--dtype", type=str, default="fp32", choices=("fp32", "bf16"),
                   help="fp32: exact fp32 matrix products; bf16: bf16 MFMA operands, fp32 accumulation and storage (sasrec, emb_dim 128)")
    return p


@torch.no_grad()
def test(model, args, val_batches):
    """train_sr.py:31-128: forward with neg_nums negatives, masked BCE, HR/NDCG/MRR of the positive's rank.  The rank of the
    positive is computed on the device per batch (amid_positive_rank_f32); only B ints per batch ever reach the host."""
    model.eval()
    stats = AverageMeter("loss", "loss_cls")
    ranks, ranks_raw, doms, ovs, losses = [], [], [], [], []
    # the plain SASRec model: the whole evaluation set resident in HBM, per batch three launches replayed as one graph -- the own domain's
    # sequence only (the other domain's logits are never read: utils.py:21-40, train_sr.py:63-64), candidates gathered inside the scorer,
    # BCE and both ranks in the same launch (SASRec.eval_ranks); every other model goes through model.forward below
    fused = None
    if hasattr(model, "eval_ranks") and hasattr(val_batches, "epoch_tensors") and len(val_batches) > 0:
        ep = val_batches.epoch_tensors()
        fused = model.eval_ranks(ep, FIX_VALUE)
        if fused is not None:
            losses = list(fused["loss"])
            ranks, ranks_raw = [fused["rank"].reshape(-1)], [fused["rank_raw"].reshape(-1)]
            doms, ovs = [ep["domain_id"].reshape(-1)], [ep["overlap_label"].reshape(-1)]
        else:                                  # (the draw of this epoch's negatives is made: iterate the same batches)
            val_batches = [{k: (v if k == "label" else v[i]) for k, v in ep.items()} for i in range(ep["seq_d1"].shape[0])]
    for b in ([] if fused is not None else val_batches):
        outs = model(b["user_node"], b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"], b["long_tail_mask_d1"],
                     b["long_tail_mask_d2"], False)
        p1, p2 = outs[0], outs[1]                                                         # isDR models return six outputs
        p1, p2 = p1.reshape(len(b["i_node"]), -1), p2.reshape(len(b["i_node"]), -1)
        y = b["label"]
        m2 = b["domain_id"].float().unsqueeze(1)
        bce = torch.nn.functional.binary_cross_entropy
        losses.append((bce(p1, y, reduction="none") * (1 - m2) + bce(p2, y, reduction="none") * m2).mean())   # train_sr.py:63-64
        ranks.append(device_positive_ranks(p1, p2, b["domain_id"], FIX_VALUE))            # train_sr.py:114-115
        if args.overlap:
            ranks_raw.append(device_positive_ranks(p1, p2, b["domain_id"], 0.0))          # the overlap splits skip fix_value
            ovs.append(b["overlap_label"])
        doms.append(b["domain_id"])
    for v in torch.stack(losses).tolist():                                                # one host transfer for the epoch
        stats.update(loss=v, loss_cls=v)
    rank, dom = torch.cat(ranks), torch.cat(doms)
    out = {"loss": stats.loss, "d1": scores_from_ranks(rank[dom == 0]), "d2": scores_from_ranks(rank[dom == 1])}
    if args.overlap:
        raw, ov = torch.cat(ranks_raw), torch.cat(ovs)
        out.update(d1_ov=scores_from_ranks(raw[(dom == 0) & (ov == 1)]), d1_no=scores_from_ranks(raw[(dom == 0) & (ov == 0)]),
                   d2_ov=scores_from_ranks(raw[(dom == 1) & (ov == 1)]), d2_no=scores_from_ranks(raw[(dom == 1) & (ov == 0)]))
    return out


def train(model, train_batches, args, val_batches, exchange=None):
    """train_sr.py:130-355 with the loop body (:190-217) fused into model.train_step."""
    best = {}
    for epoch in range(args.epoch):
        stats = AverageMeter("loss", "loss_cls")
        model.train()
        t0, n_samples = time.perf_counter(), 0
        pooled = not args.no_pool and hasattr(model, "begin_epoch_pool")
        # the epoch's batches resident in HBM, one graph replay per STEPS_PER_GRAPH steps (a replayed graph costs ~8 us of idle device
        # time between launches), no per-step tensor work on the host; the loss is looked at every 20 iterations as in the reference
        chunk = STEPS_PER_GRAPH if (pooled and not args.no_graph and exchange is None and not args.max_steps) else 1
        if pooled:
            n_batches = model.begin_epoch_pool(train_batches.epoch_tensors(), exchange=exchange)
            steps = ((None, None) for _ in range(n_batches))
        else:
            steps = enumerate(train_batches)
        pending = 0
        for i, (_, b) in enumerate(steps):
            if pooled:
                pending += 1
                last = i + 1 == n_batches
                if chunk > 1 and pending < chunk and not last:
                    continue                                   # the graph of `chunk` steps is replayed when its last step comes up
                if pending == chunk and chunk > 1:
                    loss = model.pool_step(use_graph=True, exchange=exchange, n_steps=chunk)
                else:
                    for _ in range(pending):                   # the epoch's tail (or chunk == 1): single-step graphs
                        loss = model.pool_step(use_graph=not args.no_graph, exchange=exchange)
                n_done, pending = pending, 0
            else:
                loss = model.train_step(b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"], b["label"], b["domain_id"],
                                        use_graph=not args.no_graph, exchange=exchange)
                n_done = 1
            n_samples += n_done * args.bs * (exchange.world if exchange is not None else 1)
            # the reference looks at the loss of iterations 0, 20, 40, ... (train_sr.py:217-219): logged here when the replayed chunk
            # [i + 1 - n_done, i] holds such an iteration, with the loss of the chunk's LAST step (the steps of a graph leave one loss
            # behind) -- the only host sync of the loop
            if (i // 20) * 20 > i - n_done:
                if pooled:
                    model.engine.sync()
                stats.update(loss=loss.item(), loss_cls=loss.item())
                if hasattr(model, "check_indices"):
                    model.check_indices()                                                 # nn.Embedding would have raised (model_seq.py:27-29)
                logger.info(f"train total loss:{stats.loss}, cls loss:{stats.loss_cls} \t")
            if args.max_steps and i + 1 >= args.max_steps:
                break
        if pooled:
            model.end_epoch_pool()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        logger.info(f"epoch {epoch}: {n_samples} samples in {dt:.4f} s = {n_samples / dt:.0f} samples/s (loader included)")
        t1 = time.perf_counter()
        res = test(model, args, val_batches)
        torch.cuda.synchronize()
        n_eval = len(val_batches) * args.bs
        logger.info(f"epoch {epoch}: evaluated {n_eval} samples x {args.neg_nums + 1} candidates in {time.perf_counter() - t1:.4f} s "
                    f"= {n_eval / (time.perf_counter() - t1):.0f} samples/s")
        names = ("HR@1", "NDCG@1", "HR@5", "NDCG@5", "HR@10", "NDCG@10", "MRR")
        msg = [f"Epoch: {epoch}/{args.epoch} \tTrain Loss: {stats.loss:.4f} \tVal loss: {res['loss']:.4f}"]
        for key, sc in res.items():
            if key == "loss":
                continue
            for n, v in zip(names, sc):
                best[(key, n)] = max(best.get((key, n), 0.0), v)
            msg.append(f"val {key} cur/max " + ", ".join(f"{n}: {v:.4f}/{best[(key, n)]:.4f}" for n, v in zip(names, sc)))
        logger.info("\n".join(msg))
    return best


def init_data_parallel(args):
    """One process per GPU under ``python -m torch.distributed.run --nproc-per-node N train_sr.py ...`` (not in the reference, which
    is single-GPU: ``DataParallel`` is commented out at train_sr.py:473).  ``--bs`` is the batch PER GPU.  Returns (rank, world)."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1
    import torch.distributed as dist
    backend = os.environ.get("AMID_DIST_BACKEND", "nccl")            # "gloo" only for the single-GPU two-process test
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "nccl":
        args.device = f"cuda:{local}"
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(args.device))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def main(argv=None):
    args = build_parser().parse_args(argv)
    rank, world = init_data_parallel(args)
    summary = []
    for i in range(args.seeds):
        torch.manual_seed(i); np.random.seed(i); random.seed(i)                           # train_sr.py:439-443
        args.log_file = "log" + str(i) + ".txt"
        user_length = 895510                                                              # train_sr.py:447
        item_length = 447410                                                              # train_sr.py:450
        root = os.path.join(args.data_root, f"{args.dataset_type}_dataset")
        # "-dm a+b": two datasets trained as ONE job on the shared table (BASELINE.json configs[3]): b's item
        # ids sit item_length + 2 rows behind a's (a's ids and the pad id are all <= item_length + 1), batches alternate a0 b0 a1 b1 ...
        parts = args.domain_type.split("+")
        if len(parts) > 2:
            raise SystemExit("-dm takes one dataset or two joined with '+'")
        trains, vals = [], []
        for j, dm in enumerate(parts):
            ds_train = DualDomainSeqDataset(seq_len=args.seq_len, isTrain=True, neg_nums=args.neg_nums, long_length=args.long_length,
                                            pad_id=item_length + 1, seed=i,
                                            csv_path=os.path.join(root, f"{dm}_train{int(args.overlap_ratio * 100)}.csv"))
            ds_val = DualDomainSeqDataset(seq_len=args.seq_len, isTrain=False, neg_nums=args.neg_nums, long_length=args.long_length,
                                          pad_id=item_length + 1, seed=1000 + i, csv_path=os.path.join(root, f"{dm}_test.csv"))
            if j:
                ds_train.shift_items(item_length + 2)
                ds_val.shift_items(item_length + 2)
            trains.append(DeviceBatches(ds_train, args.bs, shuffle=True, device=args.device, seed=i, rank=rank, world=world))
            # (isItC under data parallel: InterComp's Linear(bs, 1) is built for the GLOBAL batch of world x --bs rows, so the
            # evaluation, which every rank runs whole, steps through batches of that size)
            vals.append(DeviceBatches(ds_val, args.bs * (world if (args.isItC or args.isInC) else 1), shuffle=False, device=args.device, seed=i))
        if len(parts) == 2:
            # joint mode: the second dataset's ids sit item_length + 2 rows behind the first's in the reference-sized table of
            # 2 * item_length rows (train_sr.py:456) -- checked here, on the host, before any step runs (the fused step would only flag an
            # out-of-range id at its next log point)
            top = max(int(t.ds.max_item_id()) for t in (trains[1], vals[1]))
            if top >= 2 * item_length:
                raise SystemExit(f"-dm {args.domain_type}: item id {top - (item_length + 2)} of the second dataset lands on row {top} of a "
                                 f"{2 * item_length}-row table (ids up to {item_length - 3} fit)")
        train_batches = trains[0] if len(parts) == 1 else JointBatches(*trains)
        val_batches = vals[0] if len(parts) == 1 else JointBatches(*vals)
        item_length *= 2                                                                  # train_sr.py:456 ("for pad id")
        user_length *= 2
        cls = {"gru4rec": GRU4Rec, "sasrec": SASRec, "bert4rec": BERT4Rec}.get(args.model.lower())
        if cls is None:
            raise SystemExit(f"unknown --model {args.model!r} (gru4rec | sasrec | bert4rec)")
        torch.cuda.set_device(torch.device(args.device))
        model = cls(user_length=user_length, user_emb_dim=args.emb_dim, item_length=item_length, item_emb_dim=args.emb_dim,
                    seq_len=args.seq_len, hid_dim=args.hid_dim, bs=args.bs * (world if (args.isItC or args.isInC) else 1), isInC=args.isInC, isItC=args.isItC,
                    threshold1=args.ts1,
                    threshold2=args.ts2, lr=args.lr, seed=i, **({"compute": "bf16"} if args.dtype == "bf16" else {}))
        exchange = None
        if world > 1:
            # identical replicas (weights from seed i on every rank), but each rank's own dropout stream
            model.engine.set_step(model.engine.step, seed=model.engine.rank_seed(i, rank))
            from .dist import SparseDenseExchange
            n_idx = args.bs * (2 * args.seq_len + 2)                                      # per-rank index count of a train batch (1 negative)
            exchange = SparseDenseExchange(model.engine.merge_backend(world * n_idx),
                                           host_staging=os.environ.get("AMID_DIST_BACKEND", "nccl") != "nccl")
        init_logger(args.model_dir if rank == 0 else os.path.join(args.model_dir, f"rank{rank}"), args.log_file)
        logger.info(vars(args))
        best = train(model, train_batches, args, val_batches, exchange)
        summary.append(best)
    keys = sorted(summary[0]) if summary else []
    init_logger(args.model_dir if rank == 0 else os.path.join(args.model_dir, f"rank{rank}"), "log_all.txt")
    for k in keys:                                                                        # train_sr.py:549-569: mean / std over the seeds
        v = np.array([s[k] for s in summary])
        logger.info(f"{k[0]} {k[1]}: mean {v.mean():.4f} std {v.std():.4f}")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return summary


if __name__ == "__main__":
    main()
