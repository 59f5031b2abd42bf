"""The doubly-robust trainer with the reference's ``train_sr_dr.py`` command line (what ``run.sh`` launches), on the MI355X engine.

Same flags as ``train_sr.py`` plus ``--lr2``, ``--isDR``, ``--dr_e_w`` (train_sr_dr.py:543-575).  An epoch is two loops
(train_sr_dr.py:186-224 and :362-402):

  1. over ``<domain>_train<ratio>.csv``: loss_cls + dr_e_w * loss_dr_e on the three heads, Adam(lr)            (:216-224)
  2. over ``<domain>_train<ratio>_DR.csv`` (with ``ob_label``): loss_dr_r, a SECOND Adam(lr * lr2) over the same
     parameters with its own moments                                                                        (:392-398, :668-669)

with the evaluation of ``train_sr.py`` in between.  Both loops are ``model.train_step(...)`` (one hipGraph replay per step once
captured; the graph is keyed by the Adam state and the objective); ``engine.select_optimizer`` switches Adam states, applying the
pending lazy row updates of the state being left first.

    python train_sr_dr.py --data_root /path/to/AMID -ds mybank -dm loan_account --overlap_ratio 0.25 --model sasrec \
        --overlap True --isItC True --ts2 0.4 --neg_nums 999 --lr2 0.01 --dr_e_w 0.01            (run.sh:1)
"""
from __future__ import annotations

import logging
import os
import random
import time

import numpy as np
import torch

from . import train_sr as base
from .dataset_seq import DeviceBatches, DualDomainSeqDataset
from .model_seq import BERT4Rec, SASRec
from .utils import AverageMeter, init_logger

logger = logging.getLogger()


def build_parser():
    p = base.build_parser()
    p.add_argument("--lr2", type=float, default=0.01, help="the second optimizer runs at lr * lr2 (train_sr_dr.py:547, :669)")
    p.add_argument("--isDR", type=bool, default=True)
    p.add_argument("--dr_e_w", type=float, default=0.1, help="weight of loss_dr_e in the first objective (train_sr_dr.py:575, :221)")
    return p


def train(model, train_batches, train_batches_dr, args, val_batches, exchange=None):
    best = {}
    eng = model.engine
    world = exchange.world if exchange is not None else 1
    for epoch in range(args.epoch):
        stats = AverageMeter("loss_cls", "loss_dr_e", "loss_dr_r")
        model.train()
        t0, n_samples = time.perf_counter(), 0
        eng.select_optimizer(0, lr=args.lr)                                              # optimizer   (train_sr_dr.py:668)
        pooled = not args.no_pool        # each loop's epoch resident in HBM, one graph replay per step (see train_sr.py of this repo)
        t_pool = time.perf_counter()
        steps = ((None, None) for _ in range(model.begin_epoch_pool(train_batches.epoch_tensors(), exchange=exchange, dr_objective=0))) if pooled \
            else enumerate(train_batches)
        t_pool = time.perf_counter() - t_pool
        for i, (_, b) in enumerate(steps):
            if pooled:
                losses = model.pool_step(use_graph=not args.no_graph, exchange=exchange, dr_objective=0)
            else:
                losses = model.train_step(b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"], b["label"], b["domain_id"],
                                          use_graph=not args.no_graph, exchange=exchange, dr_objective=0)
            n_samples += args.bs * world
            if i % 20 == 0:                                                               # train_sr_dr.py:226-227
                eng.sync()
                lc, le, _ = losses.tolist()
                model.check_indices()
                stats.update(loss_cls=lc, loss_dr_e=le)
                logger.info(f"train cls loss:{stats.loss_cls}, dr_e loss:{stats.loss_dr_e} \t")
            if args.max_steps and i + 1 >= args.max_steps:
                break
        if pooled:
            model.end_epoch_pool()
        torch.cuda.synchronize()
        t_eval = time.perf_counter()
        res = base.test(model, args, val_batches)
        torch.cuda.synchronize()
        t_eval = time.perf_counter() - t_eval
        model.train()
        t_sw = time.perf_counter()
        eng.select_optimizer(1, lr=args.lr * args.lr2)                                   # optimizer2  (train_sr_dr.py:669)
        t_sw = time.perf_counter() - t_sw
        t0p = time.perf_counter()
        steps = ((None, None) for _ in range(model.begin_epoch_pool(train_batches_dr.epoch_tensors(), exchange=exchange, dr_objective=1))) if pooled \
            else enumerate(train_batches_dr)
        t_pool += time.perf_counter() - t0p
        for i, (_, b) in enumerate(steps):
            if pooled:
                losses = model.pool_step(use_graph=not args.no_graph, exchange=exchange, dr_objective=1)
            else:
                losses = model.train_step(b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"], b["label"], b["domain_id"],
                                          use_graph=not args.no_graph, exchange=exchange, ob_label=b["ob_label"], dr_objective=1)
            n_samples += args.bs * world
            if i % 20 == 0:                                                               # train_sr_dr.py:400-402
                eng.sync()
                stats.update(loss_dr_r=losses.tolist()[2])
                model.check_indices()
                logger.info(f"train loss_dr_r:{stats.loss_dr_r} \t")
            if args.max_steps and i + 1 >= args.max_steps:
                break
        if pooled:
            model.end_epoch_pool()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        logger.info(f"epoch {epoch}: {n_samples} samples in {dt:.2f} s = {n_samples / dt:.0f} samples/s (loaders and evaluation included; "
                    f"epoch pools {t_pool:.3f} s, evaluation {t_eval:.3f} s, optimizer switch {t_sw:.3f} s)")
        names = ("HR@1", "NDCG@1", "HR@5", "NDCG@5", "HR@10", "NDCG@10", "MRR")
        msg = [f"Epoch: {epoch}/{args.epoch} \tTrain cls Loss: {stats.loss_cls:.4f} \tTrain dr_e Loss: {stats.loss_dr_e:.4f} \t"
               f"Train dr_r Loss: {stats.loss_dr_r:.4f} \tVal loss: {res['loss']:.4f}"]
        for key, sc in res.items():
            if key == "loss":
                continue
            for n, v in zip(names, sc):
                best[(key, n)] = max(best.get((key, n), 0.0), v)
            msg.append(f"val {key} cur/max " + ", ".join(f"{n}: {v:.4f}/{best[(key, n)]:.4f}" for n, v in zip(names, sc)))
        logger.info("\n".join(msg))
    return best


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not args.isDR:
        raise SystemExit("train_sr_dr.py trains the doubly-robust heads (the reference's loop unpacks six outputs, train_sr_dr.py:204); "
                         "use train_sr.py without them")
    if args.model.lower() not in ("sasrec", "bert4rec"):
        raise SystemExit("the doubly-robust heads are built for --model sasrec (what run.sh trains) and bert4rec")
    # one process per GPU under `python -m torch.distributed.run --nproc-per-node N train_sr_dr.py ...` (the reference is single-GPU):
    # --bs is the batch PER GPU, both loops shard their batches by rank and exchange gradients every step (amid_amd/dist.py); with
    # --isItC (run.sh) InterComp's Linear(bs, 1) spans the GLOBAL batch of world x --bs rows (engine._enqueue_user_vectors)
    rank, world = base.init_data_parallel(args)
    if world > 1 and args.model.lower() != "sasrec":
        raise SystemExit("data-parallel train_sr_dr.py: --model sasrec")
    gbs = args.bs * (world if (args.isItC or args.isInC) else 1)
    summary = []
    for i in range(args.seeds):
        torch.manual_seed(i); np.random.seed(i); random.seed(i)                           # train_sr_dr.py:624-627
        args.log_file = "log" + str(i) + ".txt"
        user_length, item_length = 895510, 447410                                         # train_sr_dr.py:631-634
        root = os.path.join(args.data_root, f"{args.dataset_type}_dataset")
        stem = os.path.join(root, f"{args.domain_type}_train{int(args.overlap_ratio * 100)}")
        mk = lambda path, is_train, seed: DualDomainSeqDataset(seq_len=args.seq_len, isTrain=is_train, neg_nums=args.neg_nums,   # noqa: E731
                                                               long_length=args.long_length, pad_id=item_length + 1, seed=seed, csv_path=path)
        ds_train, ds_dr = mk(stem + ".csv", True, i), mk(stem + "_DR.csv", True, 500 + i)            # :635-640
        ds_val = mk(os.path.join(root, f"{args.domain_type}_test.csv"), False, 1000 + i)
        train_batches = DeviceBatches(ds_train, args.bs, shuffle=True, device=args.device, seed=i, rank=rank, world=world)
        train_batches_dr = DeviceBatches(ds_dr, args.bs, shuffle=True, device=args.device, seed=500 + i, rank=rank, world=world)
        val_batches = DeviceBatches(ds_val, gbs, shuffle=False, device=args.device, seed=i)
        torch.cuda.set_device(torch.device(args.device))
        model = (SASRec if args.model.lower() == "sasrec" else BERT4Rec)(user_length=2 * user_length, user_emb_dim=args.emb_dim, item_length=2 * item_length, item_emb_dim=args.emb_dim,
                       seq_len=args.seq_len, hid_dim=args.hid_dim, bs=gbs, isInC=args.isInC, isItC=args.isItC, threshold1=args.ts1,
                       threshold2=args.ts2, isDR=True, lr=args.lr, seed=i, **({"compute": "bf16"} if args.dtype == "bf16" else {}))
        model.engine.dr_e_w = float(args.dr_e_w)
        exchange = None
        if world > 1:      # identical replicas (weights from seed i on every rank), each rank's own dropout stream (as train_sr.py)
            model.engine.set_step(model.engine.step, seed=model.engine.rank_seed(i, rank))
            from .dist import SparseDenseExchange
            n_idx = args.bs * (2 * args.seq_len + 2)
            exchange = SparseDenseExchange(model.engine.merge_backend(world * n_idx),
                                           host_staging=os.environ.get("AMID_DIST_BACKEND", "nccl") != "nccl")
        init_logger(args.model_dir if rank == 0 else os.path.join(args.model_dir, f"rank{rank}"), args.log_file)
        logger.info(vars(args))
        summary.append(train(model, train_batches, train_batches_dr, args, val_batches, exchange))
    keys = sorted(summary[0]) if summary else []
    init_logger(args.model_dir if rank == 0 else os.path.join(args.model_dir, f"rank{rank}"), "log_all.txt")
    for k in keys:
        v = np.array([s[k] for s in summary])
        logger.info(f"{k[0]} {k[1]}: mean {v.mean():.4f} std {v.std():.4f}")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return summary


if __name__ == "__main__":
    main()
