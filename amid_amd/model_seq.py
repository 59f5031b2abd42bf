"""Drop-in ``nn.Module`` surface of the reference's ``model_seq.py`` for the MI355X hot path.

Same class names, constructor signatures, ``forward`` signatures and ``state_dict`` keys as the
reference; the arithmetic runs in the hand-written HIP kernels of
``libamid_hip.so`` (there is no PyTorch fallback: without the library, construction raises).

  SASRec(user_length, user_emb_dim, item_length, item_emb_dim, seq_len, hid_dim, bs,
         isInC, isItC, threshold1, threshold2, isDR=False)          model_seq.py:391
      forward(u_node, i_node, neg_samples, seq_d1, seq_d2, long_tail_mask_d1, long_tail_mask_d2,
              isTrain=True) -> (logits_d1, logits_d2)                 model_seq.py:416
  BERT4Rec(same arguments)                                            model_seq.py:250
      forward(same arguments) -> (logits_d1, logits_d2)               model_seq.py:277
  embItemLayerEnhance(item_length, emb_dim)                           model_seq.py:23
  predictModule(emb_dim, hid_dim)                                     model_seq.py:33
  Log2feats(user_length, user_emb_dim, item_length, item_emb_dim, seq_len, hid_dim)   model_seq.py:332

As in the reference, ``u_node``, both ``long_tail_mask_*``, ``isTrain``, ``user_length`` and ``bs``
are accepted and ignored; dropout follows ``module.training``.  Two ways to train:

  reference loop   ``opt = torch.optim.Adam(model.parameters(), lr)``; ``loss.backward(); opt.step()``
                   (train_sr.py:213-215) works unchanged: backward runs the HIP kernels and hands autograd
                   the gradients (the table gradient is materialised densely for torch's optimizer);
  fused step       ``model.train_step(batch)`` = forward + masked BCE + backward + dense-equivalent lazy
                   Adam as one hipGraph replay -- what ``train_sr.py`` of this repo and ``bench.py`` use.

Out of scope this round (constructors kept for import / state_dict parity, ``forward`` raises):
GRU4Rec (recurrent), embUserLayerEnhance (dead code in the reference).  BERT4Rec(isInC=True) / (isItC=True) -- the comp module in
FRONT of the encoders, model_seq.py:283-294 -- are built (the reference itself fails with both flags).
SASRec(isItC=True, isDR=True) -- InterComp after the encoders + the doubly-robust heads, what run.sh trains through
train_sr_dr.py -- IS built (csrc/intercomp.hip, amid_dr_loss_f32), and so is SASRec(isInC=True) -- InnerComp on the gathered rows
before the encoders, which then run over 2 * seq_len tokens (csrc/innercomp.hip).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from ._lib import lib, ptr_array
from .engine import SASREC_LN_EPS, SasrecEngine
from .engine_bert import Bert4recEngine


def _register_tree(root: nn.Module, dotted: str, param: nn.Parameter) -> None:
    """Register ``param`` under a dotted state_dict name, creating plain container modules on the way."""
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, nn.Module())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], param)


def _not_built(what: str, cite: str):
    raise NotImplementedError(
        f"{what} ({cite}) is outside the MI355X hot path built so far (SURVEY.md section 8(f)); the constructor exists for "
        f"signature / state_dict parity only.  There is deliberately no PyTorch fallback.")


# ------------------------------------------------------------------------------------------------
class _SasrecFunction(torch.autograd.Function):
    """Autograd bridge: forward / backward are the HIP launch sequences of SasrecEngine."""

    @staticmethod
    def forward(ctx, model, need_grad, i_node, neg_samples, seq_d1, seq_d2, *params):
        eng: SasrecEngine = model.engine
        B, T = seq_d1.shape
        neg = neg_samples.reshape(B, -1)
        pl = eng.plan(B, T, 1 + neg.shape[1], need_grad=bool(need_grad))
        eng.stream.wait_stream(torch.cuda.current_stream())
        eng.load_batch(pl, i_node, neg, seq_d1, seq_d2)
        if model.training:
            eng.enqueue_step_begin()                   # fresh dropout counter per training forward
        eng.enqueue_prepare(pl, sparse=need_grad)
        if need_grad and eng.table_m is not None:
            eng.enqueue_catchup(pl)                    # lazy-Adam rows must be current before they are gathered
        eng.enqueue_forward(pl, train=model.training, with_loss=False)
        torch.cuda.current_stream().wait_stream(eng.stream)
        ctx.model, ctx.pl, ctx.train = model, pl, model.training
        eng.check_index_error(pl)
        outs = (pl.p1, pl.p2) + ((pl.ips1, pl.ips2, pl.g1, pl.g2) if eng.dr else ())      # isDR: model_seq.py:436-440
        return tuple(o.clone() for o in outs)

    @staticmethod
    def backward(ctx, *gouts):
        model, pl = ctx.model, ctx.pl
        eng: SasrecEngine = model.engine
        eng.stream.wait_stream(torch.cuda.current_stream())
        dsts = (pl.dp1, pl.dp2) + ((pl.dips1, pl.dips2, pl.dg1, pl.dg2) if eng.dr else ())
        with torch.cuda.stream(eng.stream):
            for d, g in zip(dsts, gouts):
                if g is None:
                    d.zero_()
                else:
                    d.copy_(g.reshape(d.shape))
        eng.enqueue_backward(pl, train=ctx.train)
        grads = []
        with torch.cuda.stream(eng.stream):
            for name in model._param_names:
                if name == "item_emb_layer.emb_item.weight":
                    if model.fused_optimizer:
                        grads.append(None)             # rows stay in pl.uniq_ids / pl.uniq_grad for the lazy Adam
                    else:
                        U = int(pl.n_uniq.item())
                        dense = torch.zeros_like(eng.table)
                        dense.index_copy_(0, pl.uniq_ids[:U].long(), pl.uniq_grad[:U])
                        grads.append(dense)
                else:
                    grads.append(eng.dense.view(name, eng.dense.grad).clone())
        torch.cuda.current_stream().wait_stream(eng.stream)
        model._last_plan = pl
        return (None, None, None, None, None, None, *grads)


class SASRec(nn.Module):
    """model_seq.py:390-443 on the HIP engine (isInC / isItC either way: with either every batch must hold exactly `bs` rows,
    as in the reference where trans_bs is Linear(bs, 1) over the batch, and state_dict gains inc_d{1,2}.* / itc_d{1,2}.*; with
    isInC the encoders run over 2 * seq_len tokens and pos_emb has 2 * seq_len rows, :398-401; isDR either way:
    with it forward returns six outputs -- logits, ips, gfunc per domain, :436-440 -- and state_dict gains predict_ips.*,
    predict_gfunc.*)."""

    ENGINE_CLS = SasrecEngine
    COMP_IN_FRONT = False        # InterComp after the encoders (model_seq.py:426-431), the configuration run.sh trains

    def __init__(self, user_length, user_emb_dim, item_length, item_emb_dim, seq_len, hid_dim, bs, isInC, isItC, threshold1,
                 threshold2, isDR=False, device: Optional[str] = None, lr: float = 5e-4, seed: int = 0, compute: str = "f32"):
        """Beyond the reference's arguments: device, lr / seed (the fused train_step owns Adam and the dropout counter) and
        compute ("f32": exact fp32 matrix products, the default; "bf16": bf16 MFMA operands with fp32 accumulation, SASRec with
        emb_dim 128 only -- BASELINE.json configs[2])."""
        super().__init__()
        if isInC and isItC and self.COMP_IN_FRONT:
            raise ValueError("BERT4Rec(isInC=True, isItC=True): the reference itself fails on this combination (its key mask keeps 2T "
                             "keys for 4T tokens, model_seq.py:294); pick one")
        if user_emb_dim != item_emb_dim:
            raise ValueError("the reference feeds item rows into encoders built with user_emb_dim: the two must be equal")
        lib()                                                   # fail loudly without libamid_hip.so
        self.user_emb_dim = user_emb_dim
        self.isInC, self.isItC, self.isDR = isInC, isItC, isDR
        dev = device or ("cuda:%d" % torch.cuda.current_device())
        kw = {}
        if self.COMP_IN_FRONT:          # BERT4Rec: either module runs on the gathered rows, before the encoders (:283-294)
            if isInC or isItC:
                kw.update(comp="inc" if isInC else "itc", comp_bs=bs, comp_threshold=threshold1 if isInC else threshold2)
        else:
            if isItC:
                kw.update(itc_bs=bs, itc_threshold=threshold2)
            if isInC:
                kw.update(inc_bs=bs, inc_threshold=threshold1)
        if isDR:
            kw["dr"] = True
        if compute != "f32":
            kw["compute"] = compute
        self.engine = self.ENGINE_CLS(item_length, item_emb_dim, seq_len, hid_dim, device=dev, lr=lr, seed=seed, **kw)
        eng = self.engine
        self._param_names = ["item_emb_layer.emb_item.weight"] + list(eng.dense.slots)
        self._init_reference_defaults(seed)
        _register_tree(self, "item_emb_layer.emb_item.weight", nn.Parameter(eng.table))
        for name in eng.dense.slots:
            _register_tree(self, name, nn.Parameter(eng.dense.view(name)))
        self.fused_optimizer = False            # set by train_step(): table gradient stays sparse, Adam runs in HIP
        self._last_plan = None

    def _init_reference_defaults(self, seed: int) -> None:
        """Same initial distributions as the reference's modules (nn.Embedding N(0,1); nn.Linear / Conv1d /
        MultiheadAttention defaults; LayerNorm 1/0), drawn from a private generator."""
        eng = self.engine
        g = torch.Generator(device="cpu").manual_seed(seed)
        D, hid = eng.D, eng.hid
        with torch.no_grad():
            eng.table.copy_(torch.randn(eng.n_rows, D, generator=g))
            for name in eng.dense.slots:
                v = eng.dense.view(name)
                if name.endswith("pos_emb.weight"):
                    v.copy_(torch.randn(v.shape, generator=g))
                elif "layernorm" in name:
                    v.fill_(1.0 if name.endswith("weight") else 0.0)
                elif name.endswith("norm.a_2") or name.endswith("norm.b_2"):      # BERT4Rec LayerNorm, model_seq.py:119-120
                    v.fill_(1.0 if name.endswith("a_2") else 0.0)
                elif name.endswith("in_proj_weight"):          # xavier_uniform_ (nn.MultiheadAttention._reset_parameters)
                    a = (6.0 / (3 * D + D)) ** 0.5
                    v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) * a)
                elif name.endswith("in_proj_bias") or name.endswith("out_proj.bias"):
                    v.zero_()
                else:                                          # kaiming_uniform_(a=sqrt(5)) => U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                    fan_in = {"predictModule.fc.0.weight": 2 * D, "predictModule.fc.0.bias": 2 * D, "predictModule.fc.2.weight": hid,
                              "predictModule.fc.2.bias": hid}.get(name, 4 * D if ".feed_forward.w_2." in name else D)
                    if ".trans_bs." in name:                   # Inter/InnerComp's Linear(bs, 1) over the batch (model_seq.py:480, :457)
                        fan_in = eng.itc_bs or eng.inc_bs
                    elif name.startswith(("predict_ips.fc.0", "predict_gfunc.fc.0")):
                        fan_in = 2 * D
                    elif name.startswith(("predict_ips.fc.2", "predict_gfunc.fc.2")):
                        fan_in = hid
                    a = 1.0 / fan_in ** 0.5
                    v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) * a)
        torch.cuda.synchronize(eng.device)

    # -- reference forward ---------------------------------------------------------------------
    def forward(self, u_node, i_node, neg_samples, seq_d1, seq_d2, long_tail_mask_d1, long_tail_mask_d2, isTrain=True):
        if not self.training and self.engine.table_m is not None:
            self.engine.flush_table()               # rows with pending zero-gradient Adam steps must be current for eval
        params = [self.get_parameter(n) for n in self._param_names]
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        outs = _SasrecFunction.apply(self, need_grad, i_node, neg_samples, seq_d1, seq_d2, *params)
        return tuple(o.squeeze() for o in outs)        # model_seq.py:54 ; isDR: six outputs (:436-440)

    # -- fused fast path -------------------------------------------------------------------------
    def train_step(self, i_node, neg_samples, seq_d1, seq_d2, labels, domain_id, use_graph: bool = True, exchange=None,
                   ob_label=None, dr_objective: int = 0) -> torch.Tensor:
        """train_sr.py:190-217 as one launch sequence (one hipGraph replay once captured).  Returns the loss (device scalar;
        under data parallelism the mean over THIS rank's shard).  exchange: an amid_amd.dist.SparseDenseExchange for
        data-parallel training (one process per GPU): local gradients, one dense all-reduce + sparse all-gather, Adam.
        isDR models: dr_objective 0 = loss_cls + dr_e_w * loss_dr_e (train_sr_dr.py:216-224), 1 = loss_dr_r with ob_label
        (:392-398); returns the three-element tensor (loss_cls, loss_dr_e, loss_dr_r) of this batch.  Pick the Adam state with
        engine.select_optimizer() (the reference alternates two optimizers)."""
        eng = self.engine
        self.fused_optimizer = True
        B, T = seq_d1.shape
        neg = neg_samples.reshape(B, -1)
        pl = eng.plan(B, T, 1 + neg.shape[1], need_grad=True)
        eng.stream.wait_stream(torch.cuda.current_stream())
        if eng.dr:
            eng.dr_mode = int(dr_objective)
            if ob_label is None:
                if eng.dr_mode == 1:
                    raise ValueError("dr_objective 1 (loss_dr_r) needs ob_label (train_sr_dr.py:372)")
                ob_label = torch.zeros(B, dtype=torch.int64, device=seq_d1.device)
        eng.load_batch(pl, i_node, neg, seq_d1, seq_d2, labels, domain_id, ob_label if eng.dr else None)
        if exchange is not None and exchange.world > 1:
            # (isItC / isInC: collectives inside the step -- no local graph; train_step_dp replays graph segments once it knows a bound)
            if use_graph and not (eng.itc_bs or eng.inc_bs) and not eng.has_local_graph(pl):
                eng.capture_local_grads(pl)
            eng.train_step_dp(pl, exchange, use_graph=use_graph)
        elif use_graph:
            if not eng.has_graph(pl):
                eng.capture_train_step(pl)
            eng.replay_train_step(pl)
        else:
            eng.enqueue_train_step(pl)
        torch.cuda.current_stream().wait_stream(eng.stream)
        self._last_plan = pl
        return pl.dr_losses if eng.dr else pl.loss

    def begin_epoch_pool(self, ep: Dict[str, torch.Tensor], exchange=None, dr_objective: int = 0) -> int:
        """Make a whole epoch resident in HBM (ep = DeviceBatches.epoch_tensors()): the batches are packed into one
        [n_batches, words] tensor and every following pool_step() consumes the next one, picked on the device by the step counter
        -- no per-step input copies or host tensor work (train_sr.py:185-199 builds and moves each batch inside the loop).
        Returns the number of batches.  Under data parallelism the world's largest per-step unique-row count of the epoch is reduced
        here, once, so the sparse exchange of every step runs without a device -> host sync."""
        eng = self.engine
        if eng.dr:
            eng.dr_mode = int(dr_objective)          # isDR: a pool belongs to the (Adam state, objective) selected now
        nb, B, T = ep["seq_d1"].shape
        neg = ep["neg_samples"].reshape(nb, B, -1)
        pl = eng.plan(B, T, 1 + neg.shape[2], need_grad=True)
        eng.stream.wait_stream(torch.cuda.current_stream())
        pool = eng.pack_epoch(pl, ep["i_node"], neg, ep["seq_d1"], ep["seq_d2"], ep["label"], ep["domain_id"], ep.get("ob_label") if eng.dr else None)
        self._pool_umax = None
        if exchange is not None and exchange.world > 1:
            idx = torch.cat((ep["i_node"].reshape(nb, -1), neg.reshape(nb, -1), ep["seq_d1"].reshape(nb, -1), ep["seq_d2"].reshape(nb, -1)), 1)
            srt = torch.sort(idx, dim=1).values
            cnt = ((srt[:, 1:] != srt[:, :-1]).sum(1) + 1).max().reshape(1)
            exchange.all_reduce_max(cnt)
            self._pool_umax = (int(cnt.item()) + 255) // 256 * 256          # one bucketed bound per epoch: the exchange's graph pair is reused
        torch.cuda.current_stream().synchronize()
        if not eng.refill_input_pool(pl, pool):            # same shape, step on a pool boundary: the captured graphs stay valid
            eng.set_input_pool(pl, pool)
        self._pool_plan = pl
        self.fused_optimizer = True
        return nb

    def pool_step(self, use_graph: bool = True, exchange=None, dr_objective: int = 0, n_steps: int = 1) -> torch.Tensor:
        """One train step on the next batch of the pool installed by begin_epoch_pool(); returns what train_step() returns.
        n_steps > 1 (single GPU, graphs): that many consecutive steps as ONE replayed graph (the caller sees the last step's loss)."""
        eng, pl = self.engine, self._pool_plan
        if eng.dr:
            eng.dr_mode = int(dr_objective)
        if n_steps > 1:
            if not use_graph or (exchange is not None and exchange.world > 1):
                raise ValueError("several steps per call need graph replay on a single GPU")
            if (eng._graph_key(), n_steps) not in getattr(pl, "graphs_n", {}):
                eng.capture_train_steps(pl, n_steps)
            eng.replay_train_steps(pl, n_steps)
        elif exchange is not None and exchange.world > 1:
            # (isItC / isInC: collectives inside the step -- no local graph; with the epoch's bound train_step_dp replays graph SEGMENTS
            # cut at those collectives, engine._coll)
            if use_graph and not (eng.itc_bs or eng.inc_bs) and not eng.has_local_graph(pl):
                eng.capture_local_grads(pl)
            eng.train_step_dp(pl, exchange, use_graph=use_graph, umax=self._pool_umax)
        elif use_graph:
            if not eng.has_graph(pl):
                eng.capture_train_step(pl)
            eng.replay_train_step(pl)
        else:
            eng.enqueue_train_step(pl)
        self._last_plan = pl
        return pl.dr_losses if eng.dr else pl.loss

    def eval_ranks(self, ep: Dict[str, torch.Tensor], fix_value: float, use_graph: bool = True) -> Optional[Dict[str, torch.Tensor]]:
        """test()'s arithmetic (train_sr.py:31-128) over a whole evaluation set resident in HBM (ep = DeviceBatches.epoch_tensors()):
        per batch the eval-mode forward of every sample's OWN domain sequence, its 1 + neg_nums scores, the masked BCE mean (:63-64) and
        the positive's rank with and without fix_value (:114-115; utils.py:21-40, :296-297) -- three launches replayed as one graph
        (SasrecEngine.enqueue_eval).  Returns device tensors rank [n, B], rank_raw [n, B] (int32) and loss [n], or None when this model
        evaluates through forward() (isItC / isInC / isDR, BERT4Rec, shapes the one-launch forward does not cover)."""
        eng = self.engine
        nb, B, T = ep["seq_d1"].shape
        neg = ep["neg_samples"].reshape(nb, B, -1)
        pl = eng.plan(B, T, 1 + neg.shape[2], need_grad=False)
        if not eng.eval_fused_ok(pl):
            return None
        eng.stream.wait_stream(torch.cuda.current_stream())
        packed = eng.pack_epoch(pl, ep["i_node"], neg, ep["seq_d1"], ep["seq_d2"], ep["label"], ep["domain_id"])
        eng.stream.wait_stream(torch.cuda.current_stream())
        out = eng.eval_epoch(pl, packed, fix_value, with_loss=True, use_graph=use_graph)
        torch.cuda.current_stream().wait_stream(eng.stream)
        eng.check_index_error(pl)
        return {"rank": out[:, :B], "rank_raw": out[:, B:2 * B], "loss": out[:, 2 * B:].view(torch.float32).sum(1)}

    def check_indices(self) -> None:
        """Raise IndexError if any batch since the last check carried an item id outside the table (nn.Embedding raises on the spot,
        model_seq.py:27-29; the fused step flags it on the device and keeps going with row 0).  One device -> host read: call it
        where the loop synchronises anyway (the loss log every 20 iterations, the end of an epoch)."""
        pl = getattr(self, "_last_plan", None) or getattr(self, "_pool_plan", None)
        if pl is not None:
            self.engine.check_index_error(pl)

    def end_epoch_pool(self) -> None:
        """Hand the device back to torch's stream after an epoch of pool_step()s (the pool stays installed for the next epoch's
        refill; train_step() on the same plan is refused while it is -- use drop_epoch_pool())."""
        torch.cuda.current_stream().wait_stream(self.engine.stream)
        self.check_indices()

    def drop_epoch_pool(self) -> None:
        """Remove the pool of the current (Adam state, objective)."""
        eng = self.engine
        eng.sync()
        if getattr(self, "_pool_plan", None) is not None:
            eng.set_input_pool(self._pool_plan, None)

    def flush(self) -> None:
        """Bring lazily-updated table rows up to date (call before eval / state_dict() / checkpoints)."""
        self.engine.flush_table()
        self.engine.sync()

    def state_dict(self, *args, **kwargs):
        self.flush()
        return super().state_dict(*args, **kwargs)


# ------------------------------------------------------------------------------------------------
class _GatherFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, idx):
        L = lib()
        flat = idx.reshape(-1).contiguous()
        out = torch.empty(flat.numel(), weight.shape[1], dtype=weight.dtype, device=weight.device)
        err = torch.zeros(1, dtype=torch.int32, device=weight.device)
        is64 = 1 if flat.dtype == torch.int64 else 0
        if not is64 and flat.dtype != torch.int32:
            raise TypeError("item ids must be int64 or int32")
        L.call("amid_gather_rows_f32", weight.data_ptr(), weight.shape[0], weight.shape[1], flat.data_ptr(), is64, flat.numel(),
               out.data_ptr(), err.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if int(err.item()):
            raise IndexError("index out of range in embItemLayerEnhance")
        ctx.save_for_backward(flat)
        ctx.shape = weight.shape
        return out.reshape(*idx.shape, weight.shape[1])

    _sort_ws = {}          # (device, n) -> the zero-filled sort workspace of backward

    @staticmethod
    def backward(ctx, g):
        (flat,) = ctx.saved_tensors
        L, s = lib(), torch.cuda.current_stream().cuda_stream
        n, D = flat.numel(), ctx.shape[1]
        if D not in (64, 128, 256):
            raise NotImplementedError(f"embItemLayerEnhance backward: the segment-reduce kernel is built for emb_dim in (64, 128, 256), got {D}")
        dev = g.device
        idx32 = flat.to(torch.int32)
        rows = g.reshape(n, D).contiguous()
        # the sort's workspace starts with a fixed ~8.4 MB head that must be zero once and that every call leaves consistent: one
        # zero-filled workspace per (device, n) for the life of the process instead of an allocation + memset per backward
        ws = _GatherFunction._sort_ws.get((dev, n))
        if ws is None:
            ws = _GatherFunction._sort_ws[(dev, n)] = torch.zeros(L.value("amid_sort_unique_workspace_bytes", n), dtype=torch.uint8, device=dev)
        pos, uniq = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
        seg, nu = torch.empty(n + 1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        sof = torch.empty(n, dtype=torch.int32, device=dev)
        L.call("amid_sort_unique_i32", idx32.data_ptr(), n, ctx.shape[0], ws.data_ptr(), pos.data_ptr(), uniq.data_ptr(), seg.data_ptr(),
               sof.data_ptr(), nu.data_ptr(), s)
        ws2 = torch.empty(L.value("amid_segreduce_workspace_bytes", n, D), dtype=torch.uint8, device=dev)
        ug = torch.empty(n, D, dtype=torch.float32, device=dev)
        L.call("amid_embgrad_segreduce_f32", rows.data_ptr(), pos.data_ptr(), seg.data_ptr(), sof.data_ptr(), n, D, ws2.data_ptr(),
               ug.data_ptr(), s)
        U = int(nu.item())
        dense = torch.zeros(ctx.shape, dtype=torch.float32, device=dev)
        dense.index_copy_(0, uniq[:U].long(), ug[:U])
        return dense, None


class embItemLayerEnhance(nn.Module):
    """model_seq.py:22-29: plain lookup, no padding_idx (the pad id is an ordinary trainable row)."""

    def __init__(self, item_length, emb_dim):
        super().__init__()
        lib()
        self.emb_item = nn.Module()
        self.emb_item.register_parameter("weight", nn.Parameter(torch.randn(item_length, emb_dim, device="cuda")))

    def forward(self, item_id):
        return _GatherFunction.apply(self.emb_item.weight, item_id)


class predictModule(nn.Module):
    """model_seq.py:32-54 (forward only through the HIP scorer; training goes through SASRec)."""

    def __init__(self, emb_dim, hid_dim):
        super().__init__()
        lib()
        self.fc = nn.Sequential(nn.Linear(emb_dim * 2, hid_dim), nn.ReLU(), nn.Linear(hid_dim, 1)).cuda()
        self.emb_dim, self.hid_dim = emb_dim, hid_dim

    @torch.no_grad()
    def forward(self, user_spf1, user_spf2, i_feat):
        L = lib()
        B, NI, D = i_feat.shape
        u = torch.stack((user_spf1, user_spf2)).contiguous().float()
        items = i_feat.contiguous().float()
        p1 = torch.empty(B, NI, device=u.device)
        p2 = torch.empty(B, NI, device=u.device)
        L.call("amid_scorer_fwd_f32", u.data_ptr(), items.data_ptr(), self.fc[0].weight.data_ptr(), self.fc[0].bias.data_ptr(),
               self.fc[2].weight.data_ptr(), self.fc[2].bias.data_ptr(), None, None, B, NI, D, self.hid_dim, p1.data_ptr(), p2.data_ptr(),
               None, None, None, torch.cuda.current_stream().cuda_stream)
        return p1.squeeze(), p2.squeeze()


class Log2feats(nn.Module):
    """model_seq.py:331-387 as a standalone module (eval / inference): LN -> causal MHA -> FFN, two blocks."""

    def __init__(self, user_length, user_emb_dim, item_length, item_emb_dim, seq_len, hid_dim):
        super().__init__()
        lib()
        self.D, self.T = user_emb_dim, seq_len
        # a private two-domain engine whose table is the identity of the rows handed to forward()
        self._eng: Optional[SasrecEngine] = None
        names = [(n[len("sac1."):], s) for n, s in __import__("amid_amd.engine", fromlist=["x"]).sasrec_dense_names(seq_len, user_emb_dim, hid_dim)
                 if n.startswith("sac1.")]
        for n, shp in names:
            init = torch.ones(shp) if ("layernorm" in n and n.endswith("weight")) else torch.randn(shp) * 0.05
            if "layernorm" in n and n.endswith("bias"):
                init = torch.zeros(shp)
            _register_tree(self, n, nn.Parameter(init.cuda()))

    @torch.no_grad()
    def forward(self, log_seqs):
        B, T, D = log_seqs.shape
        eng = SasrecEngine(B * T + 2, D, self.T, 8, device=str(log_seqs.device))
        sd: Dict[str, torch.Tensor] = {"item_emb_layer.emb_item.weight": torch.cat((log_seqs.reshape(B * T, D).float(),
                                                                                     torch.zeros(2, D, device=log_seqs.device)))}
        own = dict(self.named_parameters())
        for name in eng.dense.slots:
            if name.startswith("sac"):
                sd[name] = own[name.split(".", 1)[1]]
            else:
                sd[name] = torch.zeros(eng.dense.slots[name][1], device=log_seqs.device)
        eng.load_state_dict(sd)
        pl = eng.plan(B, T, 2, need_grad=False)
        seq = torch.arange(B * T, device=log_seqs.device).reshape(B, T)
        z = torch.zeros(B, dtype=torch.long, device=log_seqs.device)
        eng.stream.wait_stream(torch.cuda.current_stream())
        eng.load_batch(pl, z, z.reshape(B, 1), seq, seq)
        eng.enqueue_prepare(pl, sparse=False)
        eng.enqueue_forward(pl, train=False, with_loss=False)
        # last LayerNorm (model_seq.py:385) of the first domain's rows, row by row on the host side of the ABI
        x = pl.x[2][: B * T]
        out = torch.empty_like(x)
        lib().call("amid_layernorm_rows_f32", x.data_ptr(), own["last_layernorm.weight"].data_ptr(), own["last_layernorm.bias"].data_ptr(),
                   B * T, D, SASREC_LN_EPS, out.data_ptr(), eng.s)          # model_seq.py:385
        torch.cuda.current_stream().wait_stream(eng.stream)
        return out.reshape(B, T, D)


# ------------------------------------------------------------------------------------------------
# signature-parity shells (constructors build the reference's parameters; forward is not built yet)
class embUserLayerEnhance(nn.Module):
    def __init__(self, user_length, emb_dim):          # model_seq.py:9-20 (never instantiated by the reference's models)
        super().__init__()
        self.emb_user_share = nn.Embedding(user_length, emb_dim)
        self.transd1 = nn.Linear(emb_dim, emb_dim)
        self.transd2 = nn.Linear(emb_dim, emb_dim)

    def forward(self, user_id):
        _not_built("embUserLayerEnhance", "model_seq.py:9-20")


def getBinaryTensor(imgTensor, boundary):              # model_seq.py:445-448
    return torch.where(imgTensor > boundary, torch.ones_like(imgTensor), torch.zeros_like(imgTensor))


class _CompFunction(torch.autograd.Function):
    """The standalone comp modules on the kernels BERT4Rec's front-of-encoder form uses (csrc/innercomp.hip): the rows are packed
    as the [2, B, T, D] pair those kernels take (slot 0 = the rows that keep their place, slot 1 = the rows mixed into the token
    group; InnerComp: the same rows twice) and slot 0's half of every result is what the module returns."""

    @staticmethod
    def forward(ctx, cross, threshold, seq_a, seq_b, w_nn, b_nn, w_bs, b_bs):
        L, st = lib(), torch.cuda.current_stream().cuda_stream
        B, T, D = seq_a.shape
        dev = seq_a.device
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)      # noqa: E731
        xg = torch.stack((seq_a.float(), seq_b.float())).contiguous()
        par = [t.detach().float().contiguous() for t in (w_nn, b_nn, w_bs, b_bs)]
        two = lambda t: ptr_array([t.data_ptr(), t.data_ptr()])                # noqa: E731
        s, gate, S, Z, sw, x0 = f(2, B), f(2, B), f(2, T, D), f(2, T, D), f(2), f(2, B, 2 * T, D)
        L.call("amid_bert_comp_score_f32", xg.data_ptr(), B, T, D, cross, s.data_ptr(), st)
        L.call("amid_bert_comp_fwd_f32", xg.data_ptr(), s.data_ptr(), two(par[0]), two(par[1]), two(par[2]), two(par[3]), float(threshold),
               cross, B, T, D, gate.data_ptr(), S.data_ptr(), Z.data_ptr(), sw.data_ptr(), x0.data_ptr(), st)
        ctx.save_for_backward(xg, gate, S, sw, *par[:3])
        ctx.cross = cross
        return x0[0]

    @staticmethod
    def backward(ctx, dout):
        xg, gate, S, sw, w_nn, b_nn, w_bs = ctx.saved_tensors
        L, st = lib(), torch.cuda.current_stream().cuda_stream
        _, B, T, D = xg.shape
        dev = xg.device
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)      # noqa: E731
        two = lambda t: ptr_array([t.data_ptr(), t.data_ptr()])                # noqa: E731
        dx0 = torch.zeros(2, B, 2 * T, D, dtype=torch.float32, device=dev)
        dx0[0] = dout
        dZ, dS, rows, dxg = f(2, T, D), f(2, T, D), f(2, T, 2), f(2, B, T, D)
        dw_nn, db_nn, dw_bs, db_bs = f(2, D, D), f(2, D), f(2, 1, B), f(2, 1)
        per = lambda t: ptr_array([t[0].data_ptr(), t[1].data_ptr()])          # noqa: E731
        L.call("amid_bert_comp_bwd_f32", xg.data_ptr(), dx0.data_ptr(), gate.data_ptr(), S.data_ptr(), sw.data_ptr(), two(w_nn), two(b_nn),
               two(w_bs), ctx.cross, B, T, D, dZ.data_ptr(), dS.data_ptr(), rows.data_ptr(), per(dw_nn), per(db_nn), per(dw_bs), per(db_bs),
               dxg.data_ptr(), st)
        # slot 1 received no gradient, so module 1's shares are exact zeros: slot 0 / module 0 carry everything
        d_a = dxg[0]
        d_b = dxg[1] if ctx.cross else None
        return None, None, d_a, d_b, dw_nn[0], db_nn[0], dw_bs[0], db_bs[0]


def _comp_check(mod, seq):
    lib()
    if seq.dim() != 3 or not seq.is_cuda:
        raise ValueError("expected a [b, n, d] tensor on the GPU")
    if seq.shape[0] != mod.bs:
        raise ValueError(f"the batch must hold exactly bs = {mod.bs} rows (trans_bs is Linear(bs, 1) over the batch), got {seq.shape[0]}")


class InnerComp(nn.Module):
    def __init__(self, user_emb_dim, bs, threshold):   # model_seq.py:450-457
        super().__init__()
        self.bs, self.threshold = bs, threshold
        self.trans_nn = nn.Linear(user_emb_dim, user_emb_dim)
        self.trans_bs = nn.Linear(bs, 1)

    def forward(self, seq):                            # model_seq.py:459-472: [b, n, d] -> [b, 2n, d]
        _comp_check(self, seq)
        return _CompFunction.apply(0, self.threshold, seq, seq.detach(), self.trans_nn.weight, self.trans_nn.bias, self.trans_bs.weight,
                                   self.trans_bs.bias)


class InterComp(nn.Module):
    def __init__(self, user_emb_dim, bs, threshold):   # model_seq.py:474-481
        super().__init__()
        self.bs, self.threshold = bs, threshold
        self.trans_nn = nn.Linear(user_emb_dim, user_emb_dim)
        self.trans_bs = nn.Linear(bs, 1)

    def forward(self, seq_d1, seq_d2):                 # model_seq.py:483-497: information seq_d2 --> seq_d1, [b, 2n, d]
        _comp_check(self, seq_d1)
        _comp_check(self, seq_d2)
        if seq_d1.shape != seq_d2.shape:
            raise ValueError("seq_d1 and seq_d2 must have the same shape")
        return _CompFunction.apply(1, self.threshold, seq_d1, seq_d2, self.trans_nn.weight, self.trans_nn.bias, self.trans_bs.weight,
                                   self.trans_bs.bias)


class GRU4Rec(nn.Module):
    def __init__(self, user_length, user_emb_dim, item_length, item_emb_dim, seq_len, hid_dim, bs, isInC, isItC, threshold1, threshold2,
                 isDR=False):                          # model_seq.py:58
        super().__init__()
        _not_built("GRU4Rec (recurrent encoder, not the attention path)", "model_seq.py:56-113")


class BERT4Rec(SASRec):
    """model_seq.py:248-309 on the HIP engine (isDR either way; isInC or isItC -- the reference fails with both -- put the comp
    module in front of the encoders, which then see 2 * seq_len tokens, :283-294): two stacks of two TransformerBlocks whose
    hidden size 128 / 4 heads / FFN 512 / dropout 0.1 the reference hard-codes (:264-267), so emb dims must be 128; no
    positional embedding; ONE key mask from seq_d2 > 0 for both stacks (:288); plain mean over time, then predictModule.
    Same constructor, forward signature, state_dict keys (transform{1,2}.{0,1}.*) and train_step() as SASRec above."""

    ENGINE_CLS = Bert4recEngine
    COMP_IN_FRONT = True         # BERT4Rec applies InnerComp / InterComp BEFORE the encoders (2T tokens, :283-294)
