"""Data side of the hot path (reference: dataset_seq.py:12-22, :131-274), pre-tokenised.

The reference parses two JSON lists per sample inside ``__getitem__``, builds a Python set difference
per sample for the negatives and collates to float32 tensors (2.5 k samples/s per worker,
SURVEY.md section 3(D)).  Here the CSV is tokenised ONCE into left-padded int64 ``[N, T]`` arrays
(same ``seq_padding`` rule, same positive / duplicate-removal rule), negatives are drawn per epoch
ON THE DEVICE with the same constraint (uniform over the domain's item pool minus the row's own
sequence, without replacement: ``amid_sample_negatives_i64``) and batches are index_select-ed on the device.  Ids stay integers
end to end (the reference's float32 wire format is exact only below 2**24).
"""
from __future__ import annotations

import json
from typing import Dict, Iterator, List

import numpy as np
import torch


def seq_padding(seq: List[int], length_enc: int, long_length: int, pad_id: int):
    """dataset_seq.py:12-22 (called with length_enc = seq_len + 1): left-pad / keep the tail."""
    long_mask = 1 if len(seq) >= long_length else 0
    if len(seq) >= length_enc:
        enc = list(seq[-length_enc + 1:])
    else:
        enc = [pad_id] * (length_enc - len(seq) - 1) + list(seq)
    return enc, long_mask


class DualDomainSeqDataset:
    """Same constructor as the reference (dataset_seq.py:131-132)."""

    def __init__(self, seq_len, isTrain, neg_nums, long_length, pad_id, csv_path="", seed: int = 0):
        import pandas as pd
        df = pd.read_csv(csv_path)
        self.seq_len, self.isTrain, self.neg_nums, self.long_length, self.pad_id = seq_len, isTrain, neg_nums, long_length, pad_id
        raw1 = [json.loads(s) for s in df["seq_d1"].tolist()]
        raw2 = [json.loads(s) for s in df["seq_d2"].tolist()]
        self.domain_id = np.asarray(df["domain_id"].tolist(), dtype=np.int64)
        self.user_nodes = np.asarray(df["user_id"].tolist(), dtype=np.int64)
        # observation label of the doubly-robust trainer's second loader (DualDomainSeqDatasetDR, dataset_seq.py:453); zeros if absent
        self.ob_label = (np.asarray(df["ob_label"].tolist(), dtype=np.int64) if "ob_label" in df.columns
                         else np.zeros(len(df), dtype=np.int64))
        self.pool = [np.array(sorted({i for s in raw1 for i in s}), dtype=np.int64),
                     np.array(sorted({i for s in raw2 for i in s}), dtype=np.int64)]      # dataset_seq.py:141-142
        N, T = len(raw1), seq_len
        self.seq_d1 = np.empty((N, T), dtype=np.int64)
        self.seq_d2 = np.empty((N, T), dtype=np.int64)
        self.i_node = np.empty(N, dtype=np.int64)
        self.overlap_label = np.zeros(N, dtype=np.int64)
        self.long_tail_mask_d1 = np.zeros(N, dtype=np.int64)
        self.long_tail_mask_d2 = np.zeros(N, dtype=np.int64)
        self.own_items: List[np.ndarray] = []
        for r in range(N):
            s1, s2 = list(raw1[r]), list(raw2[r])
            self.overlap_label[r] = 1 if (len(s1) and len(s2)) else 0                  # :181-184
            own = s1 if self.domain_id[r] == 0 else s2
            self.own_items.append(np.unique(np.asarray(own, dtype=np.int64)))           # excluded from the negatives (:188/:206)
            item = own[-1]                                                              # positive = last item (:189/:207)
            own = [x for x in own[:-1] if x != item]                                    # :190-195
            if self.domain_id[r] == 0:
                s1 = own
            else:
                s2 = own
            self.i_node[r] = item
            e1, m1 = seq_padding(s1, T + 1, long_length, pad_id)
            e2, m2 = seq_padding(s2, T + 1, long_length, pad_id)
            self.seq_d1[r], self.seq_d2[r] = e1, e2
            self.long_tail_mask_d1[r], self.long_tail_mask_d2[r] = m1, m2
        self.rng = np.random.default_rng(seed)

    def __len__(self) -> int:
        return len(self.i_node)

    @classmethod
    def from_tokenised(cls, path: str, isTrain: bool = True, neg_nums: int = 1, seed: int = 0) -> "DualDomainSeqDataset":
        """A dataset from a tokenised fixture (tests/golden/make_tokenised.py: the arrays the REFERENCE's DualDomainSeqDataset
        produced for a CSV -- dataset_seq.py:177-248 --, its two item pools and every row's own item set) instead of a CSV: the
        same object the constructor builds, without pandas / JSON.  `ref_neg` keeps the negative the reference drew for every row
        (a valid draw: DeviceBatches(negatives="fixture") trains on exactly those)."""
        z = np.load(path)
        ds = cls.__new__(cls)
        ds.seq_len, ds.isTrain, ds.neg_nums = int(z["seq_len"]), isTrain, neg_nums
        ds.long_length, ds.pad_id = int(z["long_length"]), int(z["pad_id"])
        i64 = lambda k: z[k].astype(np.int64)       # noqa: E731
        ds.seq_d1, ds.seq_d2, ds.i_node = i64("seq_d1"), i64("seq_d2"), i64("i_node")
        ds.domain_id, ds.user_nodes, ds.overlap_label = i64("domain_id"), i64("user_node"), i64("overlap_label")
        ds.long_tail_mask_d1, ds.long_tail_mask_d2 = i64("long_tail_mask_d1"), i64("long_tail_mask_d2")
        ds.ob_label = np.zeros(len(ds.i_node), dtype=np.int64)
        ds.pool = [i64("pool_d1"), i64("pool_d2")]
        own, off = i64("own"), z["own_off"].astype(np.int64)
        ds.own_items = [own[off[r]:off[r + 1]] for r in range(len(ds.i_node))]
        ds.ref_neg = i64("neg_samples")
        ds.rng = np.random.default_rng(seed)
        return ds

    def max_item_id(self) -> int:
        """Largest table row this dataset can ask for: its items, its sequences' ids other than the pad id, its negative pools."""
        tops = [int(self.i_node.max())] + [int(p.max()) for p in self.pool if len(p)]
        for a in (self.seq_d1, self.seq_d2):
            real = a[a != self.pad_id]
            if real.size:
                tops.append(int(real.max()))
        return max(tops)

    def shift_items(self, offset: int) -> "DualDomainSeqDataset":
        """Move every item id except the pad id by `offset` (in place; returns self): the joint mode puts a second dataset's items
        behind the first one's in the shared table (cfg 4)."""
        sh = lambda a: np.where(a == self.pad_id, a, a + offset)     # noqa: E731
        self.seq_d1, self.seq_d2, self.i_node = sh(self.seq_d1), sh(self.seq_d2), self.i_node + offset
        self.pool = [p + offset for p in self.pool]
        self.own_items = [o + offset for o in self.own_items]
        if hasattr(self, "ref_neg"):
            self.ref_neg = self.ref_neg + offset
        return self


class DeviceBatches:
    """DataLoader(batch_size, shuffle, drop_last=True) over a tokenised dataset resident on the device."""

    def __init__(self, ds: DualDomainSeqDataset, batch_size: int, shuffle: bool, device, seed: int = 0, rank: int = 0, world: int = 1,
                 negatives: str = "device"):
        """rank / world: data parallel -- every rank walks the same shuffled order (same seed) in global batches of
        world x batch_size rows and keeps its own contiguous slice (DistributedSampler-style, drop_last).
        negatives: "device" = drawn per epoch by amid_sample_negatives_i64; "fixture" = the draw stored with a tokenised fixture
        (DualDomainSeqDataset.from_tokenised: what the reference's random.sample produced under random.seed(0); needs no GPU)."""
        self.ds, self.bs, self.shuffle, self.device = ds, batch_size, shuffle, torch.device(device)
        self.rank, self.world = rank, world
        if negatives not in ("device", "fixture"):
            raise ValueError(f"negatives must be 'device' or 'fixture', got {negatives!r}")
        if negatives == "fixture" and (not hasattr(ds, "ref_neg") or ds.ref_neg.shape[1] < (1 if ds.isTrain else ds.neg_nums)):
            raise ValueError("negatives='fixture' needs a tokenised dataset carrying enough stored negatives")
        self.negatives = negatives
        to = lambda a: torch.from_numpy(a).to(self.device)       # noqa: E731
        self.t = dict(user_node=to(ds.user_nodes), i_node=to(ds.i_node), seq_d1=to(ds.seq_d1), seq_d2=to(ds.seq_d2),
                      domain_id=to(ds.domain_id), overlap_label=to(ds.overlap_label), ob_label=to(ds.ob_label),
                      long_tail_mask_d1=to(ds.long_tail_mask_d1),
                      long_tail_mask_d2=to(ds.long_tail_mask_d2))
        self.gen = torch.Generator().manual_seed(seed)
        # negative sampling state on the device: the two item pools and every row's own items (dataset_seq.py:141-142, :188/:206)
        self.pool = [to(p) for p in ds.pool]
        off = np.zeros(len(ds) + 1, dtype=np.int32)
        np.cumsum([len(o) for o in ds.own_items], out=off[1:])
        self.own_off = torch.from_numpy(off).to(self.device)
        self.own = to(np.concatenate(ds.own_items) if len(ds) else np.zeros(0, np.int64))
        self.seed, self.epoch = int(seed), 0
        k = 1 if ds.isTrain else ds.neg_nums
        self.k = k
        for r, o in enumerate(ds.own_items):
            if k + len(o) > len(ds.pool[int(ds.domain_id[r] != 0)]):
                raise ValueError("negative pool smaller than neg_nums")
        self.label = torch.zeros(batch_size, 1 + k, device=self.device)
        self.label[:, 0] = 1.0                                                            # dataset_seq.py:191,199

    def sample_negatives(self) -> torch.Tensor:
        """[N, k] negatives for one epoch, drawn ON THE DEVICE (amid_sample_negatives_i64): uniform without replacement from the
        row's domain pool minus its own sequence -- the reference's random.sample(pool - set(seq), k) (dataset_seq.py:198, :215).
        Every rank of a data-parallel run draws the same table (same seed, same epoch counter)."""
        if self.negatives == "fixture":
            self.epoch += 1
            return torch.from_numpy(np.ascontiguousarray(self.ds.ref_neg[:, : self.k])).to(self.device)
        from ._lib import lib
        N = len(self.ds)
        out = torch.empty(N, self.k, dtype=torch.int64, device=self.device)
        self.epoch += 1
        lib().call("amid_sample_negatives_i64", self.pool[0].data_ptr(), self.pool[0].numel(), self.pool[1].data_ptr(),
                   self.pool[1].numel(), self.own.data_ptr(), self.own_off.data_ptr(), self.t["domain_id"].data_ptr(), N, self.k,
                   self.seed, self.epoch, out.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream)
        return out

    def __len__(self) -> int:
        return len(self.ds) // (self.bs * self.world)                                     # drop_last=True (train_sr.py:452,455)

    def _new_epoch(self):
        """Negatives and row order of the next epoch (advances the sampler's epoch counter and the shuffle generator)."""
        neg = self.sample_negatives()
        n = len(self.ds)
        order = torch.randperm(n, generator=self.gen) if self.shuffle else torch.arange(n)
        return neg, order.to(self.device)

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        neg, order = self._new_epoch()
        for b in range(len(self)):
            lo = (b * self.world + self.rank) * self.bs
            sel = order[lo:lo + self.bs]
            batch = {k: v.index_select(0, sel) for k, v in self.t.items()}
            batch["neg_samples"] = neg.index_select(0, sel)
            batch["label"] = self.label
            yield batch

    def epoch_tensors(self) -> Dict[str, torch.Tensor]:
        """The batches __iter__ would yield for the next epoch, stacked: every value gains a leading [n_batches] axis (label stays
        [bs, 1 + k]).  Feeds SasrecEngine.pack_epoch / set_input_pool: the whole epoch becomes resident in HBM with a handful of
        device ops, and the train loop replays one graph per step with no per-step tensor work on the host."""
        neg, order = self._new_epoch()
        nb = len(self)
        rows = (torch.arange(nb, device=self.device).unsqueeze(1) * self.world + self.rank) * self.bs + torch.arange(self.bs, device=self.device)
        sel = order[rows.reshape(-1)]
        out = {k: v.index_select(0, sel).reshape(nb, self.bs, *v.shape[1:]) for k, v in self.t.items()}
        out["neg_samples"] = neg.index_select(0, sel).reshape(nb, self.bs, -1)
        out["label"] = self.label
        return out


class JointBatches:
    """Two datasets trained as ONE job on a shared table (BASELINE.json configs[3]: "mybank loan_fund + loan_account joint train";
    not a reference mode -- the reference trains one -dm per run, train_sr.py:447-457 -- but defined by SURVEY.md section 8(d)):
    the second dataset's item ids sit `offset` rows behind the first's (DualDomainSeqDataset.shift_items; the pad row is shared),
    and the batches of the two loaders ALTERNATE (a0, b0, a1, b1, ...; the longer loader's surplus at the end), each batch
    holding rows of one dataset only so that negatives come from that dataset's own pools.  Same iteration / epoch_tensors
    interface as DeviceBatches; under data parallelism each loader shards its global batches by rank as before."""

    def __init__(self, a: DeviceBatches, b: DeviceBatches):
        if a.bs != b.bs or a.k != b.k or a.ds.seq_len != b.ds.seq_len or a.ds.pad_id != b.ds.pad_id:
            raise ValueError("joint loaders must agree on batch size, negatives per row, seq_len and pad id")
        self.a, self.b, self.bs, self.k = a, b, a.bs, a.k
        self.device, self.rank, self.world, self.label = a.device, a.rank, a.world, a.label

    def __len__(self) -> int:
        return len(self.a) + len(self.b)

    def order(self):
        """[(loader index, batch index)] of one epoch."""
        na, nb = len(self.a), len(self.b)
        out = []
        for i in range(max(na, nb)):
            if i < na:
                out.append((0, i))
            if i < nb:
                out.append((1, i))
        return out

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        ia, ib = iter(self.a), iter(self.b)
        for which, _ in self.order():
            yield next(ia if which == 0 else ib)

    def epoch_tensors(self) -> Dict[str, torch.Tensor]:
        ea, eb = self.a.epoch_tensors(), self.b.epoch_tensors()
        na = len(self.a)
        pos = torch.tensor([i if w == 0 else na + i for w, i in self.order()], device=self.device)
        out = {k: torch.cat((ea[k], eb[k]), 0).index_select(0, pos) for k in ea if k != "label"}
        out["label"] = self.label
        return out
