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



class DeviceBatches:
    """DataLoader(batch_size, shuffle, drop_last=True) over a tokenised dataset resident on the device."""

    def __init__(self, ds: DualDomainSeqDataset, batch_size: int, shuffle: bool, device, seed: int = 0, rank: int = 0, world: int = 1):
        """rank / world: data parallel -- every rank walks the same shuffled order (same seed) in global batches of
        world x batch_size rows and keeps its own contiguous slice (DistributedSampler-style, drop_last)."""
        self.ds, self.bs, self.shuffle, self.device = ds, batch_size, shuffle, torch.device(device)
        self.rank, self.world = rank, world
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
