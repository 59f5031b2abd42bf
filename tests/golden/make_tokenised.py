#!/usr/bin/env python3
"""Tokenise the reference's training CSVs by RUNNING THE REFERENCE's own DualDomainSeqDataset (dataset_seq.py:131-248) and store
the resulting arrays as fixtures: tests/golden/tok_<name>.npz.  Runs only in the build container (needs /root/reference).

The fixtures are data: for every CSV row the reference's __getitem__ outputs (left-padded seq_d1 / seq_d2, positive item, domain id,
masks, the negative it drew under random.seed(0)), the two item pools, every row's own item set (what the negatives must avoid),
and the raw JSON strings of the first 64 rows (the INPUT of the tokeniser, so that the product tokeniser can be checked against the
reference's output wherever the fixture travels).  They serve
  * tests/test_tokeniser.py: amid_amd.dataset_seq.DualDomainSeqDataset == the reference, row for row;
  * bench.py: the headline workload is drawn from the REAL cloth_sport_train75 batches (data: "real"), not a statistical model;
  * the joint mybank mode (BASELINE.json configs[3]): loan_fund + loan_account rows.

    python tests/golden/make_tokenised.py
"""
import json
import os
import random
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
import dataset_seq  # noqa: E402

ITEM_LENGTH = 447410          # train_sr.py:450
PAD_ID = ITEM_LENGTH + 1      # train_sr.py:451
SETS = [("cloth_sport_train75", "amazon_dataset", 50), ("cloth_sport_train25", "amazon_dataset", 50),
        ("phone_elec_train25", "amazon_dataset", 50), ("loan_fund_train75", "mybank_dataset", 20),
        ("loan_account_train75", "mybank_dataset", 20)]


def main():
    for name, folder, T in SETS:
        path = os.path.join(REF, folder, name + ".csv")
        ds = dataset_seq.DualDomainSeqDataset(seq_len=T, isTrain=True, neg_nums=199, long_length=7, pad_id=PAD_ID, csv_path=path)
        n = len(ds.user_nodes)
        random.seed(0)
        cols = {k: [] for k in ("user_node", "i_node", "seq_d1", "seq_d2", "long_tail_mask_d1", "long_tail_mask_d2", "domain_id",
                                "overlap_label", "neg_samples")}
        own, off = [], [0]
        for i in range(n):
            s = ds[i]
            for k in cols:
                cols[k].append(np.asarray(s[k]).reshape(-1))
            raw = json.loads(ds.seq_d1[i] if ds.domain_id[i] == 0 else ds.seq_d2[i])
            u = sorted(set(raw))
            own += u
            off.append(len(own))
        out = {k: np.stack(v).astype(np.int32) for k, v in cols.items()}
        for k in ("user_node", "i_node", "long_tail_mask_d1", "long_tail_mask_d2", "domain_id", "overlap_label"):
            out[k] = out[k].reshape(-1)
        out["pool_d1"] = np.array(sorted(ds.item_pool_d1), dtype=np.int32)
        out["pool_d2"] = np.array(sorted(ds.item_pool_d2), dtype=np.int32)
        out["own"] = np.array(own, dtype=np.int32)
        out["own_off"] = np.array(off, dtype=np.int32)
        out["raw_user_id"] = np.array(ds.user_nodes[:64], dtype=np.int64)
        out["raw_seq_d1"] = np.array(ds.seq_d1[:64]).astype(str)
        out["raw_seq_d2"] = np.array(ds.seq_d2[:64]).astype(str)
        out["raw_domain_id"] = np.array(ds.domain_id[:64], dtype=np.int64)
        dst = os.path.join(OUT, f"tok_{name}.npz")
        np.savez_compressed(dst, seq_len=T, pad_id=PAD_ID, long_length=7, n_rows=n, **out)
        print(name, n, "rows ->", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
