"""Host-side helpers of the training driver (reference: utils.py): running means, logger, ranking metrics."""
from __future__ import annotations

import logging
from pathlib import Path

import numpy as np


class AverageMeter:
    """utils.py:262-280: running mean per named quantity."""

    def __init__(self, *keys: str):
        self._sum = dict.fromkeys(keys, 0.0)
        self._n = dict.fromkeys(keys, 0)

    def update(self, **kw: float) -> None:
        for k, v in kw.items():
            if k not in self._sum:
                raise KeyError(k)
            self._sum[k] += float(v)
            self._n[k] += 1

    def __getattr__(self, k: str) -> float:
        if k.startswith("_") or k not in self._sum:
            raise AttributeError(k)
        return self._sum[k] / self._n[k] if self._n[k] else 0.0


def init_logger(log_dir: str, log_file: str) -> None:
    """utils.py:282-294: root logger to stdout + <log_dir>/<log_file>."""
    fmt = r"[%(asctime)s] %(message)s"
    logging.basicConfig(level=logging.INFO, datefmt=r"%Y/%m/%d %H:%M:%S", format=fmt)
    logging.getLogger().setLevel(logging.INFO)          # basicConfig is a no-op when a host application already installed handlers
    d = Path(log_dir)
    d.mkdir(parents=True, exist_ok=True)
    fh = logging.FileHandler(str(d / log_file))
    fh.setFormatter(logging.Formatter(fmt))
    logging.getLogger().addHandler(fh)


def positive_ranks(pred: np.ndarray) -> np.ndarray:
    """Rank (0 = best) of column 0 among each row's scores, descending -- the reference's double argsort (utils.py:297)."""
    return (-pred).argsort().argsort()[:, 0]


def get_sample_scores(pred: np.ndarray):
    """utils.py:296-312 -> (HIT@1, NDCG@1, HIT@5, NDCG@5, HIT@10, NDCG@10, MRR), means over rows."""
    rank = positive_ranks(pred).astype(np.float64)
    n = float(len(rank))
    out = []
    for k in (1, 5, 10):
        hit = rank < k
        out += [hit.sum() / n, (hit / np.log2(rank + 2.0)).sum() / n]
    return tuple(out) + ((1.0 / (rank + 1.0)).sum() / n,)


FIX_VALUE_DOC = "train_sr.py:42,114-115: pred[:, 0] -= 1e-7 before ranking, so a tie between the positive and a negative counts against the positive"


def scores_from_ranks(rank) -> tuple:
    """The seven numbers of get_sample_scores from positive ranks (numpy array or torch tensor, any device)."""
    import torch
    r = torch.as_tensor(rank).to(torch.float64)
    n = float(r.numel())
    if n == 0:
        return (float("nan"),) * 7
    out = []
    for k in (1, 5, 10):
        hit = (r < k).to(torch.float64)
        out += [float(hit.sum() / n), float((hit / torch.log2(r + 2.0)).sum() / n)]
    return tuple(out) + (float((1.0 / (r + 1.0)).sum() / n),)


def device_positive_ranks(p1, p2, domain_id, fix_value: float):
    """Positive ranks on the GPU (libamid_hip: amid_positive_rank_f32), one int per row, judged by the row's own domain head."""
    import torch
    from ._lib import lib
    B, NI = p1.shape
    rank = torch.empty(B, dtype=torch.int32, device=p1.device)
    lib().call("amid_positive_rank_f32", p1.contiguous().data_ptr(), p2.contiguous().data_ptr(), domain_id.contiguous().data_ptr(), B, NI,
               float(fix_value), rank.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return rank


def choose_predict(pred_d1: np.ndarray, pred_d2: np.ndarray, domain_id: np.ndarray):
    """utils.py:21-40: rows of domain 0 are judged by the d1 head, rows of domain 1 by the d2 head."""
    d = domain_id.reshape(len(pred_d1), -1)[:, 0]
    return pred_d1[d == 0], pred_d2[d == 1]


def choose_predict_overlap(pred_d1, pred_d2, domain_id, overlap_label):
    """utils.py:42-68: additionally split by overlapped / non-overlapped users."""
    d = domain_id.reshape(len(pred_d1), -1)[:, 0]
    o = overlap_label.reshape(len(pred_d1), -1)[:, 0]
    return pred_d1[(d == 0) & (o == 1)], pred_d1[(d == 0) & (o == 0)], pred_d2[(d == 1) & (o == 1)], pred_d2[(d == 1) & (o == 0)]
