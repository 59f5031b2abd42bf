#!/usr/bin/env python3
"""Headline benchmark: SASRec training throughput (train samples/s) on the reference's
cloth_sport_train75 configuration (BASELINE.json configs[1]): batch 256 per GPU, seq_len 50,
emb_dim 128, hid 32, 1 negative, fp32, item table 2 x 447 410 rows (train_sr.py:450,456), dropout on.

    python bench.py --gpus N --steps K --warmup W

N > 1 either way: under a launcher that has set WORLD_SIZE / RANK / LOCAL_RANK (python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...) this process is one rank; started plainly (python bench.py --gpus N) it
starts that launcher itself as a CHILD process -- before anything here has touched the GPU, never by exec -- relays
rank 0's JSON line and exits with the launcher's code.

A "step" is the loop body of the reference's train() (train_sr.py:190-217): forward, masked BCE,
backward, Adam.  Default data: the REAL cloth_sport_train75 epoch -- what the reference's own
DualDomainSeqDataset.__getitem__ returned for every row of the CSV, with its negative draw
(tests/golden/tok_cloth_sport_train75.npz, written by tests/golden/make_tokenised.py) -- packed once and
resident in HBM before the timed region; --data synthetic draws batches to the file's statistics instead
(89 % pad positions, ids <= 42 441, pad id 447 411).  Random-init weights.
Prints ONE JSON line (rank 0).

N = 1   the whole step is one hipGraph replay (graphs of four consecutive steps).
N > 1   weak scaling (256 samples per GPU), one exchange per step (amid_amd/dist.py): graph A (local gradients +
        this rank's packed chunk), the collective(s), graph B (Adam over the world's chunks).  --dense-exchange
        gather: the 1.7 MB dense gradient rides behind the sparse rows in ONE RCCL all-gather; allreduce: its own
        RCCL all-reduce beside the all-gather of the sparse rows.  The line's "dist" object carries the
        collectives and bytes per step of the run.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

B, T, D, HID, NEG = 256, 50, 128, 32, 1
ITEM_LENGTH = 447410              # train_sr.py:450
N_ROWS = 2 * ITEM_LENGTH          # train_sr.py:456
PAD_ID = ITEM_LENGTH + 1          # train_sr.py:451
MAX_REAL_ID = 42441               # largest id in cloth_sport_train75
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak
# how the weight gradients' products run, taken from the ENGINE once a step has run (wgrad_kind_of below): six bf16 piece pairs per fp32
# product by default, operands rounded to bf16 with --dtype bf16, fp32 matrix instructions where the engine says so
WGRAD_KIND = "mfma16x6"


def wgrad_kind_of(eng, model: str) -> str:
    if model == "bert4rec":
        return {"9": "mfma16x9", "6": "mfma16x6"}.get(eng.WGRAD_SPLIT, "mfma")
    return {1: "mfma16", 2: "mfma16x9", 3: "mfma16x6"}.get(eng._wgrad_mode(eng.D), "mfma")
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 matrix peak (the 5 PF headline figure includes 2:1 sparsity)
PEAK_HBM_GBPS = 8000.0

# --workload: the default is the configuration BASELINE.json's metric is quoted on (configs[1]); the two cfg5 entries are the
# synthetic gather / scatter stress of SURVEY.md section 8(d) (table 10 M x 128 fp32 = 5.12 GB >> the 256 MiB Infinity Cache,
# batch 4096 per GPU): reported in DESIGN.md / profiles/, never the headline line.
CFG5_ROWS = 10_000_000
WORKLOADS = {
    # BASELINE.json configs[0]: the reference's CPU-runnable plumbing case, at its defaults -- batch 64, emb_dim 64 (8 heads of 8 dims;
    # train_sr.py:360-364) -- on the real cloth_sport_train25 epoch.  A side measurement like cfg3..cfg5
    "cfg1": dict(B=64, D=64, n_rows=N_ROWS, pad_id=PAD_ID, max_id=MAX_REAL_ID, kind="real",
                 label="cloth_sport_train25 SASRec train step at the reference's defaults, batch 64, emb_dim 64 (BASELINE.json configs[0])"),
    "cfg2": dict(B=256, n_rows=N_ROWS, pad_id=PAD_ID, max_id=MAX_REAL_ID, kind="real",
                 label="cloth_sport_train75-shaped SASRec train step (BASELINE.json configs[1])"),
    # BASELINE.json configs[2] / [3], shaped after the data statistics of SURVEY.md section 8(d); side measurements like cfg5
    "cfg3": dict(B=512, n_rows=N_ROWS, pad_id=PAD_ID, max_id=89987, kind="real", mean_len=8.0,
                 label="phone_elec_train25-shaped SASRec train step, batch 512 (BASELINE.json configs[2]; quoted with --dtype bf16)"),
    "cfg4": dict(B=256, T=20, n_rows=N_ROWS, pad_id=PAD_ID, max_id=123132, kind="real", mean_len=2.2,
                 label="mybank loan_fund + loan_account shaped SASRec train step, seq_len 20, batch 256 per GPU (BASELINE.json configs[3])"),
    "cfg5-uniform": dict(B=4096, n_rows=CFG5_ROWS + 2, pad_id=CFG5_ROWS + 1, max_id=CFG5_ROWS - 1, kind="uniform",
                         label="synthetic S-uniform (BASELINE.json configs[4]): every position a uniform id in [0, 10M), no pads"),
    "cfg5-real": dict(B=4096, n_rows=CFG5_ROWS + 2, pad_id=CFG5_ROWS + 1, max_id=CFG5_ROWS - 1, kind="zipf",
                      label="synthetic S-real (BASELINE.json configs[4]): cloth_sport length histogram (89 % pads), ids Zipf(1.05)"),
}


def synth_batch(gen, device, wl=None):
    """One batch shaped like collate_fn_enhance's output after train_sr.py:191-200."""
    wl = wl or WORKLOADS["cfg2"]
    Bw, pad, hi, kind = wl["B"], wl["pad_id"], wl["max_id"], wl["kind"]
    T, mean_len = wl.get("T", globals()["T"]), wl.get("mean_len", 4.5)

    def ids(shape):
        if kind == "zipf":            # Zipf(1.05) ranks folded into [1, hi]: a few hot rows, a long cold tail
            import numpy as np
            rng = np.random.default_rng(int(torch.randint(0, 2 ** 31 - 1, (1,), generator=gen)))
            n = 1
            for d in shape:
                n *= d
            return torch.from_numpy(((rng.zipf(1.05, n) - 1) % hi + 1).astype("int64")).reshape(shape)
        return torch.randint(0 if kind == "uniform" else 1, hi + 1, shape, generator=gen)

    if kind == "uniform":
        seqs = [ids((Bw, T)), ids((Bw, T))]
    else:
        lens1 = torch.clamp(torch.poisson(torch.full((Bw,), mean_len), generator=gen).long() + 1, max=T)
        lens2 = torch.clamp(torch.poisson(torch.full((Bw,), mean_len), generator=gen).long(), max=T)      # other domain may be empty
        col = torch.arange(T).unsqueeze(0)
        seqs = []
        for lens in (lens1, lens2):
            s = torch.full((Bw, T), pad, dtype=torch.long)
            v = ids((Bw, T))
            real = col >= (T - lens).unsqueeze(1)
            s[real] = v[real]
            seqs.append(s)
    label = torch.zeros(Bw, 1 + NEG)
    label[:, 0] = 1.0
    b = dict(i_node=ids((Bw,)), neg_samples=ids((Bw, NEG)), seq_d1=seqs[0], seq_d2=seqs[1], label=label,
             domain_id=(torch.rand(Bw, generator=gen) < 0.5).long())
    return {k: v.to(device) for k, v in b.items()}


FIXTURES = os.path.join(ROOT, "tests", "golden")
REAL = {   # --workload -> tokenised fixtures (tests/golden/make_tokenised.py: the arrays the reference's own DualDomainSeqDataset produced)
    "cfg1": ("cloth_sport_train25",),
    "cfg2": ("cloth_sport_train75",),
    "cfg3": ("phone_elec_train25",),          # phone_elec_train75 is not in the reference checkout (.MISSING_LARGE_BLOBS): BASELINE.md section 4
    "cfg4": ("loan_fund_train75", "loan_account_train75"),      # joint mode: SURVEY.md section 8(d)
}
JOINT_OFFSET = ITEM_LENGTH + 2    # loan_account's items sit behind loan_fund's in the shared table (the pad row is shared)


def real_batches(workload, Bw, device, rank, world, seed=0):
    """The workload's real training batches: the reference's tokenisation (fixtures), its negative draw, a seeded shuffle; one epoch,
    resident on the device.  Returns (epoch tensors with a leading [n_batches] axis, description)."""
    from amid_amd.dataset_seq import DeviceBatches, DualDomainSeqDataset, JointBatches
    names = REAL[workload]
    dss = [DualDomainSeqDataset.from_tokenised(os.path.join(FIXTURES, f"tok_{n}.npz")) for n in names]
    if len(dss) == 2:
        dss[1].shift_items(JOINT_OFFSET)
        assert dss[1].max_item_id() < N_ROWS, "joint mode: the second dataset's shifted ids must fit the 2 x item_length table"
    loaders = [DeviceBatches(d, Bw, shuffle=True, device=device, seed=seed, rank=rank, world=world, negatives="fixture") for d in dss]
    ld = loaders[0] if len(loaders) == 1 else JointBatches(*loaders)
    desc = " + ".join(names) + f" ({sum(len(d) for d in dss)} rows, tokenised by the reference's DualDomainSeqDataset; negatives: its draw)"
    return ld, desc


def init_params(eng, seed):
    """Random-init weights of the reference architecture (nn.Embedding N(0,1); small dense weights)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        if eng.n_rows > 2_000_000:            # cfg5: 5 GB of N(0,1) drawn on the device (seeded), not on the host
            dg = torch.Generator(device=eng.device).manual_seed(seed)
            eng.table.normal_(generator=dg)
        else:
            eng.table.copy_(torch.randn(eng.n_rows, eng.D, generator=g))
        flat = torch.randn(eng.dense.numel, generator=g) * 0.05
        eng.dense.data.copy_(flat)
        for name in eng.dense.slots:
            v = eng.dense.view(name)
            if ("layernorm" in name and name.endswith("weight")) or name.endswith("norm.a_2"):
                v.fill_(1.0)
            elif name.endswith("norm.b_2"):
                v.zero_()
            elif name.endswith("bias"):
                v.zero_()
            elif name.endswith("pos_emb.weight"):
                v.copy_(torch.randn(v.shape, generator=g))
    torch.cuda.synchronize()


def algorithmic_work(Bw=B, n_uniq=None, T=T):
    """Per launch, at this workload (formulas: DESIGN.md section 5 / SURVEY.md section 8(d)).  BERT4Rec: 24 B T D^2 per
    domain-layer forward (4 D^2 projections + the 128 -> 512 -> 128 feed-forward), the same again twice for backward.
    The fused SASRec train step encodes and differentiates the LIVE sequences only -- of each sample the sequence of its own domain,
    B of the 2 B (the loss multiplies the other domain's BCE by zero, train_sr.py:205-211): its kernels are priced on B T rows."""
    M2 = 2 * Bw * T
    n_idx = M2 + Bw * (1 + NEG)
    U = n_uniq if n_uniq is not None else n_idx
    gemm = 2.0 * M2 * D * D                     # FLOP of one [M2, D] x [D, D] projection
    gl = gemm / 2                               # the same over the live rows
    H, hd = 8, D // 8
    return {
        # strip kernels of the fused step ("#k": k-th launch of that entry in a step -- the fused cross-layer launch comes first)
        # the whole encoder forward, both layers: 12 projections over the live rows + the two attention cores (4 T^2 hd per head)
        "amid_sas_seq_fwd_f32": ("mfma", 12 * gl + 2 * 4.0 * T * T * hd * Bw * H),
        # compute=bf16: the same launch with bf16 operands on the projections (fp32 accumulation; the attention core stays fp32): priced on the bf16 peak
        "amid_sas_seq_fwd_bf16w_f32": ("mfma16", 12 * gl + 2 * 4.0 * T * T * hd * Bw * H),
        # the default forward: its twelve projections as six bf16 piece-pair products each (fp32 accuracy), the attention core on fp32 MFMA
        "amid_sas_seq_fwd_split_f32": ("mfma16x6", 12 * gl + 2 * 4.0 * T * T * hd * Bw * H),
        "amid_sas_strip_qkv_fwd_f32": ("mfma", 3 * gl),
        "amid_sas_strip_oproj_ffn_fwd_f32#0": ("mfma", 6 * gl),       # layer 0's out-proj + FFN, layer 1's q / k / v
        "amid_sas_strip_oproj_ffn_fwd_f32#1": ("mfma", 3 * gl),
        "amid_sas_seq_bwd_f32": ("mfma", 12 * gl + 2 * 10.0 * T * T * hd * Bw * H),    # the five launches below as one
        "amid_sas_strip_ffn_bwd_f32": ("mfma", 3 * gl),
        "amid_sas_strip_qkv_bwd_f32#0": ("mfma", 6 * gl),             # layer 1's q / k / v backward, layer 0's FFN / out-proj backward
        "amid_sas_strip_qkv_bwd_f32#1": ("mfma", 3 * gl),
        "amid_attn_fwd_live_f32": ("mfma", 4.0 * T * T * hd * Bw * H),
        "amid_attn_bwd_live_f32": ("mfma", 10.0 * T * T * hd * Bw * H),
        "amid_embed_fwd_live_f32": ("hbm", (Bw * T + Bw * (1 + NEG)) * (4 + 2 * D * 4) + Bw * T * (D // 4)),
        # K1 with the lazy-Adam catch-up folded in: + the stamp of every gathered position (m and v of the few lagging rows: not counted)
        "amid_embed_fwd_replay_f32": ("hbm", (Bw * T + Bw * (1 + NEG)) * (8 + 2 * D * 4) + Bw * T * (D // 4)),
        "amid_embed_fwd_live_compact_f32": ("hbm", (Bw * T + Bw * (1 + NEG)) * (12 + 2 * D * 4) + Bw * T * (D // 4)),
        # the same gather + 24 weights read and written as three bf16 planes by its extra workgroups (the forward's weight images)
        # BERT4Rec's K1: the plain gather of the live sequences + items, and the step's 96 weight-tile images (fp32 read, three bf16 planes written)
        "amid_embed_fwd_tiles_f32": ("hbm", (Bw * T + Bw * (1 + NEG)) * (4 + 2 * D * 4) + 96 * D * D * 10),
        "amid_embed_fwd_w16_f32": ("hbm", (Bw * T + Bw * (1 + NEG)) * (12 + 2 * D * 4) + Bw * T * (D // 4) + 48 * D * D * 10),
        "amid_sas_qkv_fwd_f32": ("mfma", 3 * gemm),
        "amid_sas_oproj_fwd_f32": ("mfma", gemm),
        "amid_sas_oproj_ffn_fwd_f32": ("mfma", 3 * gemm),
        "amid_sas_oproj_ffn_qkv_fwd_f32": ("mfma", 6 * gemm),
        "amid_sas_ffn_fwd_f32": ("mfma", 2 * gemm),
        "amid_sas_ffn_bwd_f32": ("mfma", 3 * gemm),
        "amid_sas_qkv_bwd_f32": ("mfma", 3 * gemm),
        "amid_sas_qkv_ffn_bwd_f32": ("mfma", 6 * gemm),
        "amid_sas_wgrad_f32": (WGRAD_KIND, 12 * gemm),          # both layers in one launch
        # the train step's own backward: of each domain only the samples of that domain carry a gradient (the loss multiplies the
        # other domain's BCE by zero, train_sr.py:205-211), the kernels walk those sequences only -- half the rows, priced as such
        "amid_sas_wgrad_rows_f32": (WGRAD_KIND, 6 * gemm),
        "amid_sas_ffn_bwd_rows_f32": ("mfma", 1.5 * gemm),
        "amid_sas_qkv_bwd_rows_f32": ("mfma", 1.5 * gemm),
        "amid_sas_qkv_ffn_bwd_rows_f32": ("mfma", 3 * gemm),
        "amid_attn_bwd_rows_f32": ("mfma", 10.0 * T * T * hd * Bw * H),
        # BERT4Rec on the strip kernels over the live rows (csrc/bert_strip.hip): 128 x 128 products over B T rows
        "amid_bert_strip_qkv_fwd_pro_f32": ("mfma", 3 * gl), "amid_bert_strip_qkv_fwd_f32": ("mfma", 3 * gl),
        "amid_bert_strip_oproj_ffn_fwd_f32#0": ("mfma", 12 * gl),      # block 0's out-projection + feed-forward (1 + 4 + 4) and block 1's q / k / v
        "amid_bert_strip_oproj_ffn_fwd_f32#1": ("mfma", 9 * gl),
        "amid_bert_strip_ffn_bwd_f32": ("mfma", 9 * gl),
        "amid_bert_strip_qkv_bwd_f32#0": ("mfma", 12 * gl),            # block 1's q / k / v backward + block 0's feed-forward / out-projection backward
        "amid_bert_strip_qkv_bwd_f32#1": ("mfma", 3 * gl),
        # ... with every product as six bf16 piece pairs (round 5, Bert4recEngine.STRIP_P3): the same algorithmic FLOP against the bf16 peak / 6
        "amid_bert_strip_qkv_fwd_pro_p3_f32": ("mfma16x6", 3 * gl),
        "amid_bert_strip_oproj_ffn_fwd_p3_f32#0": ("mfma16x6", 12 * gl), "amid_bert_strip_oproj_ffn_fwd_p3_f32#1": ("mfma16x6", 9 * gl),
        "amid_bert_strip_ffn_bwd_p3_f32": ("mfma16x6", 9 * gl),
        "amid_bert_strip_qkv_bwd_p3_f32#0": ("mfma16x6", 12 * gl), "amid_bert_strip_qkv_bwd_p3_f32#1": ("mfma16x6", 3 * gl),
        "amid_attn_bert_fwd_live_f32": ("mfma", 4.0 * T * T * (D // 4) * Bw * 4),
        "amid_attn_bert_bwd_live_f32": ("mfma", 10.0 * T * T * (D // 4) * Bw * 4),
        "amid_bert_qkv_fwd_f32": ("mfma", 3 * gemm),
        "amid_bert_oproj_fwd_f32": ("mfma", gemm),
        "amid_bert_ffn1_fwd_f32": ("mfma", 4 * gemm),       # [M, 128] x [128, 512]
        "amid_bert_ffn2_fwd_f32": ("mfma", 4 * gemm),       # [M, 512] x [512, 128]
        "amid_bert_ffn2_bwd_f32": ("mfma", 4 * gemm),       # dz W_2 -> d pre
        "amid_bert_ffn1_bwd_f32": ("mfma", 5 * gemm),       # d pre W_1 -> d y2, and dt W_o -> d o
        "amid_bert_qkv_bwd_f32": ("mfma", 3 * gemm),
        "amid_bert_wgrad_f32": ("mfma", 12 * gemm),         # one layer per launch: q, k, v, o and the 4 + 4 tiles of w_1, w_2
        "amid_bert_wgrad_rows_f32": ("mfma", 6 * gemm),      # the live sequences only (see amid_sas_wgrad_rows_f32)
        "amid_bert_wgrad_mode_f32": (WGRAD_KIND, 6 * gemm),  # the engine's call: live sequences, products on bf16 pieces by default
        # matrix-core kernels (H hd = D either way: SASRec 8 x 16, BERT4Rec 4 x 32); unpadded T x T products
        "amid_attn_fwd_f32": ("mfma", 4.0 * T * T * hd * 2 * Bw * H),
        "amid_attn_bwd_f32": ("mfma", 10.0 * T * T * hd * 2 * Bw * H),
        # K1: index + row read + row write (+ the feature-mask bytes); K3: rows + positions read, unique rows + ids written;
        # K4: table row, m, v read and written for every unique row (+ its `last` stamp), grad row read
        "amid_embed_fwd_f32": ("hbm", n_idx * (4 + 2 * D * 4) + M2 * (D // 4)),
        "amid_embgrad_segreduce_f32": ("hbm", n_idx * (D * 4 + 8) + U * (D * 4 + 8)),
        # the fused train step's sparse side runs on the live sequences' positions + the items (half the index list)
        "amid_embgrad_segreduce_live": ("hbm", (Bw * T + Bw * (1 + NEG)) * (D * 4 + 8) + U * (D * 4 + 8)),
        "amid_lazy_adam_catchup_live_f32": ("hbm", (Bw * T + Bw * (1 + NEG)) * 12 + U * (D * 4 * 6 + 8)),
        "amid_lazy_adam_catchup_positions_f32": ("hbm", n_idx * 4 + U * (D * 4 * 6 + 8)),
        "amid_optimizer_step_f32": ("hbm", U * (D * 4 * 7 + 8)),
        # round 5, the folded step: packing + catch-up (+ sort phase 1) over the compact list in one launch; the last strip launch with the
        # embedding backward on the strip; the weight gradients carrying the sort's last phase; the optimizer finishing the segment reduce
        # the folded step's forward (row statistics instead of qn / y: seven saved tensors per layer) and its weight gradients
        "amid_sas_seq_fwd_split_lnstat_f32": ("mfma16x6", 12 * gl + 2 * 4.0 * T * T * hd * Bw * H),
        # ... with the step's head on the tail of its workgroups (the head's own arithmetic -- 2 x 0.13 MFLOP a sample -- is not counted)
        "amid_sas_seq_fwd_split_lnstat_head_f32": ("mfma16x6", 12 * gl + 2 * 4.0 * T * T * hd * Bw * H),
        # ... and with the gather K1 as its prologue (round 6): + the rows it gathers and stores for the backward (not counted: the launch is matrix-bound)
        "amid_sas_seq_fwd_gather_head_f32": ("mfma16x6", 12 * gl + 2 * 4.0 * T * T * hd * Bw * H),
        "amid_sas_seq_fwd_gather_f32": ("mfma16x6", 12 * gl + 2 * 4.0 * T * T * hd * Bw * H),
        "amid_sas_seq_fwd_gather_head_p1_f32": ("mfma16", 12 * gl + 2 * 4.0 * T * T * hd * Bw * H),
        "amid_sas_seq_fwd_gather_p1_f32": ("mfma16", 12 * gl + 2 * 4.0 * T * T * hd * Bw * H),
        "amid_step_head_w16_f32": ("hbm", (Bw * T + Bw * (1 + NEG)) * 28 + (2 * Bw * T + 2 * Bw * (1 + NEG)) * 20 + U * (D * 4 * 6 + 8) + 48 * D * D * 10),
        "amid_sas_wgrad_rows_sort_ln_f32": (WGRAD_KIND, 6 * gemm),
        "amid_sas_strip_qkv_bwd_sort_scorer_f32": ("mfma", 6 * gl),     # layer 1's q / k / v backward + layer 0's feed-forward backward (+ riders)
        "amid_step_head_f32": ("hbm", (Bw * T + Bw * (1 + NEG)) * 28 + (2 * Bw * T + 2 * Bw * (1 + NEG)) * 20 + U * (D * 4 * 6 + 8)),
        "amid_sas_strip_qkv_bwd_emb_f32": ("mfma", 3 * gl),
        "amid_sas_wgrad_rows_sort_f32": (WGRAD_KIND, 6 * gemm),
        "amid_optimizer_step_spans_f32": ("hbm", U * (D * 4 * 7 + 8)),
    }


KERNEL_SYMBOL = {          # C-ABI entry -> substring of the device kernel's name in rocprofv3 output
    "amid_sas_qkv_fwd_f32": "sas_qkv_fwd_kernel", "amid_sas_oproj_fwd_f32": "sas_oproj_fwd_kernel", "amid_sas_ffn_fwd_f32": "sas_ffn_fwd_kernel",
    "amid_sas_ffn_bwd_f32": "sas_ffn_bwd_kernel", "amid_sas_qkv_bwd_f32": "sas_qkv_bwd_kernel", "amid_sas_wgrad_f32": "sas_wgrad_kernel",
    "amid_sas_wgrad_rows_f32": ("sas_wgrad_split_kernel", "sas_wgrad_kernel", "sas_wgrad16_kernel"), "amid_bert_wgrad_mode_f32": ("bert_wgrad_split_kernel", "bert_wgrad_kernel"), "amid_attn_bwd_rows_f32": "attn_bwd_mfma_kernel",
    "amid_sas_ffn_bwd_rows_f32": "sas_ffn_bwd_kernel", "amid_sas_qkv_bwd_rows_f32": "sas_qkv_bwd_kernel",
    "amid_sas_qkv_ffn_bwd_rows_f32": "sas_qkv_ffn_bwd_kernel",
    "amid_sas_qkv_ffn_bwd_f32": "sas_qkv_ffn_bwd_kernel", "amid_sas_oproj_ffn_qkv_fwd_f32": "sas_oproj_ffn_qkv_fwd_kernel",
    "amid_sas_oproj_ffn_fwd_f32": "sas_oproj_ffn_fwd_kernel",
    "amid_attn_fwd_f32": "attn_fwd_mfma_kernel", "amid_attn_bwd_f32": "attn_bwd_mfma_kernel", "amid_embed_fwd_f32": "embed_fwd_kernel",
    "amid_sas_seq_fwd_f32": ("seqn_fwd_kernel", "seq_fwd_kernel"), "amid_sas_seq_fwd_bf16w_f32": ("seqn_fwd_kernel", "seq_fwd_kernel"), "amid_sas_seq_fwd_split_f32": ("seqn_fwd_px_kernel", "seqn_fwd_kernel"),
    "amid_embed_fwd_w16_f32": "embed_fwd_kernel", "amid_embed_fwd_tiles_f32": "embed_fwd_kernel",
    "amid_sas_seq_bwd_f32": ("seqn_bwd_kernel", "seq_bwd_kernel"), "amid_sas_strip_qkv_fwd_f32": "strip_qkv_fwd_kernel", "amid_sas_strip_oproj_ffn_fwd_f32#0": "strip_oproj_ffn_fwd_kernelILi128ELb1",
    "amid_sas_strip_oproj_ffn_fwd_f32#1": "strip_oproj_ffn_fwd_kernelILi128ELb0", "amid_sas_strip_ffn_bwd_f32": "strip_ffn_bwd_kernel",
    "amid_sas_strip_qkv_bwd_f32#0": "strip_qkv_bwd_kernelILi128ELb1", "amid_sas_strip_qkv_bwd_f32#1": "strip_qkv_bwd_kernelILi128ELb0",
    "amid_attn_fwd_live_f32": "attn_fwd_mfma_kernel", "amid_attn_bwd_live_f32": "attn_bwd_mfma_kernel", "amid_embed_fwd_live_f32": "embed_fwd_kernel", "amid_embed_fwd_live_compact_f32": "embed_fwd_kernel", "amid_embed_fwd_replay_f32": "embed_fwd_kernel",
    "amid_embgrad_segreduce_f32": "segreduce_chunks_kernel", "amid_sas_wgrad_rows_sort_f32": "sas_wgrad_split_kernel",
    "amid_sas_strip_qkv_bwd_emb_f32": "strip_qkv_bwd_kernelILi128ELb0", "amid_step_head_f32": "step_head_kernel",
    "amid_sas_seq_fwd_split_lnstat_f32": ("seqn_fwd_px_kernel",), "amid_sas_seq_fwd_split_lnstat_head_f32": ("seqn_fwd_px_head_kernel",), "amid_sas_wgrad_rows_sort_ln_f32": "sas_wgrad_split_kernel",
    "amid_sas_strip_qkv_bwd_sort_scorer_f32": "strip_qkv_bwd_kernelILi128ELb1", "amid_head_fwd_bwd_own_vec_f32": "head_fwd_bwd_kernel",
    "amid_grad_tail_live_f32": "grad_tail_live_kernel", "amid_grad_tail_nospans_f32": "grad_tail_kernel", "amid_optimizer_step_spans_f32": "optimizer_step_spans_kernel",
    "amid_grad_tail_opt_f32": "grad_tail_opt_kernel", "amid_step_head_w16_f32": "step_head_kernel",
    "amid_sas_seq_fwd_gather_head_f32": ("seqn_fwd_px_head_kernel",), "amid_sas_seq_fwd_gather_f32": ("seqn_fwd_px_kernel",),
    "amid_sas_seq_fwd_gather_head_p1_f32": ("seqn_fwd_px_head_kernel",), "amid_sas_seq_fwd_gather_p1_f32": ("seqn_fwd_px_kernel",),
    "amid_bert_strip_qkv_fwd_pro_p3_f32": "bert_strip_qkv_fwd_kernel", "amid_bert_strip_oproj_ffn_fwd_p3_f32#0": "bert_strip_oproj_ffn_fwd_kernelILb1",
    "amid_bert_strip_oproj_ffn_fwd_p3_f32#1": "bert_strip_oproj_ffn_fwd_kernelILb0", "amid_bert_strip_ffn_bwd_p3_f32": "bert_strip_ffn_bwd_kernel",
    "amid_bert_strip_qkv_bwd_p3_f32#0": "bert_strip_qkv_bwd_kernelILb1", "amid_bert_strip_qkv_bwd_p3_f32#1": "bert_strip_qkv_bwd_kernelILb0",
    "amid_bert_strip_qkv_fwd_pro_f32": "bert_strip_qkv_fwd_kernel", "amid_bert_strip_oproj_ffn_fwd_f32#0": "bert_strip_oproj_ffn_fwd_kernelILb1",
    "amid_bert_strip_oproj_ffn_fwd_f32#1": "bert_strip_oproj_ffn_fwd_kernelILb0", "amid_bert_strip_ffn_bwd_f32": "bert_strip_ffn_bwd_kernel",
    "amid_bert_strip_qkv_bwd_f32#0": "bert_strip_qkv_bwd_kernelILb1", "amid_bert_strip_qkv_bwd_f32#1": "bert_strip_qkv_bwd_kernelILb0",
    "amid_attn_bert_fwd_live_f32": "attn_fwd_bert_kernel", "amid_attn_bert_bwd_live_f32": "attn_bwd_bert_kernel",
}


def pmc_traffic(entry: str, tag: str):
    """HBM bytes per launch of one kernel REPLAYED from the committed PMC summary of THIS configuration
    (profiles/r*_<tag>_hbm_traffic.json, produced by profiles/summarize.py from two separate rocprofv3 --pmc passes of the same
    command; tag = workload_model_dtype); None when no summary of this configuration is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_hbm_traffic.json")))
    sym = KERNEL_SYMBOL.get(entry)
    if not files or sym is None:
        return None, None
    ks = json.load(open(files[-1]))["kernels"]
    for one in (sym if isinstance(sym, tuple) else (sym,)):          # (several builds of one entry point: the first that was profiled)
        for name, v in ks.items():
            if one in name:
                return v["hbm_bytes_per_launch"], os.path.basename(files[-1])
    return None, None


def gather_stress(device, n_steps=6):
    """BASELINE.json configs[4], S-uniform, one GPU: a few eager steps at batch 4096 over a 10 M x 128 fp32 table (5.12 GB, far beyond
    the 256 MiB Infinity Cache); returns {kernel: avg launch us, algorithmic GB/s, fraction of the 8 TB/s HBM peak} for the step's
    HBM-bound kernels (K1 gather, K3 segment reduce + partial sums, K4a lazy catch-up, K4b Adam), timed with HIP events."""
    from amid_amd._lib import KernelTimer, lib
    from amid_amd.engine import SasrecEngine
    wl = WORKLOADS["cfg5-uniform"]
    eng = SasrecEngine(wl["n_rows"], D, T, HID, device=device, lr=5e-4, seed=1234)
    init_params(eng, seed=0)
    pl = eng.plan(wl["B"], T, 1 + NEG, need_grad=True)
    gen = torch.Generator().manual_seed(7)
    packs = []
    for _ in range(3):
        b = synth_batch(gen, device, wl)
        packs.append(eng.pack_batch(pl, b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"], b["label"], b["domain_id"]))
    L = lib()
    for i in range(2 + n_steps):
        if i == 2:
            eng.sync()
            L.timer = KernelTimer()
        eng.load_packed(pl, packs[i % 3])
        eng.enqueue_train_step(pl)
        eng.sync()
    durs = L.timer.collect(L)
    L.timer = None
    work = algorithmic_work(wl["B"], int(pl.n_uniq.item()), T)
    if getattr(pl, "red_bytes", None):
        k3 = "amid_embgrad_segreduce_live" if getattr(pl, "compact", False) else "amid_embgrad_segreduce_f32"
        work["amid_grad_tail_f32"] = ("hbm", work[k3][1] + pl.red_bytes)
    role = {"amid_embed_fwd_live_f32": "K1 gather (live sequences)", "amid_embed_fwd_f32": "K1 gather",
            "amid_embed_fwd_live_compact_f32": "K1 gather (live sequences; writes the compact index list)",
            "amid_embed_fwd_replay_f32": "K1 gather (live sequences; lazy-Adam catch-up folded in: lagging rows replayed in registers)",
            "amid_embed_fwd_w16_f32": "K1 gather (live sequences; compact index list; + the step's 144 weight-image planes -- forward and backward strips -- by extra workgroups)",
            "amid_lazy_adam_catchup_live_f32": "K4a lazy-Adam catch-up (live sequences)",
            "amid_grad_tail_f32": "K3 segment reduce + dense partial sums", "amid_embgrad_segreduce_f32": "K3 segment reduce",
            "amid_lazy_adam_catchup_positions_f32": "K4a lazy-Adam catch-up", "amid_optimizer_step_f32": "K4b Adam (dense + unique rows)"}
    out = {"workload": wl["label"], "batch": wl["B"], "unique_rows": int(pl.n_uniq.item()), "kernels": {}}
    for name, v in durs.items():
        if name in role and name in work:
            us = 1e3 * sum(v) / len(v)
            gbps = work[name][1] / (us * 1e-6) / 1e9
            out["kernels"][name] = {"role": role[name], "avg_launch_us": round(us, 1), "achieved": round(gbps, 1), "unit": "GB/s",
                                    "peak": PEAK_HBM_GBPS, "frac": round(gbps / PEAK_HBM_GBPS, 4)}
    del eng, pl
    torch.cuda.empty_cache()
    return out


PREWARM_MS = 100.0


def steps_per_graph_for(requested: int, steps: int, graphable: bool) -> int:
    """Consecutive train steps per replayed hipGraph in the timed region: the CLI's train loop replays graphs of STEPS_PER_GRAPH = 4
    (amid_amd/train_sr.py), so the bench does too whenever the K timed steps are whole graphs -- the warm-up is rounded UP to whole
    graphs instead of shrinking the graph (the driver's --steps 20 --warmup 5 keeps 4).  Only a K that is not a multiple falls back
    to the largest divisor below."""
    spg = max(1, requested) if graphable else 1
    while spg > 1 and steps % spg:
        spg -= 1
    return spg


def usable_cpus() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


EVAL_NEG = 999          # run.sh:1 evaluates with --neg_nums 999: 1 000 candidates per row


def eval_batch(gen, device, Bw):
    """A test() batch (train_sr.py:31-128): synth_batch's sequences with EVAL_NEG negatives per row."""
    wl = dict(WORKLOADS["cfg2"], B=Bw)
    b = synth_batch(gen, "cpu", wl)
    b["neg_samples"] = torch.randint(1, wl["max_id"] + 1, (Bw, EVAL_NEG), generator=gen)
    b["label"] = torch.zeros(Bw, 1 + EVAL_NEG)
    b["label"][:, 0] = 1.0
    return {k: v.to(device) for k, v in b.items()}


def eval_throughput(eng, device, Bw, T, n_batches=64):
    """The evaluation loop's throughput (what test() does per batch, train_sr.py:31-128, at run.sh's 999 negatives), as the CLI's test()
    runs it since round 6 (SASRec.eval_ranks -> SasrecEngine.eval_epoch): the batches resident in HBM, per batch one device copy of the
    packed image, FOUR launches replayed as one graph -- index marshal + live list, K1 over the own-domain sequences, the inference forward
    over them (nothing saved), amid_eval_head_f32 (LN_last + mean, the scorer over the 1 000 candidates gathered in the launch, masked
    BCE, the positive's rank with and without fix_value) -- and one device copy of the 3 B result words.  test() reads only the own
    domain's logits (utils.py:21-40) and masks the other domain's loss terms (train_sr.py:63-64): nothing else is computed.
    `kernels`: each launch timed with HIP events on the engine's stream, priced against the roof that bounds it."""
    from amid_amd._lib import KernelTimer, lib
    gen = torch.Generator().manual_seed(4321)
    batches = [eval_batch(gen, device, Bw) for _ in range(4)]
    pl = eng.plan(Bw, T, 1 + EVAL_NEG, need_grad=False)
    if not eng.eval_fused_ok(pl):
        raise RuntimeError("the four-launch evaluation batch does not cover this engine / shape")
    packed4 = torch.stack([eng.pack_batch(pl, b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"], b["label"], b["domain_id"]) for b in batches])
    packed = packed4[torch.arange(n_batches, device=device) % 4].contiguous()
    torch.cuda.synchronize(device)
    eng.flush_table()
    eng.sync()
    out = eng.eval_epoch(pl, packed[:8], 1e-7)            # warm-up + graph capture
    eng.sync()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    out = eng.eval_epoch(pl, packed, 1e-7)
    eng.sync()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    eng.check_index_error(pl)
    rank = out[:, :Bw]
    # per-launch durations (eager, HIP events on the engine's stream) and what bounds each launch
    L = lib()
    L.timer = KernelTimer()
    n_prof = 10
    try:
        for i in range(n_prof):
            eng.load_packed(pl, packed[i])
            eng.enqueue_eval(pl, 1e-7, build_images=False)      # (as the replayed graph: the weight images were built once, by eval_epoch)
            eng.sync()
        durs = L.timer.collect(L)
    finally:
        L.timer = None
    NI = 1 + EVAL_NEG
    gl = 2.0 * Bw * T * D * D
    work = {
        # the candidates' rows + their ids, the own sequences' last-layer rows; scores / ranks / loss stay in the workgroup
        "amid_eval_head_f32": ("hbm", Bw * NI * (D * 4 + 4) + Bw * T * D * 4 + Bw * NI * 4),
        "amid_sas_seq_fwd_split_infer_f32": ("mfma16x6", 12 * gl + 2 * 4.0 * T * T * (D // 8) * Bw * 8),
        "amid_sas_seq_fwd_gather_infer_f32": ("mfma16x6", 12 * gl + 2 * 4.0 * T * T * (D // 8) * Bw * 8),
        "amid_embed_fwd_live_f32": ("hbm", Bw * T * (8 + 2 * D * 4) + Bw * T * (D // 4)),
        "amid_pack_indices_live": ("hbm", (2 * Bw * T + Bw * NI) * 12),
    }
    kernels = {}
    for name, v in durs.items():
        us = 1e3 * sum(v) / len(v)
        ent = {"avg_launch_us": round(us, 2)}
        if name in work:
            kind, amount = work[name]
            if kind == "hbm":
                ent.update(bound="hbm", achieved=round(amount / (us * 1e-6) / 1e9, 1), peak=PEAK_HBM_GBPS, unit="GB/s")
            else:
                ent.update(bound="mfma", achieved=round(amount / (us * 1e-6) / 1e12, 2), peak=round(PEAK_BF16_MFMA_TFLOPS / 6, 1), unit="TFLOP/s",
                           operands="fp32 as three bf16 pieces, 6 piece pairs")
            ent["frac"] = round(ent["achieved"] / ent["peak"], 4)
        kernels[name] = ent
    head = kernels.get("amid_eval_head_f32", {})
    if head:       # the scorer's chains: NI hid D fma per sample on the vector pipe, beside the gather
        head["vector_fma_tflops"] = round(2.0 * Bw * NI * HID * D / (head["avg_launch_us"] * 1e-6) / 1e12, 2)
    return {"metric": "eval samples/sec (test(): own-domain forward + 1 + neg candidates per row + loss + the positive's rank, on the device)",
            "value": round(Bw * n_batches / dt, 1), "unit": "samples/s", "batches_per_s": round(n_batches / dt, 2),
            "ms_per_batch": round(1e3 * dt / n_batches, 4), "batch": Bw, "seq_len": T, "neg_nums": EVAL_NEG, "batches_timed": n_batches,
            "mean_rank_last_batch": round(float(rank[-1].float().mean().item()), 2),
            "launches_per_batch": sum(len(v) for v in durs.values()) // n_prof, "sum_kernel_us": round(sum(k["avg_launch_us"] for k in kernels.values()), 2),
            "kernels": kernels,
            "what": "per batch: one device copy of the packed batch, one replayed graph of three launches (index marshal + live list, inference "
                    "forward over the B own-domain sequences with their rows gathered in its prologue, eval head: gather of the 1 000 candidates per row inside the "
                    "scorer + masked BCE + ranks), one device copy of 3 B result words; synthetic cloth_sport-shaped batches"}


def cpu_eval_baseline(P, budget_s=8.0):
    """The oracle's eval forward + the reference's rank arithmetic (get_sample_scores) on this host's cores, same batch shape."""
    from oracle import amid_oracle as orc
    gen = torch.Generator().manual_seed(4321)
    ts = []
    t_begin = time.perf_counter()
    with torch.no_grad():
        while len(ts) < 6 and (len(ts) < 2 or time.perf_counter() - t_begin < budget_s):
            b = eval_batch(gen, "cpu", B)
            t0 = time.perf_counter()
            p1, p2 = orc.sasrec_forward(P, b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"])
            pred = torch.where(b["domain_id"][:, None] == 0, p1.reshape(B, -1), p2.reshape(B, -1)).numpy().copy()
            pred[:, 0] -= 1e-7
            orc.get_sample_scores(pred)
            ts.append(time.perf_counter() - t0)
    timed = sorted(ts[1:])
    med = timed[len(timed) // 2]
    return {"value": round(B / med, 1), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port", "ms_per_batch": round(med * 1e3, 1),
            "sample": f"{len(timed)} timed batches (+1 warm-up, <= {budget_s:.0f} s) of oracle sasrec_forward + get_sample_scores at batch {B} x seq {T} "
                      f"x {1 + EVAL_NEG} candidates, median"}


def cpu_baseline(budget_s=25.0):
    """The oracle (CPU restatement of the reference path: dense embedding grads + dense Adam over the
    894 820-row table) timed on this host's cores on a bounded sample of the same workload."""
    from oracle import amid_oracle as orc
    threads = min(usable_cpus(), 32)            # torch intra-op scaling of this path flattens well before 32 threads
    torch.set_num_threads(threads)
    P = orc.random_params(orc.sasrec_param_shapes(N_ROWS, D, T, HID), seed=0)
    opt = orc.DenseAdam(P, lr=5e-4)
    gen = torch.Generator().manual_seed(99)
    shapes = orc.philox_masks_sasrec(1, T, D, 0, 1)
    masks = {k: (torch.rand((B,) + tuple(v.shape[1:])) >= 0.5).float() for k, v in shapes.items()}
    ts = []
    t_begin = time.perf_counter()
    while len(ts) < 12 and (len(ts) < 3 or time.perf_counter() - t_begin < budget_s):
        b = synth_batch(gen, "cpu")
        t0 = time.perf_counter()
        orc.train_step("sasrec", P, opt, b, masks)
        ts.append(time.perf_counter() - t0)
    timed = sorted(ts[2:])                             # two warm-up steps (allocator, thread pool, first-touch of 1.4 GB of state)
    med = timed[len(timed) // 2]
    try:
        cpu_eval = cpu_eval_baseline(P)
    except Exception as e:                                  # a side measurement must never cost the headline's baseline
        cpu_eval = {"error": f"{type(e).__name__}: {e}"}
    return {"value": round(B / med, 2), "unit": "samples/s", "cores": threads, "kind": "port", "eval": cpu_eval,
            "ms_per_step": round(med * 1e3, 1),
            "sample": f"{len(timed)} timed steps (+2 warm-up, <= {budget_s:.0f} s) of oracle/amid_oracle.py train_step at the same "
                      f"batch {B} x seq {T} x dim {D}, dense Adam over the {N_ROWS}-row table, median; host reports "
                      f"{os.cpu_count()} logical CPUs, {usable_cpus()} usable"}


def self_launch(n: int) -> int:
    """python bench.py --gpus N without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <same args>` as a
    child of this process (which has not touched the GPU: importing torch does not initialise HIP) and relay its output."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    rc = 1
    for attempt in range(3):        # the port is picked by bind / close / reuse: another process may take it in between -- a launcher that
        sock = socket.socket()      # dies within seconds is retried on a fresh port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        t0 = time.perf_counter()
        rc = subprocess.run(cmd, env=env).returncode
        if rc == 0 or time.perf_counter() - t0 > 15.0:
            break
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying a hipGraph")
    ap.add_argument("--input", default="pool", choices=("pool", "copy"),
                    help="pool: the packed batches stay in HBM and the step picks its batch by the device step counter; "
                         "copy: one device-to-device copy into the static input buffer per step")
    ap.add_argument("--model", default="sasrec", choices=("sasrec", "bert4rec"),
                    help="sasrec (default, the headline); bert4rec: the same workload through the BERT4Rec encoder (reference a7), "
                         "reported with config.model, never the headline line")
    ap.add_argument("--dtype", default="f32", choices=("f32", "bf16"),
                    help="f32 (default, the headline): exact fp32 matrix products; bf16: bf16 MFMA operands with fp32 accumulation "
                         "(BASELINE.json configs[2]) -- reported with dtype bf16, never the headline line")
    ap.add_argument("--data", default="real", choices=("real", "synthetic"),
                    help="real (default where the workload has a fixture: cfg2, cfg3, cfg4): the reference's training CSV as tokenised by "
                         "its own dataset class (tests/golden/tok_*.npz); synthetic: batches drawn to the file's statistics")
    ap.add_argument("--steps-per-graph", type=int, default=4,
                    help="consecutive train steps captured into one replayed hipGraph (single GPU, pool input; default 4 = what the "
                         "CLI's train loop replays, amid_amd/train_sr.py STEPS_PER_GRAPH)")
    ap.add_argument("--no-stress", action="store_true", help="skip the cfg5 gather / scatter stress appended to the headline line")
    ap.add_argument("--no-fused-tail", action="store_true",
                    help="A/B: round 4's fifteen-launch step (SasrecEngine.FUSED_TAIL = False) instead of the folded twelve-launch one")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B: set a path switch of the engine class before it is built (SasrecEngine.NAME, e.g. SEQ_BACKWARD=0, "
                         "FWD_SPLIT=False, SEQ_FWD_VARIANT=42); the line then carries config.switches and is not the headline")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS),
                    help="cfg2 = the headline configuration (default); cfg5-* = synthetic gather / scatter stress (SURVEY.md 8(d))")
    ap.add_argument("--dense-exchange", default="gather", choices=("gather", "allreduce"),
                    help="N > 1: how the flat dense gradient crosses the ranks -- gather: behind the sparse rows in the step's one "
                         "all-gather; allreduce: its own RCCL all-reduce next to it (amid_amd/engine.py DENSE_EXCHANGE)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))          # rank processes are children; this one never initialises the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match the launcher's WORLD_SIZE {world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # AMID_BENCH_BACKEND=gloo (+ host-staged payloads, ranks folded onto the visible GPUs) only exists to smoke-test the
    # N > 1 control flow on a single-GPU box; real runs use RCCL ("nccl") with one GPU per rank.
    backend = os.environ.get("AMID_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count()) if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from amid_amd._lib import KernelTimer, lib
    from amid_amd.dist import SparseDenseExchange
    from amid_amd.engine import SasrecEngine

    wl = WORKLOADS[args.workload]
    Bw = wl["B"]
    T = wl.get("T", globals()["T"])
    globals()["D"] = wl.get("D", D)          # (cfg1: emb_dim 64; every formula below reads the module's D)
    switches = {}
    for kv in args.set:                       # A/B only: class attributes of the engine (and of its plan for WGRAD_SPLITS)
        name, _, val = kv.partition("=")
        from amid_amd.plan import SasrecPlan
        holder = SasrecPlan if name == "WGRAD_SPLITS" else SasrecEngine
        if not hasattr(holder, name) and args.model == "bert4rec":
            from amid_amd.engine_bert import Bert4recEngine as holder        # (its own switches: STRIP_P3)
        if not hasattr(holder, name):
            raise SystemExit(f"--set {kv}: {holder.__name__} has no switch {name}")
        cur = getattr(holder, name)
        new = (val not in ("0", "False", "false")) if isinstance(cur, bool) else type(cur)(val)
        setattr(holder, name, new)
        switches[name] = new
    if args.no_fused_tail:
        SasrecEngine.FUSED_TAIL = False
        switches["FUSED_TAIL"] = False
    if args.model == "bert4rec":
        from amid_amd.engine_bert import Bert4recEngine
        eng = Bert4recEngine(wl["n_rows"], D, T, HID, device=device, lr=5e-4, seed=1234)
    else:
        eng = SasrecEngine(wl["n_rows"], D, T, HID, device=device, lr=5e-4, seed=1234, compute=args.dtype)
    init_params(eng, seed=0)                  # identical replicas on every rank
    pl = eng.plan(Bw, T, 1 + NEG, need_grad=True)
    gen = torch.Generator().manual_seed(1000 + rank)          # each rank draws its own shard of the global batch
    use_real = (args.data == "real" and args.workload in REAL
                and all(os.path.exists(os.path.join(FIXTURES, f"tok_{n}.npz")) for n in REAL[args.workload]))
    loader, data_desc = None, "synthetic (batches drawn to the statistics of the file, SURVEY.md section 8(d))"
    pool, uniq_counts = [], []
    if use_real:                              # one epoch of the REAL file, sharded by rank, packed once: [n_batches, words] in HBM
        loader, data_desc = real_batches(args.workload, Bw, device, rank, world)
        ep = loader.epoch_tensors()
        packed = eng.pack_epoch(pl, ep["i_node"], ep["neg_samples"], ep["seq_d1"], ep["seq_d2"], ep["label"], ep["domain_id"])
        pool = list(packed)
        idx = torch.cat([ep[k].reshape(len(pool), -1) for k in ("i_node", "neg_samples", "seq_d1", "seq_d2")], 1)
        srt = torch.sort(idx, dim=1).values
        uniq_counts = ((srt[:, 1:] != srt[:, :-1]).sum(1) + 1).tolist()
        data_desc = "real: " + data_desc
    n_pool = len(pool) if use_real else (60 if args.workload in ("cfg2", "cfg3", "cfg4") else 16)
    for _ in range(0 if use_real else n_pool):          # batches are packed in the engine's input layout: one device copy per step
        b = synth_batch(gen, device, wl)
        pool.append(eng.pack_batch(pl, b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"], b["label"], b["domain_id"]))
        uniq_counts.append(int(torch.unique(torch.cat([b[k].reshape(-1) for k in ("i_node", "neg_samples", "seq_d1", "seq_d2")])).numel()))
    torch.cuda.synchronize()
    umax_pool = uniq_counts
    if world > 1:
        # data-pipeline work, outside the timed region: every batch's unique-row count is known when it is packed, and the
        # world's per-step maximum is reduced once here, so the sparse exchange needs no device -> host sync in the step
        cnt = torch.tensor(uniq_counts, dtype=torch.int64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
        umax_pool = [int(v) for v in cnt.tolist()]
        # one padded length for the whole run (the pool's largest count, rounded up): the graph pair of the exchange is captured once
        umax_pool = [(max(umax_pool) + 255) // 256 * 256] * len(umax_pool)

    # inputs resident in HBM (the bench contract): the epoch's packed batches form one [n_pool, in_words] tensor and the step's
    # first kernel picks its batch by the device step counter -- no per-step input copy (--input copy restores the D2D copy)
    use_pool = args.input == "pool"
    if use_pool:
        eng.set_input_pool(pl, torch.stack(pool))

    def load(i):
        if not use_pool:
            eng.load_packed(pl, pool[i % n_pool])

    use_graph = not args.no_graph
    spg = steps_per_graph_for(args.steps_per_graph, args.steps, use_graph and use_pool and world == 1)
    exchange = SparseDenseExchange(eng.merge_backend(world * pl.shape.n_idx), host_staging=(backend != "nccl")) if world > 1 else None
    load(0)
    if use_graph:
        if world == 1:
            eng.capture_train_step(pl)
            if spg > 1:
                eng.capture_train_steps(pl, spg)
        else:
            eng.capture_local_grads(pl)

    def step(i):
        load(i)
        if world == 1:
            if use_graph and spg > 1:
                if i % spg == 0:
                    eng.replay_train_steps(pl, spg)
            elif use_graph:
                eng.replay_train_step(pl)
            else:
                eng.enqueue_train_step(pl)
        else:
            eng.train_step_dp(pl, exchange, use_graph=use_graph, umax=umax_pool[i % n_pool], dense=args.dense_exchange)

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    done = 0                                   # steps enqueued so far (a graph of spg steps counts spg): `step` is called with a running index

    def run(n):
        nonlocal done
        for _ in range(n):
            step(done)
            done += 1

    def window(n):
        """n steps between two barriers; the MAX over the ranks, seconds."""
        barrier()
        t0 = time.perf_counter()
        run(n)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    # W untimed warm-up steps, in whole graphs: ceil(W / spg) replays (never fewer than W steps)
    run(-(-args.warmup // spg) * spg)
    barrier()
    # pre-warm, not declared in `warmup`: a fresh lease runs its first milliseconds at idle clocks and cold caches, and the driver's
    # --warmup 5 is 2 ms of work; replay whole graphs until PREWARM_MS of wall time have gone by (the same on every rank: the
    # count of replays is agreed on from rank 0's clock), reported as prewarm_ms / prewarm_steps
    prewarm_steps, t_pw = 0, time.perf_counter()
    while True:
        run(spg * 8)
        prewarm_steps += spg * 8
        barrier()
        go_on = (time.perf_counter() - t_pw) * 1e3 < PREWARM_MS
        if world > 1:
            flag = torch.tensor([int(go_on)], dtype=torch.int64, device=device if backend == "nccl" else "cpu")
            dist.broadcast(flag, src=0)
            go_on = bool(int(flag.item()))
        if not go_on:
            break
    prewarm_ms = (time.perf_counter() - t_pw) * 1e3
    ex0 = dict(exchange.stats) if exchange is not None else None
    dt = window(args.steps)                    # THE timed region: exactly K steps between two barriers
    ex1 = dict(exchange.stats) if exchange is not None else None
    # four more identical windows behind the headline one (the headline stays window 0): the spread of one 8-ms window on this box
    windows = [dt] + [window(args.steps) for _ in range(4)]
    loss_last = float(pl.loss.item())

    # ---- per-kernel durations, measured live with HIP events on the engine's stream (eager launches) ----
    roof, kernels = None, {}
    if rank == 0:
        L = lib()
        n_prof = 20
        L.timer = KernelTimer()
        for i in range(n_prof):
            load(i)
            if world == 1:
                eng.enqueue_train_step(pl)
            else:
                eng.enqueue_local_grads(pl)
                eng.enqueue_optimizer(pl)
            eng.sync()
        durs = L.timer.collect(L)
        L.timer = None
        for name in ("amid_sas_strip_oproj_ffn_fwd_f32", "amid_sas_strip_qkv_bwd_f32", "amid_bert_strip_oproj_ffn_fwd_f32",
                     "amid_bert_strip_qkv_bwd_f32", "amid_bert_strip_oproj_ffn_fwd_p3_f32", "amid_bert_strip_qkv_bwd_p3_f32"):      # two different launches per step under one entry
            v = durs.pop(name, None)
            if v is not None and len(v) == 2 * n_prof:
                durs[name + "#0"], durs[name + "#1"] = v[0::2], v[1::2]
            elif v is not None:
                durs[name] = v
        globals()["WGRAD_KIND"] = wgrad_kind_of(eng, args.model)
        work = algorithmic_work(Bw, int(pl.n_uniq.item()), T)
        red_bytes = (getattr(pl, "red_bytes_s", None) if getattr(pl, "seq_bwd_used", False) else None) or getattr(pl, "red_bytes_v", None) or getattr(pl, "red_bytes", None)
        if red_bytes:         # K3 (the segment reduce of the row gradients) + the reduce of every dense partial sum, one launch
            k3 = "amid_embgrad_segreduce_live" if getattr(pl, "compact", False) else "amid_embgrad_segreduce_f32"
            work["amid_grad_tail_f32"] = ("hbm", work[k3][1] + red_bytes)
            work["amid_grad_tail_nospans_f32"] = work["amid_grad_tail_f32"]      # (its second phase rides in the optimizer launch)
            if getattr(pl, "tail2", False):      # the folded step's tail: compact list, the position rows summed from the rows, no second phase
                work["amid_grad_tail_live_f32"] = ("hbm", work["amid_embgrad_segreduce_live"][1] + getattr(pl, "red_bytes_t", red_bytes)
                                                   + Bw * T * D * 4 + 2 * T * D * 4)
                # ... with the optimizer folded in (round 6): + the rows' and the dense parameters' Adam traffic (p, m, v read and written)
                work["amid_grad_tail_opt_f32"] = ("hbm", work["amid_grad_tail_live_f32"][1] + work["amid_optimizer_step_spans_f32"][1]
                                                  + eng.dense.numel * 4 * 6)
        total_ms = 0.0
        for name, v in durs.items():
            name = name[:-4] if name.endswith(("_rt3", "_rt4", "_rt5")) else name        # the 48- / 64- / 80-row builds of the row-tile kernels
            per_step = sum(v) / n_prof
            total_ms += per_step
            calls = len(v) // n_prof
            ent = {"ms_per_step": round(per_step, 4), "launches_per_step": calls, "avg_launch_us": round(1e3 * per_step / max(calls, 1), 2)}
            if name in work:
                kind, amount = work[name]
                avg_s = per_step / max(calls, 1) * 1e-3
                if kind == "hbm":
                    ent.update(bound="hbm", achieved=round(amount / avg_s / 1e9, 1), peak=PEAK_HBM_GBPS, unit="GB/s")
                elif kind == "mfma16":
                    ent.update(bound="mfma", achieved=round(amount / avg_s / 1e12, 2), peak=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s", operands="bf16")
                elif kind.startswith("mfma16x"):    # an fp32 product as six (nine) bf16 piece products: the bf16 pipe executes 6 (9) x the algorithmic FLOP
                    n = int(kind[7:])
                    ent.update(bound="mfma", achieved=round(amount / avg_s / 1e12, 2), peak=round(PEAK_BF16_MFMA_TFLOPS / n, 1), unit="TFLOP/s",
                               operands=f"fp32 as three bf16 pieces, {n} piece pairs (peak = the bf16 matrix peak / {n})",
                               frac_of_fp32_mfma_peak=round(amount / avg_s / 1e12 / PEAK_F32_MFMA_TFLOPS, 4))
                else:
                    ent.update(bound="mfma", achieved=round(amount / avg_s / 1e12, 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s")
                ent["frac"] = round(ent["achieved"] / ent["peak"], 4)
            kernels[name] = ent
        dom = max((k for k in kernels if k in work), key=lambda k: kernels[k]["ms_per_step"])      # most time per step
        # The dominant kernel once more with the queue kept full, as in the timed region: the pass above synchronises after every step
        # (it brackets EVERY launch with events), so each step starts on an idle device; here only the dominant kernel's launches carry
        # events and the steps are enqueued back to back.  This average prices the roofline; the per-step-synchronised one stays beside it.
        dom_entry = [n for n in durs if (n[:-4] if n.endswith(("_rt3", "_rt4", "_rt5")) else n) == dom or n.split("#")[0] == dom]
        hot_us = None
        if use_pool and dom_entry:
            L.timer = KernelTimer(only={n.split("#")[0] for n in dom_entry})
            for i in range(n_prof):
                if world == 1:
                    eng.enqueue_train_step(pl)
                else:
                    eng.enqueue_local_grads(pl)
                    eng.enqueue_optimizer(pl)
            eng.sync()
            hot = L.timer.collect(L)
            L.timer = None
            hv = [x for v in hot.values() for x in v]
            if hv:
                hot_us = 1e3 * sum(hv) / len(hv)
        roof = {"kernel": dom, "bound": kernels[dom]["bound"], "achieved": kernels[dom]["achieved"], "peak": kernels[dom]["peak"],
                "unit": kernels[dom]["unit"], "frac": kernels[dom]["frac"], "traffic": None,
                "avg_launch_us": kernels[dom]["avg_launch_us"], "sum_kernel_ms_per_step": round(total_ms, 4)}
        if hot_us is not None:
            amount = work[dom][1]
            scale = 1e9 if roof["bound"] == "hbm" else 1e12
            roof["avg_launch_us_synced_steps"] = roof["avg_launch_us"]
            roof["avg_launch_us"] = round(hot_us, 2)
            roof["achieved"] = round(amount / (hot_us * 1e-6) / scale, 1 if roof["bound"] == "hbm" else 2)
            roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
            if "frac_of_fp32_mfma_peak" in kernels[dom]:       # (the fp32 products run on the bf16 pipe: what the fp32 pipe's peak would make of it)
                roof["frac_of_fp32_mfma_peak"] = round(roof["achieved"] / PEAK_F32_MFMA_TFLOPS, 4)
                roof["operands"] = kernels[dom]["operands"]
            roof["timing"] = "HIP events around this kernel's launches only, 20 steps enqueued back to back (the queue stays full, as in the timed region)"
        if dom in ("amid_sas_seq_fwd_split_lnstat_head_f32", "amid_sas_seq_fwd_gather_head_f32", "amid_sas_seq_fwd_gather_head_p1_f32") and use_pool and world == 1:
            # like for like with the earlier rounds' lines: the encoder forward as its own launch (the head back in a launch of its own),
            # timed the same way -- HIP events around that kernel's launches only, steps enqueued back to back -- behind the timed region
            alone = {"amid_sas_seq_fwd_gather_head_f32": "amid_sas_seq_fwd_gather_f32", "amid_sas_seq_fwd_gather_head_p1_f32": "amid_sas_seq_fwd_gather_p1_f32"}.get(dom, "amid_sas_seq_fwd_split_lnstat_f32")
            eng.HEAD_ON_FWD = False
            try:
                L.timer = KernelTimer(only={alone})
                for i in range(n_prof):
                    eng.enqueue_train_step(pl)
                eng.sync()
                hv = [x for v in L.timer.collect(L).values() for x in v]
            finally:
                L.timer = None
                eng.HEAD_ON_FWD = True
            if hv:
                us = 1e3 * sum(hv) / len(hv)
                ach = round(work[alone][1] / (us * 1e-6) / 1e12, 2)
                roof["encoder_as_its_own_launch"] = {"kernel": alone, "avg_launch_us": round(us, 2), "achieved": ach, "unit": "TFLOP/s",
                                                     "peak": roof["peak"], "frac": round(ach / roof["peak"], 4),
                                                     "what": "the same forward without the head on its tail (SasrecEngine.HEAD_ON_FWD = False), same timing"}
        if dom in ("amid_sas_seq_fwd_split_lnstat_head_f32", "amid_sas_seq_fwd_gather_head_f32", "amid_sas_seq_fwd_gather_head_p1_f32"):
            roof["note"] = ("this launch also runs the train step's head on the tail of its workgroups (LN_last + mean, scorer, loss and their "
                            "backward: ~6.5 us of its duration); `achieved` divides the ENCODER's algorithmic FLOP by the whole launch -- the forward "
                            "alone is the dominant kernel of the line `bench.py --set HEAD_ON_FWD=0` (profiles/r05_bench_twelve_launches.json)")
        roof["traffic"], src = pmc_traffic(dom, f"{args.workload}_{args.model}_{args.dtype}")
        if src:
            roof["traffic_unit"] = "bytes/launch"
            roof["traffic_source"] = "replayed from profiles/" + src + " (PMC passes of this configuration; not collected in this run)"

    # ---- the same job with the epoch's data pipeline inside the clock: every epoch the loader shuffles, draws fresh negatives
    # (device sampler), packs the epoch and refills the resident pool; then one graph replay per batch (train_sr.py's epoch loop)
    loader_incl = None
    if use_real and use_pool and use_graph and world == 1 and args.model == "sasrec":
        from amid_amd.dataset_seq import DeviceBatches, DualDomainSeqDataset, JointBatches
        dss = [DualDomainSeqDataset.from_tokenised(os.path.join(FIXTURES, f"tok_{n}.npz")) for n in REAL[args.workload]]
        if len(dss) == 2:
            dss[1].shift_items(JOINT_OFFSET)
        lds = [DeviceBatches(d, Bw, shuffle=True, device=device, seed=1, negatives="device") for d in dss]
        ld2 = lds[0] if len(lds) == 1 else JointBatches(*lds)
        while (-eng.step) % n_pool != eng.input_pool(pl)[1]:       # finish the running epoch: the pool is refilled on its boundary
            eng.replay_train_step(pl)
        n_ep = max(2, min(20, args.steps // n_pool))

        def epoch():
            e = ld2.epoch_tensors()
            pk = eng.pack_epoch(pl, e["i_node"], e["neg_samples"], e["seq_d1"], e["seq_d2"], e["label"], e["domain_id"])
            eng.stream.wait_stream(torch.cuda.current_stream())
            if not eng.refill_input_pool(pl, pk):
                raise RuntimeError("pool refill off an epoch boundary")
            pk.record_stream(eng.stream)
            for k in range(0, n_pool - n_pool % spg, spg):        # as the CLI's train loop: graphs of spg steps, single steps for the tail
                if spg > 1:
                    eng.replay_train_steps(pl, spg)
                else:
                    eng.replay_train_step(pl)
            for _ in range(n_pool % spg if spg > 1 else 0):
                eng.replay_train_step(pl)

        epoch()
        barrier()
        t1 = time.perf_counter()
        for _ in range(n_ep):
            epoch()
        barrier()
        dt_l = time.perf_counter() - t1
        loader_incl = {"value": round(Bw * n_pool * n_ep / dt_l, 1), "unit": "samples/s", "epochs": n_ep, "batches_per_epoch": n_pool,
                       "ms_per_step": round(1e3 * dt_l / (n_pool * n_ep), 4),
                       "what": "per epoch: device shuffle + negative sampling + pack + pool refill, then one graph replay per batch"}

    if world > 1:
        dist.barrier()
    if rank == 0:
        out = {
            "metric": "train samples/sec", "value": round(Bw * world * args.steps / dt, 1), "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": data_desc,
            "config": {"workload": wl["label"] if args.model == "sasrec" else wl["label"].replace("SASRec", "BERT4Rec") + " [encoder: bert4rec]",
                       "batch_per_gpu": Bw,
                       "global_batch": Bw * world, "seq_len": T, "emb_dim": D, "hid_dim": HID, "neg": NEG, "table_rows": wl["n_rows"],
                       "unique_rows_last_step": int(pl.n_uniq.item()),
                       "input": "HBM-resident batch pool" if use_pool else "device copy per step",
                       "dropout": "on (p=0.5)" if args.model == "sasrec" else "on (p=0.1)", "optimizer": "Adam (dense-equivalent lazy rows)", "graph": use_graph, "steps_per_graph": spg,
                       "parallelism": f"dp{world}"},
            "loss_last": round(loss_last, 6),
            "prewarm_ms": round(prewarm_ms, 1), "prewarm_steps": prewarm_steps,
            "window_ms_per_step": [round(1e3 * w / args.steps, 4) for w in windows],
            "roofline": roof,
            "kernels": kernels,
        }
        if switches:
            out["config"]["switches"] = {k: str(v) for k, v in switches.items()}          # an A/B line, not the headline
        if loader_incl:
            out["samples_per_s_loader_included"] = loader_incl
        if world > 1:
            per = lambda k: round((ex1[k] - ex0[k]) / args.steps, 1)       # noqa: E731
            owner = ex1["owner_steps"] > ex0["owner_steps"]
            dense_bytes = eng.dense.numel * 4
            from amid_amd.dist import packed_rows
            sparse_bytes = packed_rows(min(umax_pool[0], eng.n_sparse_train(pl, dp=True)), D)[1] * D * 4
            out["dist"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                           "collectives_per_step": per("collectives"), "bytes_sent_per_step_per_rank": per("bytes_out"),
                           "bytes_received_per_step_per_rank": per("bytes_in"),
                           "sparse_exchange": "owner-bucketed (all-to-all + all-gather)" if owner
                           else "all-gather of per-rank unique rows",
                           "dense_exchange": "all-reduce (eager owner-bucketed step)" if owner else args.dense_exchange,
                           "dense_gradient_bytes": dense_bytes, "sparse_chunk_bytes_per_rank": sparse_bytes,
                           # what either dense exchange puts on this rank's receive side per step at this world size
                           "dense_bytes_received_per_step": {"gather": world * dense_bytes, "allreduce": dense_bytes},
                           "padded_unique_rows_per_rank": umax_pool[0]}
        if world == 1 and args.workload == "cfg2" and args.model == "sasrec" and args.dtype == "f32" and not args.no_stress:
            try:
                out["gather_stress"] = gather_stress(device)
            except Exception as e:              # a side measurement must never cost the headline line
                out["gather_stress"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and args.workload == "cfg2" and args.model == "sasrec" and args.dtype == "f32" and not args.no_stress:
            try:
                out["eval"] = eval_throughput(eng, device, Bw, T)
            except Exception as e:
                out["eval"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu_baseline and world == 1 and args.workload == "cfg2":
            try:
                out["cpu_baseline"] = cpu_baseline()
                out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
                cpu_eval = out["cpu_baseline"].pop("eval", None)
                if cpu_eval is not None and isinstance(out.get("eval"), dict):
                    out["eval"]["cpu_baseline"] = cpu_eval
            except Exception as e:
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
