"""The N-split build of the one-launch encoder forward (csrc/sasrec_seqn.hip: NS waves share a 16-row strip, each owning D / NS
columns) against the whole-row build (csrc/sasrec_seq.hip, itself held to the oracle and the reference goldens by
test_gpu_sasrec.py / test_gpu_timed_path.py) through the C ABI, on the same inputs: both sum every product over k in the same order,
draw the same dropout counters and save the same tensors, so everything a layer saves -- x, qn, q, k, v, o, row statistics, r, y, h --
and the encoder output must agree BIT FOR BIT, train and eval, with and without the live-sequence list (rows outside the list must
stay untouched).  Reference arithmetic: Log2feats.forward, model_seq.py:371-383."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

D, H = 128, 8
# (B, T, variants): every (strips per sequence, column parts) build that covers T
CASES = [(256, 50, (42,)), (37, 64, (42,)), (9, 33, (42,)), (64, 20, (22, 24)), (300, 32, (22, 24)), (5, 17, (22, 24)),
         (64, 16, (14, 18)), (300, 9, (14, 18)), (3, 1, (14, 18))]


def _setup(B, T, seed, live):
    from amid_amd._lib import lib, ptr_array
    L = lib()
    g = torch.Generator().manual_seed(seed)
    M = B * T
    host = (ctypes.c_ubyte * L.value("amid_step_state_bytes"))()
    L.call("amid_step_state_pack", ctypes.addressof(host), 77, 3, 5e-4, 0.9, 0.999, 1e-8)
    st = torch.frombuffer(bytearray(host), dtype=torch.uint8).cuda()
    dom = (torch.rand(B, generator=g) < 0.5).long()
    if live == "all1":
        dom[:] = 1
    lv, row_live = None, torch.ones(2 * M, dtype=torch.bool)
    if live:
        d0, d1 = torch.nonzero(dom == 0).flatten(), torch.nonzero(dom != 0).flatten()
        lv = torch.cat((d0, d1, torch.tensor([d0.numel()]))).int().cuda()
        row_live = torch.cat((dom == 0, dom != 0)).repeat_interleave(T)
    rnd = lambda *s, sc=0.1, base=0.0: (base + sc * torch.randn(*s, generator=g)).cuda()      # noqa: E731
    fam = lambda *s, **k: [rnd(*s, **k) for _ in range(4)]                                      # noqa: E731  [layer][domain]
    P = dict(ln1_w=fam(D, base=1.0), ln1_b=fam(D), w_in=fam(3 * D, D), b_in=fam(3 * D), w_o=fam(D, D), b_o=fam(D), ln2_w=fam(D, base=1.0),
             ln2_b=fam(D), w1=fam(D, D), b1=fam(D), w2=fam(D, D), b2=fam(D))
    x0 = rnd(2 * M, D, sc=1.0)
    tmq = ((torch.rand(2 * M, D // 4, generator=g) < 0.04).to(torch.uint8) * 9).cuda()
    return L, ptr_array, st, lv, row_live, P, x0, tmq


def _run(L, pa, variant, B, T, st, lv, P, x0, tmq, train, bf16=False):
    M = B * T
    nan = lambda *s: torch.full(s, float("nan"), device="cuda")      # noqa: E731
    saved = {k: [nan(2 * M, D) for _ in range(2)] for k in "qn q k v o r y h".split()}
    saved["stats"] = [nan(2 * M, H, 2) for _ in range(2)]
    x1, xout = nan(2 * M, D), nan(2 * M, D)
    tl = lambda ts: pa([t.data_ptr() for t in ts])      # noqa: E731
    prev = L.value("amid_sas_seq_fwd_variant", variant)
    try:
        args = (2, tl([x0, x1]), xout.data_ptr(), *[tl(P[k]) for k in "ln1_w ln1_b w_in b_in w_o b_o ln2_w ln2_b w1 b1 w2 b2".split()],
                *[tl(saved[k]) for k in "qn q k v o stats r y h".split()], tmq.data_ptr(), 1e-8, B, T, D, H,
                lv.data_ptr() if lv is not None else None, st.data_ptr(), train, 0.5)
        s = torch.cuda.current_stream().cuda_stream
        if bf16:        # the weights' bf16 fragment images, [layer][domain][q, k, v, o, c1, c2]; bf16 = 3: three planes (hi, mid, lo) each
            planes = 3 if bf16 == 3 else 1
            w16 = torch.empty(2, 2, 6, planes, D * D, dtype=torch.bfloat16, device="cuda")
            srcs = []
            for i in range(4):
                srcs += [P["w_in"][i].data_ptr() + 4 * j * D * D for j in range(3)] + [P["w_o"][i].data_ptr(), P["w1"][i].data_ptr(), P["w2"][i].data_ptr()]
            L.call("amid_sas_weights_bf16_planes", pa(srcs), 24, D, 0, planes, w16.data_ptr(), s)
            L.call("amid_sas_seq_fwd_split_f32" if planes == 3 else "amid_sas_seq_fwd_bf16w_f32", *args, w16.data_ptr(), s)
        else:
            L.call("amid_sas_seq_fwd_f32", *args, s)
        torch.cuda.synchronize()
    finally:
        L.value("amid_sas_seq_fwd_variant", prev)
    out = {f"{k}{l}": v[l] for k, v in saved.items() for l in (0, 1)}
    out["x1"], out["xout"] = x1, xout
    return out


@pytest.mark.parametrize("B,T,variants", CASES)
@pytest.mark.parametrize("live", [None, "mixed", "all1"])
@pytest.mark.parametrize("train", [0, 1])
def test_n_split_forward_is_bit_identical_to_whole_row_forward(B, T, variants, live, train):
    L, pa, st, lv, row_live, P, x0, tmq = _setup(B, T, seed=B * 13 + T, live=live)
    ref = _run(L, pa, 1, B, T, st, lv, P, x0, tmq, train)
    rl = row_live.cuda()
    for v in variants:
        got = _run(L, pa, v, B, T, st, lv, P, x0, tmq, train)
        for name, want in ref.items():
            a, b = got[name], want
            assert torch.isfinite(b[rl]).all(), (name, "reference build")
            assert torch.equal(a[rl], b[rl]), (v, name, float((a[rl] - b[rl]).abs().max()))
            if bool((~rl).any()):
                assert torch.isnan(a[~rl]).all(), f"variant {v} {name}: rows outside the live list were written"


@pytest.mark.parametrize("B,T,variants", CASES)
@pytest.mark.parametrize("live", [None, "mixed"])
@pytest.mark.parametrize("train", [0, 1])
def test_forward_on_bf16_pieces_has_fp32_accuracy(B, T, variants, live, train):
    """amid_sas_seq_fwd_split_f32 -- the twelve projections as six bf16 piece-pair products per fp32 product (three pieces per operand
    element, their sum the element exactly) -- against amid_sas_seq_fwd_f32 (fp32 matrix instructions) on the same inputs, dropout
    counters and live list: every saved tensor and the output within 4e-6 of the tensor's largest entry (two fp32 evaluations of the
    same chain in different summation orders differ by as much), row statistics within 1e-5, rows outside the live list untouched.
    A product with bf16-ROUNDED operands is off by 3e-3 on the same data (asserted: the one-plane build is not what runs)."""
    L, pa, st, lv, row_live, P, x0, tmq = _setup(B, T, seed=B * 7 + T, live=live)
    rl = row_live.cuda()
    for v in variants:
        ref = _run(L, pa, v, B, T, st, lv, P, x0, tmq, train)
        got = _run(L, pa, v, B, T, st, lv, P, x0, tmq, train, bf16=3)
        worst = 0.0
        for name, want in ref.items():
            a, b = got[name][rl], want[rl]
            assert torch.isfinite(a).all(), (v, name)
            bar = (1e-5 if name.startswith("stats") else 4e-6) * max(1.0, float(b.abs().max()))      # (measured: up to 3.1e-6 with the hi plane walked first)
            err = float((a - b).abs().max())
            worst = max(worst, err / max(1.0, float(b.abs().max())))
            assert err < bar, (v, name, err, bar)
            if bool((~rl).any()):
                assert torch.isnan(got[name][~rl]).all(), f"variant {v} {name}: rows outside the live list were written"
        if (B, T) == (256, 50):
            one = _run(L, pa, v, B, T, st, lv, P, x0, tmq, train, bf16=1)
            e1 = float((one["xout"][rl] - ref["xout"][rl]).abs().max() / ref["xout"][rl].abs().max())
            assert e1 > 100 * worst, (e1, worst)


@pytest.mark.parametrize("B,T,variant", [(256, 50, 42), (300, 50, 42), (256, 20, 24), (64, 9, 14)])
def test_forward_on_bf16_pieces_repeats_bit_for_bit(B, T, variant):
    """A race screen for the hand-placed LDS reads, the plane ring's DMA and the piece exchange of seqn_fwd_px_kernel: 25 launches on the
    same inputs (train mode, mixed live list; B 300 > the CU count: workgroups of two rounds share CUs) give the same bits in every saved
    tensor.  (A read that passes its data, a fragment register refilled under a matrix instruction or a plane overwritten early would
    show up as rare differing tiles.)"""
    L, pa, st, lv, row_live, P, x0, tmq = _setup(B, T, seed=B + 3 * T, live="mixed")
    rl = row_live.cuda()
    first = _run(L, pa, variant, B, T, st, lv, P, x0, tmq, 1, bf16=3)
    for name, t in first.items():
        assert torch.isfinite(t[rl]).all(), name
    for rep in range(24):
        got = _run(L, pa, variant, B, T, st, lv, P, x0, tmq, 1, bf16=3)
        for name, want in first.items():
            assert torch.equal(got[name][rl], want[rl]), (rep, name, float((got[name][rl] - want[rl]).abs().max()))


def test_three_plane_weight_images_sum_to_the_weights_exactly():
    """amid_sas_weights_bf16_planes(planes = 3): hi + mid + lo (each a bf16, summed in fp32 in that order... exactly representable steps)
    reproduces every fp32 weight bit for bit, plane 0 is the one-plane (round-to-nearest-even) image, all three in fragment order."""
    from amid_amd._lib import lib, ptr_array
    L = lib()
    g = torch.Generator().manual_seed(6)
    W = [(torch.randn(D, D, generator=g) * torch.pow(10.0, -4.0 * torch.rand(D, D, generator=g))).cuda() for _ in range(2)]
    k = torch.arange(D)
    s_, h_, g_, r_ = k >> 5, (k >> 4) & 1, (k >> 2) & 3, k & 3
    pos = (4 * s_ + g_) * 8 + 4 * h_ + r_
    s = torch.cuda.current_stream().cuda_stream
    for tr in (0, 1):
        out3 = torch.zeros(2, 3, D, D, dtype=torch.bfloat16, device="cuda")
        out1 = torch.zeros(2, D, D, dtype=torch.bfloat16, device="cuda")
        L.call("amid_sas_weights_bf16_planes", ptr_array([w.data_ptr() for w in W]), 2, D, tr, 3, out3.data_ptr(), s)
        L.call("amid_sas_weights_bf16", ptr_array([w.data_ptr() for w in W]), 2, D, tr, out1.data_ptr(), s)
        torch.cuda.synchronize()
        for i in range(2):
            src = (W[i].t() if tr else W[i]).cpu().contiguous()
            assert torch.equal(out3[i, 0].cpu(), out1[i].cpu())
            planes = out3[i].cpu()[:, :, pos].double()             # back to [plane][n][k]
            assert torch.equal((planes[0] + planes[1] + planes[2]).float(), src), tr
            assert float((planes[1].abs() - planes[0].abs() * 2.0 ** -8).max()) <= 0.0 and float((planes[2].abs() - planes[0].abs() * 2.0 ** -16).max()) <= 0.0


def test_variant_switch_round_trips_and_refuses_a_split_that_does_not_cover_t():
    from amid_amd._lib import AmidError
    L, pa, st, lv, row_live, P, x0, tmq = _setup(4, 50, seed=1, live=None)
    prev = L.value("amid_sas_seq_fwd_variant", -1)
    assert L.value("amid_sas_seq_fwd_variant", 24) == prev and L.value("amid_sas_seq_fwd_variant", prev) == 24
    # an explicit (2 strips, ...) build cannot hold 50 rows: the entry falls back to the whole-row build instead of failing
    out = _run(L, pa, 24, 4, 50, st, lv, P, x0, tmq, 0)
    ref = _run(L, pa, 1, 4, 50, st, lv, P, x0, tmq, 0)
    assert torch.equal(out["xout"], ref["xout"])
    assert AmidError is not None


def test_weight_images_hold_the_rounded_weights_in_fragment_order():
    """amid_sas_weights_bf16: chunk 4 s + g of row n = bf16(W[n][32 s + 4 g + 0..3]), bf16(W[n][32 s + 16 + 4 g + 0..3]); transposed: of W^T."""
    from amid_amd._lib import lib, ptr_array
    L = lib()
    g = torch.Generator().manual_seed(5)
    W = [torch.randn(D, D, generator=g).cuda() for _ in range(3)]
    k = torch.arange(D)
    s_, h_, g_, r_ = k >> 5, (k >> 4) & 1, (k >> 2) & 3, k & 3
    pos = (4 * s_ + g_) * 8 + 4 * h_ + r_                    # where in-feature k of a row sits in the image
    for tr in (0, 1):
        out = torch.zeros(3, D, D, dtype=torch.bfloat16, device="cuda")
        L.call("amid_sas_weights_bf16", ptr_array([w.data_ptr() for w in W]), 3, D, tr, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for i in range(3):
            want = torch.empty(D, D, dtype=torch.bfloat16)
            src = (W[i].t() if tr else W[i]).cpu()
            want[:, pos] = src.to(torch.bfloat16)                # round to nearest even, as v_cvt_pk_bf16_f32
            assert torch.equal(out[i].cpu().view(torch.int16), want.view(torch.int16)), (tr, i)


@pytest.mark.parametrize("B,T,variants", [CASES[0], CASES[3], CASES[6]])
@pytest.mark.parametrize("live", [None, "mixed"])
@pytest.mark.parametrize("train", [0, 1])
def test_bf16_products_stay_within_bf16_rounding_of_the_fp32_forward(B, T, variants, live, train):
    """amid_sas_seq_fwd_bf16w_f32 (BASELINE.json configs[2]: the projections' operands rounded to bf16, fp32 accumulation and everything
    else) against the fp32 forward on the same inputs: every saved tensor within 2e-2 of its largest magnitude (the bar SURVEY.md
    section 8(c) sets for bf16 against the fp32 reference), and visibly different (the mode is on).  The relu / dropout decisions are
    the fp32 run's except where a pre-activation sits within rounding of zero, hence the L2 view for h and behind."""
    L, pa, st, lv, row_live, P, x0, tmq = _setup(B, T, seed=B * 17 + T, live=live)
    ref = _run(L, pa, 1, B, T, st, lv, P, x0, tmq, train)
    rl = row_live.cuda()
    for v in variants:
        got = _run(L, pa, v, B, T, st, lv, P, x0, tmq, train, bf16=True)
        worst = 0.0
        for name, want in ref.items():
            a, b = got[name][rl].double(), want[rl].double()
            assert torch.isfinite(a).all(), (v, name)
            e = float((a - b).norm() / (b.norm() + 1e-30))
            worst = max(worst, e)
            assert e < 2e-2, (v, name, e)
            if bool((~rl).any()):
                assert torch.isnan(got[name][~rl]).all(), f"variant {v} {name}: rows outside the live list were written"
        assert worst > 1e-4, "bf16 products left no trace: the fp32 kernel ran"
