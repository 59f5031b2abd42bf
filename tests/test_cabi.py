"""CPU-side checks of the C-ABI boundary: the built library loads and exports every entry point
declared in include/amid_hip.h (no compute calls: there is no GPU here)."""
import os
import subprocess

import pytest

from amid_amd import _lib


def test_header_parses_and_lists_entry_points():
    protos = _lib.parse_header()
    assert len(protos) >= 40
    for must in ("amid_gather_rows_f32", "amid_embed_fwd_f32", "amid_attn_fwd_f32", "amid_attn_bwd_f32", "amid_embgrad_segreduce_f32",
                 "amid_lazy_adam_apply_f32", "amid_sas_qkv_fwd_f32", "amid_sas_wgrad_f32", "amid_graph_launch"):
        assert must in protos


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    missing = [s for s in _lib.declared_symbols() if s not in exported]
    assert not missing, missing
    extra = [s for s in exported if s.startswith("amid_") and s not in _lib.declared_symbols()]
    assert not extra, f"exported but not declared in include/amid_hip.h: {extra}"


def test_library_loads_and_answers_host_only_queries():
    L = _lib.lib()
    assert L.value("amid_version") >= 100
    assert L.value("amid_rows_per_tile", 12800) == 100          # 25 600 rows over 256 CUs
    assert L.value("amid_rows_per_tile", 400) == 16
    assert L.value("amid_step_state_bytes") == 64          # seed, step, 4 doubles, ticket + pad, step_done
    assert L.value("amid_sort_unique_workspace_bytes", 26112) > 4 * 26112 * 4
    assert L.raw("amid_error_string")(-2).decode().startswith("amid: shape not supported")


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libamid_hip.so")
    with pytest.raises(_lib.AmidLibraryError):
        _lib._Lib()


def test_round5_entry_points_refuse_bad_arguments_without_a_gpu():
    """The argument checks of the round's new entry points return AMID_ERR_ARG (-1) / AMID_ERR_UNSUPPORTED before anything touches a device:
    null pointers, tile counts past the riders' table, a sort plan of another list, shard phases outside 1 / 2."""
    import ctypes
    L = _lib.lib()
    d = L._dll
    null = None
    one = (ctypes.c_int * 1)(128)
    ptrs = (ctypes.c_void_p * 1)(0)
    assert d.amid_diag_variants() == 0                                     # the product library carries no diagnostic build
    assert d.amid_bert_weight_images_f32(null, one, one, 1, null, null) == -1
    assert d.amid_bert_weight_images_f32(ptrs, one, one, 0, ptrs, null) == -1
    assert d.amid_bert_weight_images_f32(ptrs, one, one, 97, ptrs, null) == -1            # at most 96 tiles
    assert L.value("amid_scorer_vec_floats", 2, 32) == (2 * 32 + 2 * 32 + 32 + 1 + 3) // 4 * 4      # da [2][hid] | dc [NI][hid] | dw2 [hid] | db2, padded to 16 B
    # amid_step_head_f32: null pool
    rc = L._fn["amid_step_head_f32"](null, 0, 0, 0, null, 0, 4, 4, 1, 10, null, null, null, null, null, null, null, null, null, 128, null, null, null)
    assert rc == -1
    # the forward with the head on its tail: no piece images / no live list
    rc = L._fn["amid_sas_seq_fwd_split_lnstat_head_f32"](2, *[null] * 23, 1e-8, 4, 40, 128, 8, null, null, 1, 0.5, null, *[null] * 9, 2, 32,
                                                         *[null] * 10, null)
    assert rc == -1
    # the comp shards: a phase outside 1 / 2
    rc = L._fn["amid_bert_comp_fwd_shard_f32"](null, null, null, null, null, null, 0.5, 0, 4, 4, 128, 8, 0, 3, null, null, null, null, null, null)
    assert rc == -1


def test_round6_entry_points_refuse_bad_arguments_without_a_gpu():
    """The data-parallel form of the folded step's gradient tail: null lists, a bound past the list, a pack request without its id buffers."""
    L = _lib.lib()
    null = None
    f = L._fn["amid_grad_tail_live_dp_f32"]
    assert f(*[null] * 4, 64, 128, null, null, null, 1, null, 1, null, 4, 50, null, null, null, null, 0, 0, null, null, null, 0, null, null) == -1
    # the tail with the optimizer folded in: no ticket word / no lists
    f = L._fn["amid_grad_tail_opt_f32"]
    assert f(*[null] * 4, 64, 128, null, null, null, 1, null, 1, null, 4, 50, null, null, null, null, null, null, 100, 0, 100, null, null, null,
             null, null, null, 64, 1.0, null, null, null) == -1
    f = L._fn["amid_grad_tail_live_dp1_f32"]
    assert f(*[null] * 4, 64, 128, null, null, null, 1, null, 1, null, 4, 50, null, null, null, 100, 0, 100, null, null, 64, 0, 0, null, null, null,
             null, null) == -1
    # the forwards with the gather as their prologue: no piece images; the step head with riders: no weights
    assert L._fn["amid_sas_seq_fwd_gather_infer_f32"](2, null, *[null] * 12, 1e-8, 4, 40, 128, 8, null, null, null, null, null, null, null) == -1
    assert L._fn["amid_sas_seq_fwd_gather_f32"](2, *[null] * 23, 1e-8, 4, 40, 128, 8, null, null, 1, 0.5, null, null, 2, null, null, null, null, null) == -1
    rc = L._fn["amid_step_head_w16_f32"](null, 0, 0, 0, null, 0, 4, 4, 1, 10, null, null, null, null, null, null, null, null, null, 128, null, null, null, 0, 3,
                                         null, null, null)
    assert rc == -1
    # the inference forward: no piece images ; the evaluation head: no table
    assert L._fn["amid_sas_seq_fwd_split_infer_f32"](2, null, null, *[null] * 12, null, 1e-8, 4, 40, 128, 8, null, null, null) == -1
    assert L._fn["amid_eval_head_f32"](*[null] * 11, 4, 40, 100, 128, 32, 1e-8, 1e-7, null, null, null, null, null, null) == -1
