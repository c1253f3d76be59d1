"""ctypes binding of libgcr.so — the C ABI declared in include/gcr.h.

This is the binding a maintainer of the reference would add.  There is no
CPU fallback: if the HIP library is missing or a call fails, the op raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int32, c_int64, c_uint32, c_uint64, c_void_p

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libgcr.so")

_P = c_void_p  # every device / host buffer crosses the ABI as a plain pointer

# name -> (restype, argtypes); mirrors include/gcr.h one to one
SIGNATURES = {
    "gcr_version": (c_int32, []),
    "gcr_arch": (c_char_p, []),
    "gcr_spmm_plan_size_host": (c_int32, [_P, c_int64, c_int32, _P, _P, _P]),
    "gcr_spmm_plan_fill_host": (c_int32, [_P, c_int64, c_int32, _P, _P, _P]),
    "gcr_spmm_csr_f32": (c_int32, [_P, c_int64, _P, _P, c_int64, _P, _P, _P, _P, c_float, _P, c_int32,
                                   _P, _P, _P, c_float, c_uint32, _P, _P, c_int64, c_int64, _P]),
    "gcr_spmm_csr_acc2_f32": (c_int32, [_P, c_int64, _P, _P, c_int64, _P, _P, _P, _P, c_float, _P, c_int32,
                                        _P, _P, _P, c_float, _P, c_float, c_uint32, _P, _P, c_int64, c_int64, _P, _P]),
    "gcr_bitmap_set": (c_int32, [_P, c_int64, c_int64, _P, _P]),
    "gcr_spmm_csr_dual_f32": (c_int32, [_P, c_int64, _P, _P, c_int64, _P, _P, _P, _P, c_float, _P, c_int32,
                                        _P, _P, _P, _P, c_int64, c_int64, _P]),
    "gcr_spmm_csr_dual_acc_f32": (c_int32, [_P, c_int64, _P, _P, c_int64, _P, _P, _P, _P, c_float, _P, c_int32,
                                            _P, _P, _P, _P, _P, _P, c_int64, c_int64, _P]),
    "gcr_csr_validate": (c_int32, [_P, _P, c_int64, c_int64, c_int64, _P, _P]),
    "gcr_bpr_workspace_floats": (c_int64, [c_int64]),
    "gcr_bpr_fwd_f32": (c_int32, [_P, _P, c_int32, _P, _P, _P, c_int64, c_int32, c_int32, c_int64, c_int64,
                                  _P, _P, _P, _P]),
    "gcr_bpr_bwd_f32": (c_int32, [_P, _P, c_int32, _P, _P, _P, c_int64, c_int32, c_int64, c_int64,
                                  _P, _P, _P, _P, _P]),
    "gcr_sort_index_workspace_bytes": (c_int64, [c_int64]),
    "gcr_sort_index": (c_int32, [_P, c_int64, c_int64, _P, _P, _P, _P]),
    "gcr_bpr_edge_values_f32": (c_int32, [_P, _P, _P, c_int64, c_int64, _P, _P, _P, _P]),
    "gcr_bpr_bwd_sorted_f32": (c_int32, [_P, _P, c_int32, _P, _P, _P, c_int64, c_int32, c_int64, c_int64, _P, _P,
                                         _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gcr_bpr_neg_block_f32": (c_int32, [_P, _P, _P, c_int64, c_int32, c_int64, _P, _P, _P, _P, _P, _P, _P]),
    "gcr_sort_pairs_u64_workspace_bytes": (c_int64, [c_int64]),
    "gcr_sort_pairs_u64": (c_int32, [_P, _P, c_int64, c_int64, _P, _P, _P, _P]),
    "gcr_bpr_neg_items_sorted_f32": (c_int32, [_P, _P, c_int32, _P, _P, c_int64, c_int64, c_int64, _P, _P, _P]),
    "gcr_neg_sample": (c_int32, [_P, _P, _P, c_int64, c_int32, c_int64, c_int64, c_uint64, c_uint64, c_int32, _P, _P]),
    "gcr_edge_mask_bits": (c_int32, [c_int64, c_float, c_uint64, _P, _P, _P]),
    "gcr_row_inv_norm_f32": (c_int32, [_P, c_int64, c_int32, c_float, _P, _P]),
    "gcr_infonce_engine": (c_int32, [c_int32]),
    "gcr_infonce_fwd_workspace_bytes": (c_int64, [c_int64, c_int64, c_int32]),
    "gcr_infonce_fwd_f32": (c_int32, [_P, _P, c_int64, _P, _P, c_int64, c_int32, c_float, _P, _P, c_float, _P, _P]),
    "gcr_infonce_fwd_ex_f32": (c_int32, [_P, _P, c_int64, _P, _P, c_int64, c_int32, c_float, _P, _P, c_float, _P, c_uint32, _P]),
    "gcr_infonce_fwd_o_supported": (c_int32, [c_int32, c_uint32]),
    "gcr_infonce_fwd_o_workspace_bytes": (c_int64, [c_int64, c_int64, c_int32]),
    "gcr_infonce_fwd_o_f32": (c_int32, [_P, _P, c_int64, _P, _P, c_int64, c_int32, c_float, _P, _P, _P, c_uint32, _P]),
    "gcr_pos_logit_f32": (c_int32, [_P, _P, _P, _P, _P, c_int64, c_int64, c_int32, c_float, _P, _P]),
    "gcr_infonce_bwd_workspace_bytes": (c_int64, [c_int64, c_int64, c_int32]),
    "gcr_infonce_bwd_f32": (c_int32, [_P, _P, c_int64, _P, _P, c_int64, c_int32, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "gcr_infonce_bwd_ex_f32": (c_int32, [_P, _P, c_int64, _P, _P, c_int64, c_int32, c_float, _P, _P, _P, _P, _P, _P, c_uint32, _P]),
    "gcr_infonce_pos_bwd_f32": (c_int32, [_P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int32, c_float, _P, _P, _P]),
    "gcr_normalize_bwd_f32": (c_int32, [_P, _P, _P, c_int64, c_int32, _P, _P]),
    "gcr_normalize_bwd_n_f32": (c_int32, [_P, _P, _P, _P, c_int64, c_int32, _P, _P]),
    "gcr_normalize_bwd_raw_f32": (c_int32, [_P, _P, _P, _P, c_int64, c_int32, _P, _P]),
    "gcr_gram_tn_workspace_bytes": (c_int64, [c_int64, c_int32, c_int32]),
    "gcr_gram_tn_f32": (c_int32, [_P, _P, c_int64, c_int32, c_int32, _P, _P, _P]),
    "gcr_rows_dot_vec_f32": (c_int32, [_P, _P, c_int64, c_int32, _P, _P]),
    "gcr_weighted_colsum_workspace_bytes": (c_int64, [c_int64, c_int32]),
    "gcr_weighted_colsum_f32": (c_int32, [_P, _P, c_int64, c_int32, _P, _P, _P]),
    "gcr_gate_fwd_f32": (c_int32, [_P, _P, _P, c_int64, c_int32, _P, _P]),
    "gcr_gate_bwd_workspace_bytes": (c_int64, [c_int64, c_int32]),
    "gcr_gate_bwd_f32": (c_int32, [_P, _P, _P, _P, c_int64, c_int32, _P, _P, _P, _P, _P]),
    "gcr_channel_mix_supported": (c_int32, [c_int32]),
    "gcr_channel_mix_fwd_f32": (c_int32, [_P, _P, _P, _P, _P, c_float, c_int64, c_int32, _P, _P, _P]),
    "gcr_channel_mix_bwd_workspace_bytes": (c_int64, [c_int64, c_int32]),
    "gcr_channel_mix_bwd_f32": (c_int32, [_P, _P, _P, _P, _P, _P, c_float, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "gcr_bce_fwd_o_supported": (c_int32, [c_int32, c_uint32]),
    "gcr_bce_fwd_workspace_bytes": (c_int64, [c_int64, c_int64, c_int32]),
    "gcr_bce_fwd_f32": (c_int32, [_P, c_int64, _P, c_int64, c_int32, _P, _P, _P, c_uint32, _P]),
    "gcr_bce_bwd_workspace_bytes": (c_int64, [c_int64, c_int64, c_int32]),
    "gcr_bce_bwd_f32": (c_int32, [_P, c_int64, _P, c_int64, c_int32, _P, _P, _P, _P, c_uint32, _P]),
    "gcr_kmeans_assign_f32": (c_int32, [_P, c_int64, _P, _P, c_int64, c_int32, _P, _P, _P]),
    "gcr_kmeans_assign_accumulate_f32": (c_int32, [_P, c_int64, _P, _P, c_int64, c_int32, _P, _P, _P, c_int32, _P]),
    "gcr_kmeans_image_bytes": (c_int64, [c_int64, c_int32]),
    "gcr_kmeans_centroid_image_f32": (c_int32, [_P, _P, c_int64, c_int32, _P, _P]),
    "gcr_kmeans_search_image_f32": (c_int32, [_P, c_int64, _P, c_int64, c_int32, _P, _P, _P, c_int32, c_uint32, _P]),
    "gcr_kmeans_search_image_incr_f32": (c_int32, [_P, c_int64, _P, c_int64, c_int32, _P, _P, _P, _P, c_int32, c_uint32, _P]),
    "gcr_kmeans_lloyd_update_q_f32": (c_int32, [_P, _P, _P, c_int64, c_int32, _P, _P, _P, c_int64, c_uint64, c_int32, _P, _P, c_int32,
                                                _P]),
    "gcr_kmeans_update_f32": (c_int32, [_P, c_int64, c_int32, _P, c_int64, _P, _P, _P, _P, _P]),
    "gcr_kmeans_update_sorted_f32": (c_int32, [_P, c_int64, c_int32, _P, _P, c_int64, _P, _P, _P, _P, _P]),
    "gcr_kmeans_lloyd_update_f32": (c_int32, [_P, c_int64, c_int32, _P, _P, _P, c_int64, _P, _P, _P, _P, c_int32, c_uint64,
                                              c_int32, _P, _P]),
    "gcr_score_rows_f32": (c_int32, [_P, _P, c_int64, c_int64, _P, c_int64, c_int32, _P, _P]),
    "gcr_topk_masked_f32": (c_int32, [_P, c_int64, c_int64, _P, c_int64, _P, _P, c_int32, _P, _P, _P]),
    "gcr_rank_fused_supported": (c_int32, [c_int64, c_int32, c_int32]),
    "gcr_rank_fused_workspace_bytes": (c_int64, [c_int64]),
    "gcr_rank_fused_f32": (c_int32, [_P, _P, c_int64, c_int64, _P, c_int64, c_int32, _P, _P, c_int32, _P, _P, _P, _P, _P]),
    "gcr_rank_metrics": (c_int32, [_P, c_int64, c_int32, _P, _P, _P, c_int32, _P, _P, _P, _P]),
    "gcr_coo_to_csr_workspace_bytes": (c_int64, [c_int64]),
    "gcr_coo_to_csr": (c_int32, [_P, _P, _P, c_int64, c_int64, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gcr_csr_sym_norm_f32": (c_int32, [_P, _P, _P, c_int64, c_int64, _P, _P, _P, _P, _P, _P]),
    "gcr_csr_row_norm_f32": (c_int32, [_P, _P, _P, c_int64, _P, _P, _P]),
    "gcr_edge_mask_exact_workspace_bytes": (c_int64, [c_int64]),
    "gcr_edge_mask_exact_bits": (c_int32, [c_int64, c_int64, c_uint64, _P, _P, _P]),
    "gcr_adam_step_f32": (c_int32, [_P, _P, _P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int64,
                                    c_float, _P]),
    "gcr_adam_step_dev_f32": (c_int32, [_P, _P, _P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, _P,
                                        c_float, _P]),
    "gcr_mask_columns_f32": (c_int32, [_P, c_int64, c_int32, _P, _P, _P]),
    "gcr_spgemm_expand_f32": (c_int32, [_P, _P, _P, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gcr_csr_lookup_f32": (c_int32, [_P, _P, c_int64, _P, _P, _P, _P, _P]),
    "gcr_gather_rows_f32": (c_int32, [_P, _P, c_int64, c_int32, c_int64, _P, _P]),
    "gcr_scatter_add_rows_f32": (c_int32, [_P, _P, c_int64, c_int32, c_int64, _P, _P]),
    "gcr_dense_ids_workspace_bytes": (c_int64, [c_int64]),
    "gcr_dense_ids_u64": (c_int32, [_P, c_int64, c_int32, _P, _P, _P, _P, _P]),
    "gcr_probe_copy_f32": (c_int32, [_P, _P, c_int64, _P]),
    "gcr_probe_read_f32": (c_int32, [_P, c_int64, _P, _P]),
    "gcr_probe_gather_rows_f32": (c_int32, [_P, c_int64, _P, c_int64, _P, _P]),
}

_lib = None


class GcrError(RuntimeError):
    pass


def lib():
    """Loads libgcr.so once; raises loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GcrError(
                f"{LIB_PATH} is missing: the HIP extension is not built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        if rc <= -1000:
            raise GcrError(f"{what}: HIP runtime error {-(rc + 1000)}")
        raise GcrError(f"{what}: invalid or unsupported arguments (code {rc})")


def dptr(t):
    """Device (or host) pointer of a contiguous tensor, None -> NULL."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise GcrError("libgcr needs contiguous tensors")
    return c_void_p(t.data_ptr())


def cur_stream(device=None):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise GcrError("libgcr operates on HIP device tensors only (no CPU fallback)")
