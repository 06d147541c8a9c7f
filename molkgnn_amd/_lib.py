"""ctypes binding of libmolkgnn_hip.so (the C ABI declared in include/molkgnn_hip.h).

The library is built in-tree by ``make -C molkgnn_amd/csrc`` (or
``__graft_entry__.build()``).  There is no CPU or PyTorch fallback: if the
library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MKGNN_LIB: a diagnostic build of the same library (make VARIANT=... in csrc/), e.g. with cycle stamps compiled in
LIB_PATH = os.environ.get("MKGNN_LIB") or os.path.join(_HERE, "libmolkgnn_hip.so")
MAX_DEGREE = 4
ABI_VERSION = 7


class KernelBank(C.Structure):
    _fields_ = [("num_kernels", C.c_int32), ("reserved", C.c_int32),
                ("x_center", C.c_void_p), ("x_support", C.c_void_p),
                ("edge_attr_support", C.c_void_p), ("p_support", C.c_void_p),
                ("support_attr_sc_weight", C.c_void_p), ("center_attr_sc_weight", C.c_void_p),
                ("edge_attr_support_sc_weight", C.c_void_p)]


class KernelBankGrad(C.Structure):
    _fields_ = [("x_center", C.c_void_p), ("x_support", C.c_void_p), ("edge_attr_support", C.c_void_p),
                ("support_attr_sc_weight", C.c_void_p), ("center_attr_sc_weight", C.c_void_p),
                ("edge_attr_support_sc_weight", C.c_void_p)]


class DegreeBucket(C.Structure):
    _fields_ = [("count", C.c_int64), ("selected_index", C.c_void_p), ("nei_index", C.c_void_p),
                ("nei_edge_attr", C.c_void_p), ("p_focal", C.c_void_p), ("nei_p", C.c_void_p), ("nei_edge_unit", C.c_void_p)]


class Saved(C.Structure):
    _fields_ = [("pair_state", C.c_void_p), ("chirality", C.c_void_p)]


class ReadoutParams(C.Structure):
    _fields_ = [("lin1_weight", C.c_void_p), ("lin1_bias", C.c_void_p), ("lin2_weight", C.c_void_p),
                ("lin2_bias", C.c_void_p), ("F", C.c_int32), ("H", C.c_int32), ("G", C.c_int32)]


class TailArgs(C.Structure):
    """``mkgnn_tail_args`` (include/molkgnn_hip.h): the fused tail of a training step."""
    _fields_ = [("sim", C.c_void_p), ("sim_stride", C.c_int64), ("num_kernels", C.c_int32 * 4), ("buckets", C.c_void_p),
                ("in_rowptr", C.c_void_p), ("in_col", C.c_void_p), ("out_rowptr", C.c_void_p), ("out_col", C.c_void_p),
                ("mol_ptr", C.c_void_p), ("atom_mol", C.c_void_p), ("n_atoms", C.c_int64), ("n_mols", C.c_int64),
                ("n_loss_mols", C.c_int64), ("readout", ReadoutParams), ("head_weight", C.c_void_p), ("head_bias", C.c_void_p),
                ("target", C.c_void_p), ("dropout_p", C.c_float), ("rng_state", C.c_void_p), ("rng_used", C.c_void_p),
                ("emb", C.c_void_p), ("emb_stride", C.c_int64), ("pred", C.c_void_p), ("loss", C.c_void_p),
                ("grad_sim", C.c_void_p), ("grad_sim_stride", C.c_int64), ("grad_lin1_weight", C.c_void_p),
                ("grad_lin1_bias", C.c_void_p), ("grad_lin2_weight", C.c_void_p), ("grad_lin2_bias", C.c_void_p),
                ("grad_head_weight", C.c_void_p), ("grad_head_bias", C.c_void_p), ("defer_reduce", C.c_int32)]


TAIL_MAX_ATOMS, TAIL_MAX_EDGES = 128, 512      # MKGNN_TAIL_MAX_ATOMS / _EDGES


class CopyItem(C.Structure):
    """``mkgnn_copy_item``."""
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("numel", C.c_int64)]


class AdamWTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("state", C.c_void_p), ("numel", C.c_int64),
                ("group", C.c_int32), ("reserved", C.c_int32), ("active", C.c_void_p)]


class AdamWGroup(C.Structure):
    _fields_ = [("lr_device", C.c_void_p), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("weight_decay", C.c_float), ("maximize", C.c_int32), ("grad_scale", C.c_float)]


class ShardView(C.Structure):
    _fields_ = [("x", C.c_void_p), ("p", C.c_void_p), ("edge_src", C.c_void_p), ("edge_dst", C.c_void_p),
                ("edge_attr", C.c_void_p), ("y", C.c_void_p), ("mol_atom_ptr", C.c_void_p), ("mol_edge_ptr", C.c_void_p),
                ("mol_deg_ptr", C.c_void_p), ("n_molecules", C.c_int64), ("x_dim", C.c_int32), ("p_dim", C.c_int32),
                ("e_dim", C.c_int32), ("reserved", C.c_int32)]


Int64x6 = C.c_int64 * 6
Banks4 = KernelBank * MAX_DEGREE
BankGrads4 = KernelBankGrad * MAX_DEGREE
Buckets4 = DegreeBucket * MAX_DEGREE
Saved4 = Saved * MAX_DEGREE
Int32x4 = C.c_int32 * MAX_DEGREE

MOLECULE_MAX_LAYERS = 4
MOLECULE_MAX_ATOMS = 64
MOLECULE_MAX_MOLS = 16
MOLECULE_HEAD, MOLECULE_BACKWARD, MOLECULE_GRAD_EMB = 1, 2, 4


class MoleculeLayer(C.Structure):
    _fields_ = [("bank", Banks4), ("grad", BankGrads4), ("saved", Saved4), ("F", C.c_int32), ("reserved", C.c_int32),
                ("sim_out", C.c_void_p), ("sim_stride", C.c_int64)]


class BnStats(C.Structure):
    """``mkgnn_bn_stats``: the statistics-only companion of a batch norm (reference MolKGNNNet.py:116)."""
    _fields_ = [("x", C.c_void_p), ("x_stride", C.c_int64), ("n_rows", C.c_int64), ("C", C.c_int32),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("momentum", C.c_float),
                ("num_batches_tracked", C.c_void_p), ("row_key", C.c_void_p), ("key_limit", C.c_void_p)]


class MoleculeNet(C.Structure):
    _fields_ = [("num_layers", C.c_int32), ("E", C.c_int32), ("layer", MoleculeLayer * MOLECULE_MAX_LAYERS),
                ("bn_weight", C.c_void_p), ("bn_bias", C.c_void_p), ("bn_running_mean", C.c_void_p),
                ("bn_running_var", C.c_void_p), ("bn_num_batches_tracked", C.c_void_p),
                ("bn_eps", C.c_float), ("bn_momentum", C.c_float), ("bn_training", C.c_int32), ("reserved", C.c_int32),
                ("grad_bn_weight", C.c_void_p), ("grad_bn_bias", C.c_void_p),
                ("readout", ReadoutParams),
                ("grad_lin1_weight", C.c_void_p), ("grad_lin1_bias", C.c_void_p), ("grad_lin2_weight", C.c_void_p),
                ("grad_lin2_bias", C.c_void_p),
                ("ffn_weight", C.c_void_p), ("ffn_bias", C.c_void_p), ("grad_ffn_weight", C.c_void_p),
                ("grad_ffn_bias", C.c_void_p), ("head_dropout", C.c_float), ("reserved2", C.c_int32),
                ("rng_state", C.c_void_p), ("rng_used", C.c_void_p), ("edge_stats", C.POINTER(BnStats))]


class MoleculeBatch(C.Structure):
    _fields_ = [("n_atoms", C.c_int64), ("n_mols", C.c_int64), ("n_chunks", C.c_int64), ("max_chunk_atoms", C.c_int64),
                ("chunk_mol_ptr", C.c_void_p), ("mol_atom_ptr", C.c_void_p), ("atom_degree", C.c_void_p),
                ("atom_rank", C.c_void_p), ("buckets", Buckets4), ("x", C.c_void_p), ("x_stride", C.c_int64)]


EXPORTS = ("mkgnn_abi_version", "mkgnn_last_error", "mkgnn_row_inv_norm", "mkgnn_unit_rows8", "mkgnn_workspace_bytes",
           "mkgnn_kernelsetconv_forward", "mkgnn_kernelsetconv_backward", "mkgnn_segment_sum_rows",
           "mkgnn_readout_hidden_stride", "mkgnn_readout_workspace_bytes", "mkgnn_readout_forward",
           "mkgnn_readout_backward", "mkgnn_batchnorm_workspace_bytes", "mkgnn_batchnorm_forward",
           "mkgnn_batchnorm_backward", "mkgnn_bce_head_workspace_bytes", "mkgnn_bce_head_forward",
           "mkgnn_bce_head_backward", "mkgnn_rf_workspace_bytes", "mkgnn_rf_count", "mkgnn_rf_fill", "mkgnn_adamw_step", "mkgnn_adamw_state_floats",
           "mkgnn_bce_head_dropout_forward", "mkgnn_bce_head_dropout_backward", "mkgnn_segment_sum_block_rows",
           "mkgnn_plan_workspace_bytes", "mkgnn_plan_build", "mkgnn_backward_join", "mkgnn_backward_streams", "mkgnn_bank_prepare", "mkgnn_bank_prepare_deferred", "mkgnn_bank_prepare_flush", "mkgnn_bank_prepare_withdraw", "mkgnn_touch_hint", "mkgnn_expand_batch", "mkgnn_bce_head_fused", "mkgnn_collate_compact", "mkgnn_collate_compact_bytes",
           "mkgnn_readout_blocks_supported", "mkgnn_readout_blocks_forward", "mkgnn_readout_blocks_backward",
           "mkgnn_readout_blocks_workspace_bytes", "mkgnn_molecule_supported", "mkgnn_molecule_workspace_bytes",
           "mkgnn_molecule_step", "mkgnn_batchnorm_stats_workspace_bytes", "mkgnn_batchnorm_update_stats",
           "mkgnn_batchnorm_forward_with_stats", "mkgnn_index_workspace_bytes", "mkgnn_index_build",
           "mkgnn_rows_split_supported", "mkgnn_rows_presplit", "mkgnn_tail_supported", "mkgnn_tail_workspace_bytes", "mkgnn_tail_fused", "mkgnn_tail_flush", "mkgnn_flat_copy")

_lib: Optional[C.CDLL] = None
TORCH_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libmolkgnn_torch.so")
_torch_ops_loaded = False


def load_torch_ops():
    """Register ``torch.ops.molkgnn.*`` (csrc/torch_ops.cpp: TORCH_LIBRARY over the C ABI); raises if it is not built."""
    global _torch_ops_loaded
    import torch
    if not _torch_ops_loaded:
        load()                                   # (the shim links against the C-ABI library: same directory, $ORIGIN)
        if not os.path.exists(TORCH_LIB_PATH):
            raise MolKGNNLibraryError(f"{TORCH_LIB_PATH} is missing: build it with `make -C molkgnn_amd/csrc torch`")
        torch.ops.load_library(TORCH_LIB_PATH)
        if int(torch.ops.molkgnn.abi_version()) != ABI_VERSION:
            raise MolKGNNLibraryError("libmolkgnn_torch.so was built against another ABI version")
        _torch_ops_loaded = True
    return torch.ops.molkgnn


class MolKGNNLibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load (once) and type the shared library; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MolKGNNLibraryError(
            f"{LIB_PATH} is missing: build it with `make -C molkgnn_amd/csrc` "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise MolKGNNLibraryError(f"{LIB_PATH} does not export {name}")
    lib.mkgnn_abi_version.restype = C.c_int
    lib.mkgnn_last_error.restype = C.c_char_p
    lib.mkgnn_row_inv_norm.restype = C.c_int
    lib.mkgnn_row_inv_norm.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    lib.mkgnn_unit_rows8.restype = C.c_int
    lib.mkgnn_unit_rows8.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    lib.mkgnn_workspace_bytes.restype = C.c_size_t
    lib.mkgnn_workspace_bytes.argtypes = [Int32x4, C.c_int32, C.c_int32, C.c_int64, C.c_int64]
    lib.mkgnn_kernelsetconv_forward.restype = C.c_int
    lib.mkgnn_kernelsetconv_forward.argtypes = [
        Banks4, Buckets4, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
        C.c_void_p, C.c_int64, Saved4, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]
    lib.mkgnn_kernelsetconv_backward.restype = C.c_int
    lib.mkgnn_kernelsetconv_backward.argtypes = [
        Banks4, Buckets4, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
        C.c_void_p, C.c_int64, Saved4, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, BankGrads4,
        C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_void_p]
    lib.mkgnn_backward_streams.restype = C.c_int
    lib.mkgnn_backward_streams.argtypes = [Banks4, Buckets4, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32]
    lib.mkgnn_tail_supported.restype = C.c_int
    lib.mkgnn_tail_supported.argtypes = [C.c_int32, C.c_int32, C.c_int32, Int32x4]
    lib.mkgnn_tail_workspace_bytes.restype = C.c_size_t
    lib.mkgnn_tail_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64]
    lib.mkgnn_tail_fused.restype = C.c_int
    lib.mkgnn_tail_fused.argtypes = [C.POINTER(TailArgs), C.c_void_p, C.c_size_t, C.c_void_p]
    lib.mkgnn_flat_copy.restype = C.c_int
    lib.mkgnn_flat_copy.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.mkgnn_tail_flush.restype = C.c_int
    lib.mkgnn_tail_flush.argtypes = [C.c_void_p]
    lib.mkgnn_rows_presplit.restype = C.c_int
    lib.mkgnn_rows_presplit.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.mkgnn_rows_split_supported.restype = C.c_int
    lib.mkgnn_rows_split_supported.argtypes = [Banks4, Buckets4, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32]
    lib.mkgnn_bank_prepare.restype = C.c_int
    lib.mkgnn_bank_prepare.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mkgnn_bank_prepare_deferred.restype = C.c_int
    lib.mkgnn_bank_prepare_deferred.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.mkgnn_bank_prepare_flush.restype = C.c_int
    lib.mkgnn_bank_prepare_flush.argtypes = [C.c_void_p]
    lib.mkgnn_bank_prepare_withdraw.restype = C.c_int
    lib.mkgnn_bank_prepare_withdraw.argtypes = []
    lib.mkgnn_touch_hint.restype = C.c_int
    lib.mkgnn_touch_hint.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.mkgnn_collate_compact_bytes.restype = C.c_size_t
    lib.mkgnn_collate_compact_bytes.argtypes = [Int64x6, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.mkgnn_collate_compact.restype = C.c_int
    lib.mkgnn_collate_compact.argtypes = [C.POINTER(ShardView), C.c_int64, C.c_int64, Int64x6, C.c_int32, C.c_void_p, C.c_size_t]
    lib.mkgnn_expand_batch.restype = C.c_int
    lib.mkgnn_expand_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mkgnn_backward_join.restype = C.c_int
    lib.mkgnn_backward_join.argtypes = [C.c_void_p]
    lib.mkgnn_segment_sum_rows.restype = C.c_int
    lib.mkgnn_segment_sum_rows.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                           C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    P, I64, I32, F32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float
    lib.mkgnn_readout_hidden_stride.restype = I32
    lib.mkgnn_readout_hidden_stride.argtypes = [I32]
    lib.mkgnn_readout_workspace_bytes.restype = C.c_size_t
    lib.mkgnn_readout_workspace_bytes.argtypes = [I32, I32, I32, I64, I64]
    lib.mkgnn_readout_forward.restype = C.c_int
    lib.mkgnn_readout_forward.argtypes = [C.POINTER(ReadoutParams), P, I64, I64, P, I64, P, P, P, P, I64, P]
    lib.mkgnn_readout_backward.restype = C.c_int
    lib.mkgnn_readout_backward.argtypes = [C.POINTER(ReadoutParams), P, I64, I64, P, P, I64, P, P, P, P, I64, P, I64,
                                           P, P, P, P, P, C.c_size_t, P]
    lib.mkgnn_readout_blocks_workspace_bytes.restype = C.c_size_t
    lib.mkgnn_readout_blocks_workspace_bytes.argtypes = [I32, I32, I32, I64]
    lib.mkgnn_readout_blocks_supported.restype = C.c_int
    lib.mkgnn_readout_blocks_supported.argtypes = [I32, I32, I32, Int32x4]
    lib.mkgnn_readout_blocks_forward.restype = C.c_int
    lib.mkgnn_readout_blocks_forward.argtypes = [C.POINTER(ReadoutParams), P, I64, Int32x4, Buckets4, I64, P, P, P, I64, P, P, P, P, P, P, I64, P]
    lib.mkgnn_readout_blocks_backward.restype = C.c_int
    lib.mkgnn_readout_blocks_backward.argtypes = [C.POINTER(ReadoutParams), P, I64, Int32x4, Buckets4, I64, P, P, P, P, I64, P, P, P,
                                                  P, I64, P, P, I64, P, P, P, P, P, C.c_size_t, P]
    lib.mkgnn_batchnorm_workspace_bytes.restype = C.c_size_t
    lib.mkgnn_batchnorm_workspace_bytes.argtypes = [I32]
    lib.mkgnn_batchnorm_forward.restype = C.c_int
    lib.mkgnn_batchnorm_forward.argtypes = [P, I64, I64, I32, P, P, P, P, F32, F32, I32, P, I64, P, P, P, P, P, P, C.c_size_t, P]
    lib.mkgnn_batchnorm_stats_workspace_bytes.restype = C.c_size_t
    lib.mkgnn_batchnorm_stats_workspace_bytes.argtypes = [I32]
    lib.mkgnn_batchnorm_update_stats.restype = C.c_int
    lib.mkgnn_batchnorm_update_stats.argtypes = [C.POINTER(BnStats), P, C.c_size_t, P]
    lib.mkgnn_batchnorm_forward_with_stats.restype = C.c_int
    lib.mkgnn_batchnorm_forward_with_stats.argtypes = [P, I64, I64, I32, P, P, P, P, F32, F32, I32, P, I64, P, P, P, P, P, P, C.c_size_t,
                                                       C.POINTER(BnStats), P, C.c_size_t, P]
    lib.mkgnn_batchnorm_backward.restype = C.c_int
    lib.mkgnn_batchnorm_backward.argtypes = [P, I64, P, I64, I64, I32, P, P, P, I32, P, I64, P, P, P, P, C.c_size_t, P]
    lib.mkgnn_bce_head_forward.restype = C.c_int
    lib.mkgnn_bce_head_forward.argtypes = [P, I64, I64, I32, P, P, P, P, P, P, C.c_size_t, P]
    lib.mkgnn_bce_head_backward.restype = C.c_int
    lib.mkgnn_bce_head_backward.argtypes = [P, I64, I64, I32, P, P, P, P, P, I64, P, P, P, C.c_size_t, P]
    lib.mkgnn_bce_head_dropout_forward.restype = C.c_int
    lib.mkgnn_bce_head_dropout_forward.argtypes = [P, I64, I64, I32, P, P, P, C.c_float, P, P, P, P, P, C.c_size_t, P]
    lib.mkgnn_bce_head_dropout_backward.restype = C.c_int
    lib.mkgnn_bce_head_dropout_backward.argtypes = [P, I64, I64, I32, P, P, P, P, C.c_float, P, P, I64, P, P, P, C.c_size_t, P]
    lib.mkgnn_bce_head_fused.restype = C.c_int
    lib.mkgnn_bce_head_fused.argtypes = [P, I64, I64, I32, P, P, P, C.c_float, P, P, P, P, P, I64, P, P, P, C.c_size_t, P]
    lib.mkgnn_bce_head_workspace_bytes.restype = C.c_size_t
    lib.mkgnn_bce_head_workspace_bytes.argtypes = [I64, I32]
    lib.mkgnn_rf_workspace_bytes.restype = C.c_size_t
    lib.mkgnn_rf_workspace_bytes.argtypes = [I64]
    lib.mkgnn_rf_count.restype = C.c_int
    lib.mkgnn_rf_count.argtypes = [P, I64, I64, P, C.c_size_t, P, P]
    lib.mkgnn_rf_fill.restype = C.c_int
    lib.mkgnn_rf_fill.argtypes = [P, P, P, I64, I64, I32, P, Buckets4, P]
    lib.mkgnn_segment_sum_block_rows.restype = C.c_int
    lib.mkgnn_segment_sum_block_rows.argtypes = [P, I64, P, P, P, I64, Int32x4, I32, P, I64, P, P]
    lib.mkgnn_plan_workspace_bytes.restype = C.c_size_t
    lib.mkgnn_plan_workspace_bytes.argtypes = [I64, I64, I64]
    lib.mkgnn_plan_build.restype = C.c_int
    lib.mkgnn_plan_build.argtypes = [Buckets4, I64, P, I64, P, P, P, P, P, P, P, P, P, C.c_size_t, P]
    lib.mkgnn_index_workspace_bytes.restype = C.c_size_t
    lib.mkgnn_index_workspace_bytes.argtypes = [I64, I64, I64]
    lib.mkgnn_index_build.restype = C.c_int
    lib.mkgnn_index_build.argtypes = [P, P, P, I64, I64, I32, Buckets4, P, P, P, P, P, P, P, P, P, P, C.c_size_t, P, P]
    lib.mkgnn_adamw_step.restype = C.c_int
    lib.mkgnn_adamw_state_floats.restype = C.c_int64
    lib.mkgnn_adamw_state_floats.argtypes = [I64]
    lib.mkgnn_adamw_step.argtypes = [P, I32, P, I32, P]
    lib.mkgnn_molecule_supported.restype = C.c_int
    lib.mkgnn_molecule_supported.argtypes = [C.POINTER(MoleculeNet), I32]
    lib.mkgnn_molecule_workspace_bytes.restype = C.c_size_t
    lib.mkgnn_molecule_workspace_bytes.argtypes = [C.POINTER(MoleculeNet), I32, I64, I64]
    lib.mkgnn_molecule_step.restype = C.c_int
    lib.mkgnn_molecule_step.argtypes = [C.POINTER(MoleculeNet), C.POINTER(MoleculeBatch), I32, P, P, P, P, P, P, C.c_size_t, P]
    if lib.mkgnn_abi_version() != ABI_VERSION:
        raise MolKGNNLibraryError(f"ABI version {lib.mkgnn_abi_version()} != {ABI_VERSION}: rebuild the library")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().mkgnn_last_error()
        raise MolKGNNLibraryError(f"{what}: {msg.decode() if msg else 'error'}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None or t.numel() == 0 else t.data_ptr()


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu_tensor(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise MolKGNNLibraryError(
            f"{name} is on {t.device}: the kernel convolution runs on an MI355X only (no CPU fallback)")
