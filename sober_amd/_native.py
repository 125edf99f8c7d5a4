"""ctypes binding of libsober_hip.so (the C ABI declared in include/sober_hip.h).

The product path has NO fallback: if the shared library is missing or an entry
point fails, an exception is raised.  torch is used only for device memory and
the current HIP stream (`tensor.data_ptr()`, `torch.cuda.current_stream()`).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SOBER_HIP_LIB: a diagnostic build of the same library, e.g. the in-kernel-stamp build `make stamps`)
LIB_PATH = os.environ.get("SOBER_HIP_LIB") or os.path.join(_HERE, "libsober_hip.so")
ABI_VERSION = 7

KIND_RBF, KIND_MATERN52, KIND_TANIMOTO = 0, 1, 2
KIND_BY_NAME = {"rbf": KIND_RBF, "matern52": KIND_MATERN52, "tanimoto": KIND_TANIMOTO}

_vp, _i32, _i64, _f64 = C.c_void_p, C.c_int, C.c_int64, C.c_double

# name -> (restype, argtypes); mirrors include/sober_hip.h one to one
SIGNATURES = {
    "sober_abi_version": (_i32, []),
    "sober_diag_build": (_i32, []),
    "sober_reload_switches": (_i32, []),
    "sober_level_job_size": (_i32, []),
    "sober_nystrom_job_size": (_i32, []),
    "sober_final_job_size": (_i32, []),
    "sober_set_i64": (_i32, [_vp, _i64, _vp]),
    "sober_rank_scatter": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp]),
    "sober_plan_rows": (_i32, [_i32, _vp, _i32, _i64, _vp, _i32, _i64, _i32, _vp, _i32, _f64, _vp, _i32, _vp, _i32, _vp,
                               _vp, _vp, _vp, _vp]),
    "sober_level_loop_final": (_i32, [_vp, _vp, _i64, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "sober_padded_dim": (_i32, [_i32]),
    "sober_bit_words": (_i32, [_i32]),
    "sober_scale_points": (_i32, [_vp, _i64, _i32, _i64, _vp, _i32, _vp, _i32, _vp]),
    "sober_augment_plan": (_i32, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp]),
    "sober_scale_points_idx": (_i32, [_vp, _vp, _i64, _i32, _i64, _vp, _i32, _vp, _i32, _vp, _vp, _vp]),
    "sober_pack_bits": (_i32, [_vp, _i64, _i32, _i64, _vp, _i32, _vp, _vp, _vp]),
    "sober_mt19937_uniform53": (_i32, [_vp, _i64, _i64, _vp]),
    "sober_box_muller": (_i32, [_vp, _i64, _vp, _vp]),
    "sober_pairwise": (_i32, [_i32, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _f64, _vp, _i64, _vp]),
    "sober_kernel_matvec": (_i32, [_i32, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i32, _f64, _f64, _vp, _vp]),
    "sober_level_reduce": (_i32, [_i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _i64, _i64, _i32, _vp, _vp,
                                  _f64, _i32, _vp, _i32, _i32, _vp, _i64, _vp]),
    "sober_nonzero_ws_bytes": (_i64, [_i64]),
    "sober_nonzero_i32": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "sober_dgemm_coldiv_t": (_i32, [_i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp]),
    "sober_level_chunks": (_i32, [_i32, _i64, _i64, _i32]),
    "sober_level_parts_mfma": (_i32, [_i32, _i64, _i64, _i32]),
    "sober_level_parts_mfma_cap": (_i32, [_i32, _i64, _i32]),
    "sober_aug_dim": (_i32, [_i32]),
    "sober_augment_points": (_i32, [_vp, _i64, _i32, _i64, _vp, _i32, _vp, _i32, _vp, _i32, _vp]),
    "sober_level_reduce_mfma": (_i32, [_i32, _vp, _i32, _vp, _i32, _vp, _i64, _i64, _i32, _vp, _vp, _f64, _i32,
                                       _vp, _i32, _i32, _vp, _i64, _vp]),
    "sober_sum_partials": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp]),
    "sober_dgemm": (_i32, [_i32, _i32, _i32, _i32, _i32, _f64, _vp, _i32, _vp, _i32, _f64, _vp, _i32, _vp]),
    "sober_barycentres": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "sober_level_update": (_i32, [_vp, _i64, _i64, _i32, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _vp]),
    "sober_scatter_weights": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "sober_i64_to_i32": (_i32, [_vp, _i64, _vp, _vp]),
    "sober_car_pivot_host": (_i32, [_vp, _i32, _i32, _vp]),
    "sober_car_pivot_host_fast": (_i32, [_vp, _i32, _i32, _vp]),
    "sober_car_supported": (_i32, [_i32, _i32]),
    "sober_car_ws_bytes": (_i64, [_i32, _i32]),
    "sober_car_device": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sober_probe_rcp": (_i32, [_vp, _vp, _i64, _vp]),
    "sober_second_elimination": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "sober_car_mc_supported": (_i32, [_i32, _i32]),
    "sober_car_mc_ws_bytes": (_i64, [_i32, _i32]),
    "sober_car_mc_device": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sober_car_big_supported": (_i32, [_i32, _i32]),
    "sober_car_big_ws_bytes": (_i64, [_i32, _i32]),
    "sober_car_big_device": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sober_car_device_ex": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "sober_obj_set_sums": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i64, _vp, _vp]),
    "sober_null_vector_supported": (_i32, [_i32]),
    "sober_null_vector": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp]),
    "sober_second_elimination_rows": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "sober_obj_set_sums_queued": (_i32, [_vp, _vp, _vp, _i32, _vp, _i64, _vp, _vp, _i32, _vp, _vp]),
    "sober_obj_job_size": (_i32, []),
    "sober_level_loop_obj": (_i32, [_vp, _vp, _i64, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "sober_car_safe_supported": (_i32, [_i32, _i32]),
    "sober_car_giveup_forced": (_i32, []),
    "sober_final_commit": (_i32, [_vp, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "sober_level_car_retry": (_i32, [_vp, _vp]),
    "sober_mc_selftest": (_i32, [_vp, _vp, _vp]),
    "sober_chol_max_n": (_i32, []),
    "sober_nystrom_max_n": (_i32, []),
    "sober_cholesky_probe_batched": (_i32, [_vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sober_cholesky": (_i32, [_vp, _i32, _i32, _f64, _vp, _vp, _vp]),
    "sober_cholesky_inv": (_i32, [_vp, _i32, _i32, _f64, _vp, _vp, _vp, _vp]),
    "sober_cholesky_inv_ratio": (_i32, [_vp, _i32, _i32, _f64, _vp, _vp, _vp, _vp, _vp]),
    "sober_trsm_blocks": (_i32, [_vp, _i64, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp]),
    "sober_cholesky_probe": (_i32, [_vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp]),
    "sober_cholesky_probe_piv": (_i32, [_vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "sober_cholesky_probe_mc_ws_bytes": (_i64, [_i32, _i32]),
    "sober_cholesky_probe_mc": (_i32, [_vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sober_abs_sym": (_i32, [_vp, _i32, _i32, _vp, _i32, _vp, _vp]),
    "sober_abs_sym_dmax": (_i32, [_vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp]),
    "sober_jitter_ladder": (_i32, [_vp, _i32, _i32, _i32, _vp]),
    "sober_jitter_ladder_auto": (_i32, [_vp, _i32, _i32, _vp, _i32, _vp, _vp]),
    "sober_kmeans_ws_bytes": (_i64, [_i64, _i32, _i32]),
    "sober_kmeans_stat_offset": (_i64, [_i64, _i32, _i32]),
    "sober_kmeans_ws_bytes_screened": (_i64, [_i64, _i32, _i32]),
    "sober_kmeans_lloyd": (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _vp]),
    "sober_predict_fused_supported": (_i32, [_i32, _i32, _i32]),
    "sober_predict_fused": (_i32, [_i32, _vp, _vp, _i32, _vp, _vp, _i64, _i32, _f64, _vp, _i32, _vp, _f64, _f64, _f64, _vp, _vp,
                                   _f64, _vp, _vp, _i32, _vp]),
    "sober_predict_fused_root": (_i32, [_i32, _vp, _vp, _i32, _vp, _vp, _i64, _i32, _f64, _vp, _i32, _vp, _vp, _f64, _f64, _f64, _vp,
                                        _vp, _f64, _vp, _vp, _i32, _vp]),
    "sober_predict_finish": (_i32, [_vp, _vp, _i32, _i64, _i64, _vp, _f64, _vp, _f64, _f64, _vp, _f64, _vp, _i32, _vp]),
    "sober_reduce_ws_bytes": (_i64, [_i64]),
    "sober_cleansing_weights": (_i32, [_vp, _i64, _f64, _vp, _i64, _vp]),
    "sober_level_gather": (_i32, [_vp, _i32, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _i64,
                                  _vp]),
    "sober_gspace_finish": (_i32, [_vp, _vp, _i64, _i32, _i64, _i64, _vp, _vp, _vp]),
    "sober_wkde_draw": (_i32, [_vp, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sober_level_moments": (_i32, [_vp, _vp]),
    "sober_level_car": (_i32, [_vp, _vp]),
    "sober_level_reduce_tani_supported": (_i32, [_i32]),
    "sober_level_reduce_tani": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _i64, _i64, _i32, _vp, _vp, _f64, _i32, _vp,
                                       _i32, _i32, _vp, _i64, _vp]),
    "sober_level_loop_sharded": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "sober_rccl_load": (_i32, [C.c_char_p]),
    "sober_rccl_unique_id": (_i32, [_vp]),
    "sober_rccl_comm_init": (_i32, [_vp, _i32, _i32, _vp]),
    "sober_rccl_comm_destroy": (_i32, [_vp]),
    "sober_rccl_allreduce_f64": (_i32, [_vp, _vp, _i64, _vp]),
    "sober_rccl_allreduce_ptr": (_i64, []),
    "sober_peer_region_bytes": (_i64, [_i64]),
    "sober_peer_create": (_i32, [_i32, _i32, _i64, _vp, _vp]),
    "sober_peer_connect": (_i32, [_vp, C.c_char_p]),
    "sober_peer_connect_ptrs": (_i32, [_vp, _vp]),
    "sober_peer_region": (_i64, [_vp]),
    "sober_peer_set_spin_limit": (_i32, [_vp, C.c_uint]),
    "sober_peer_allreduce_f64": (_i32, [_vp, _vp, _i64, _vp]),
    "sober_peer_allreduce_ptr": (_i64, []),
    "sober_peer_status": (_i32, [_vp, _vp, _i64, _vp]),
    "sober_peer_destroy": (_i32, [_vp]),
    "sober_level_reduce_mfma_queued": (_i32, [_i32, _vp, _i32, _vp, _i32, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _f64,
                                              _i32, _vp, _i32, _vp, _vp, _vp]),
    "sober_level_reduce_mfma_queued_pair": (_i32, [_i32, _vp, _i32, _vp, _i32, _vp, _i64, _i32, _i32, _vp, _vp, _f64, _i32,
                                                   _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "sober_sum_partials_queued": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp]),
    "sober_level_reduce_tani_queued": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _f64,
                                              _i32, _vp, _i32, _vp, _vp, _vp]),
    "sober_level_reduce_tani_queued_pair": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _vp, _vp, _f64, _i32,
                                                   _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "sober_level_chunks_cap": (_i32, [_i32, _i64, _i32]),
    "sober_level_chunks_tani": (_i32, [_i32, _i64, _i64, _i32]),
    "sober_level_chunks_tani_cap": (_i32, [_i32, _i64, _i32]),
    "sober_level_update_queued": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sober_level_update_queued_ex": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i64, _vp]),
    "sober_level_update_queued_cls": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "sober_level_class_wpt": (_i32, [_i32, _i64, _i32]),
    "sober_level_class_slots": (_i32, [_i32]),
    "sober_level_class_depth": (_i32, [_i32, _i32, _i64, _i32]),
    "sober_level_reduce_mfma_wpt": (_i32, [_i32, _vp, _i32, _vp, _i32, _vp, _i64, _i32, _vp, _vp, _f64, _i32, _i32, _vp, _i32,
                                           _vp, _vp]),
    "sober_class_sum": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "sober_class_derive_queued": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sober_record_event_pair": (_i32, [_vp, _vp, _vp]),
    "sober_set_launch_events": (_i32, [_vp, _vp]),
    "sober_projection": (_i32, [_vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "sober_nystrom_flags_bytes": (_i64, [_i32, _i32]),
    "sober_nystrom_basis": (_i32, [_vp, _i32, _vp]),
    "sober_gather_f64": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "sober_final_scatter": (_i32, [_vp, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "sober_level_final": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "sober_level_final_job": (_i32, [_vp, _vp, _vp, _i32, _vp]),
    "sober_level_loop": (_i32, [_vp, _i64, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
}

LEVEL_VALU, LEVEL_MFMA, LEVEL_GATHER, LEVEL_TANI = 0, 1, 2, 3
LEVEL_MAX_CHUNKS, LEVEL_XS, LEVEL_QUEUE = 64, 16, 24


class FinalJob(C.Structure):
    """struct sober_final_job of include/sober_hip.h (field for field)."""
    _fields_ = [
        ("rows_sc", _vp), ("rows_norm", _vp), ("cand_sc", _vp), ("cand_norm", _vp),
        ("dt", _i32), ("done", _i32), ("N", _i64), ("row_offset", _i64),
        ("K", _vp), ("mu_live", _vp), ("out_idx", _vp), ("out_w", _vp),
        ("cand_raw", _vp), ("ld_raw", _i64), ("d_raw", _i32), ("ls_len", _i32), ("ls", _vp), ("sc_buf", _vp),
    ]


class ObjJob(C.Structure):
    """struct sober_obj_job of include/sober_hip.h (field for field)."""
    _fields_ = [("obj", _vp), ("X_tmp", _vp), ("ocol", _vp), ("kr1", _vp), ("w1", _vp), ("nk1", _vp), ("null_row", _vp),
                ("status", _vp)]


class NystromJob(C.Structure):
    """struct sober_nystrom_job of include/sober_hip.h (field for field)."""
    _fields_ = [
        ("M", _i32), ("s", _i32), ("n_rungs", _i32), ("niter", _i32), ("probe_mc", _i32), ("_pad0", _i32),
        ("G", _vp), ("shifts", _vp), ("R", _vp), ("C", _vp), ("chol_work", _vp),
        ("probe_ws", _vp), ("probe_ws_bytes", _i64),
        ("Y", _vp * 2), ("Gm", _vp), ("xinv", _vp),
        ("flags_block", _vp), ("flags_bytes", _i64), ("h_flags_block", _vp),
        ("Ut", _vp),
        ("T", _vp), ("n_obs", _i32), ("mean_nys", _vp), ("P", _vp),
    ]


class LevelJob(C.Structure):
    """struct sober_level_job of include/sober_hip.h (field for field)."""
    _fields_ = [
        ("variant", _i32), ("kind", _i32),
        ("rows", _vp), ("rows_norm", _vp), ("cand", _vp), ("cand_norm", _vp),
        ("n_rows", _i32), ("dim", _i32), ("kmat_ld", _i64),
        ("wmul", _vp), ("outputscale", _f64),
        ("S", _i32), ("n", _i32), ("P", _vp),
        ("partG", _vp), ("partTot", _vp), ("extraG", _vp), ("extraTot", _vp),
        ("G", _vp), ("Xtr", _vp), ("tot", _vp), ("X_tmp", _vp),
        ("keep_rank", _vp), ("w_star", _vp), ("mu_out", _vp),
        ("car_ws", _vp), ("car_ws_bytes", _i64), ("h_flags", _vp),
        ("ev", _vp * 4),
        ("idx", _vp), ("pos0", _i64), ("count", _i64), ("E", _i64), ("mu", _vp),
        ("phase", _i32),
        ("dR", _vp), ("h_dR", _vp),
        ("car_mode", _i32),
        ("ev_used", C.c_uint64 * 2),
        ("class_depth", _i32), ("Gc", _vp * 2), ("totc", _vp * 2), ("cls_scale", _vp), ("cls_sof", _vp),
    ]

E_DIM = -2
E_EXCHANGE = -5
CLASS_MAX_DEPTH = 4                             # SOBER_CLASS_MAX_DEPTH
CAR_DEFAULT, CAR_SAFE, CAR_HOST = 0, 1, 2     # (CAR_HOST: the host side's own third rung, never passed to the library)

_lib: Optional[C.CDLL] = None


class SoberHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libsober_hip.so (built by sober_amd/csrc/Makefile or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SoberHipError(
            f"{LIB_PATH} not found: build it with `make -C sober_amd/csrc` "
            "(python -c 'import __graft_entry__ as g; g.build()').  sober_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.sober_abi_version.restype = _i32
    got = lib.sober_abi_version()                       # (first: a stale library says "rebuild", not AttributeError)
    if got != ABI_VERSION:
        raise SoberHipError(f"libsober_hip ABI {got} != expected {ABI_VERSION}; rebuild (make -C sober_amd/csrc)")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.sober_diag_build() and os.environ.get("SOBER_ALLOW_DIAG_LIB") != "1":
        raise SoberHipError(f"{LIB_PATH} is a diagnostic build (in-kernel time stamps): not a library to compute with; "
                            "`make -C sober_amd/csrc` builds the product (SOBER_ALLOW_DIAG_LIB=1: the stamp scripts)")
    for name, cls in (("sober_level_job_size", LevelJob), ("sober_nystrom_job_size", NystromJob),
                      ("sober_final_job_size", FinalJob), ("sober_obj_job_size", ObjJob)):
        if getattr(lib, name)() != C.sizeof(cls):
            raise SoberHipError(f"{cls.__name__}: {C.sizeof(cls)} bytes here, {getattr(lib, name)()} in libsober_hip; "
                                "include/sober_hip.h and sober_amd/_native.py disagree")
    _lib = lib
    return lib


def reload_switches():
    """Have the library read its environment switches again (they are read once, at load time)."""
    load().sober_reload_switches()


def _check(rc: int, what: str):
    if rc == 0:
        return
    if rc < 0:
        msg = {-1: "bad argument", -2: "dimension not supported by the compiled tile set",
               -3: "workspace too small",
               -5: "a multi-workgroup Caratheodory kernel gave up waiting for a partner workgroup (no result)"}.get(rc, "error")
        raise SoberHipError(f"{what}: {msg} (code {rc})")
    raise SoberHipError(f"{what}: hipError_t {rc}")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# the current stream's handle: torch.cuda.current_stream(dev).cuda_stream builds a Stream object per call (1.9 us of the
# 7 us a native call costs the host, ~150 calls per step); the raw getter underneath it is 0.2 us
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t: torch.Tensor) -> Optional[int]:
    if not t.is_cuda:
        raise SoberHipError("sober_amd runs on the MI355X only: tensor is not on a HIP device")
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


def _req(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype or not t.is_contiguous():
        raise SoberHipError(f"{name}: need contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


# --------------------------------------------------------------------------- #
# thin typed wrappers (torch tensors in, nothing allocated inside the library)
# --------------------------------------------------------------------------- #
def padded_dim(d: int, generic: bool = False) -> int:
    """Padded row length of scaled points.  The register-tiled kernels (level reduction, kernel-matvec) exist for
    d <= 32; `generic=True` returns the next multiple of 4 beyond that (sober_pairwise and sober_scale_points take
    any row length)."""
    r = load().sober_padded_dim(d)
    if r == E_DIM and generic and d > 0:
        return (d + 3) // 4 * 4
    _check(min(r, 0), f"padded_dim({d})")
    return r


def bit_words(d: int, generic: bool = False) -> int:
    """64-bit words per bit-packed fingerprint (tiled kernels: <= 32 words = 2048 bits; generic: any count)."""
    r = load().sober_bit_words(d)
    if r == E_DIM and generic and d > 0:
        return (d + 63) // 64
    _check(min(r, 0), f"bit_words({d})")
    return r


def fused_dim_supported(kind: int, d: int) -> bool:
    """True iff the fused level kernels have an instantiation for this input dimension."""
    r = load().sober_bit_words(d) if kind == KIND_TANIMOTO else load().sober_padded_dim(d)
    return r > 0


def scale_points_idx(X, idx, n, lengthscale, out):
    """out[i] = X[idx[i]] / lengthscale for i < n (zero padded to out's row length)."""
    _req(lengthscale, torch.float64, "lengthscale"); _req(out, torch.float64, "out"); _req(idx, torch.int32, "idx")
    if X.dtype != torch.float64 or X.stride(-1) != 1:
        raise SoberHipError("scale_points_idx: X must be float64 with unit inner stride")
    _check(load().sober_scale_points_idx(X.data_ptr(), idx.data_ptr(), int(n), X.shape[1], X.stride(0), lengthscale.data_ptr(),
                                         lengthscale.numel(), out.data_ptr(), out.shape[1], None, None, _stream(X)),
           "sober_scale_points_idx")


def scale_points(X, lengthscale, out):
    n, d = X.shape
    _req(X, torch.float64, "X"); _req(lengthscale, torch.float64, "lengthscale"); _req(out, torch.float64, "out")
    _check(load().sober_scale_points(X.data_ptr(), n, d, X.stride(0), lengthscale.data_ptr(),
                                     lengthscale.numel(), out.data_ptr(), out.shape[1], _stream(X)),
           "sober_scale_points")


def mt19937_uniform53(state, numel, out):
    """HOST: step the CPU generator state `state` (uint8 bytes of torch.get_rng_state(), in place) through the draws
    of torch.randn(numel, dtype=float64) and leave their uniforms in `out` (host float64, numel + 16)."""
    _req(state, torch.uint8, "state"); _req(out, torch.float64, "out")
    if state.is_cuda or out.is_cuda or out.numel() < numel + 16:
        raise ValueError("mt19937_uniform53 takes host tensors, out of numel + 16 doubles")
    _check(load().sober_mt19937_uniform53(state.data_ptr(), state.numel(), numel, out.data_ptr()),
           "sober_mt19937_uniform53")


def box_muller(u, numel, out):
    _req(u, torch.float64, "u"); _req(out, torch.float64, "out")
    if u.numel() < numel + 16 or out.numel() < numel:
        raise ValueError("box_muller: u holds numel + 16 uniforms, out numel normals")
    _check(load().sober_box_muller(u.data_ptr(), numel, out.data_ptr(), _stream(out)), "sober_box_muller")


def pack_bits(X, words, norms, bad_flag):
    n, d = X.shape
    _req(X, torch.float64, "X"); _req(words, torch.int64, "words"); _req(norms, torch.float64, "norms")
    _req(bad_flag, torch.int32, "bad_flag")
    _check(load().sober_pack_bits(X.data_ptr(), n, d, X.stride(0), words.data_ptr(), words.shape[1],
                                  norms.data_ptr(), bad_flag.data_ptr(), _stream(X)), "sober_pack_bits")


def rank_scatter(X, n, ocol, rank, n1, Xp, objp):
    """sober_rank_scatter: rows X[s, :n] and ocol[s] of the sets with rank 0 .. n1-1 to their rank's place."""
    _req(X, torch.float64, "X"); _req(ocol, torch.float64, "ocol"); _req(rank, torch.int32, "rank")
    _check(load().sober_rank_scatter(X.data_ptr(), X.stride(0), X.shape[0], n, ocol.data_ptr(), rank.data_ptr(), n1,
                                     Xp.data_ptr(), objp.data_ptr(), _stream(X)), "sober_rank_scatter")


def plan_rows(kind, X_nys, X_obs, lengthscale, outputscale, S_cache, rows, Kall, W, T, G):
    """sober_plan_rows: the row table, Kall, W, T and the Gram matrix of a step in one call (X_obs None: mode "kernel")."""
    M, d = X_nys.shape
    n_obs = 0 if X_obs is None else X_obs.shape[0]
    for t in (X_nys, X_obs, S_cache, rows, Kall, W, T, G):
        if t is not None and (t.dtype != torch.float64 or t.stride(-1) != 1):
            raise SoberHipError("plan_rows: float64 tensors with unit inner stride")
    _check(load().sober_plan_rows(kind, X_nys.data_ptr(), M, X_nys.stride(0), _ptr(X_obs), n_obs,
                                  X_obs.stride(0) if X_obs is not None else 0, d, lengthscale.data_ptr(),
                                  lengthscale.numel(), float(outputscale), _ptr(S_cache),
                                  S_cache.stride(0) if S_cache is not None else 0, rows.data_ptr(), rows.shape[1],
                                  _ptr(Kall), _ptr(W), _ptr(T), G.data_ptr(), _stream(G)), "sober_plan_rows")


def pairwise(kind, a, a_norm, b, b_norm, idx, n, dt, outputscale, out):
    _check(load().sober_pairwise(kind, a.data_ptr(), _ptr(a_norm), a.shape[0], b.data_ptr(), _ptr(b_norm),
                                 _ptr(idx), n, dt, float(outputscale), out.data_ptr(), out.stride(0),
                                 _stream(out)), "sober_pairwise")


def kernel_matvec(kind, a, a_norm, v, b, b_norm, dt, outputscale, c0, out):
    _check(load().sober_kernel_matvec(kind, a.data_ptr(), _ptr(a_norm), v.data_ptr(), a.shape[0],
                                      b.data_ptr(), _ptr(b_norm), b.shape[0], dt, float(outputscale),
                                      float(c0), out.data_ptr(), _stream(out)), "sober_kernel_matvec")


def nonzero_ws_bytes(N) -> int:
    r = load().sober_nonzero_ws_bytes(N)
    _check(min(int(r), 0), "sober_nonzero_ws_bytes")
    return int(r)


def nonzero_i32(mu, idx_out, count_out, ws):
    _req(mu, torch.float64, "mu"); _req(idx_out, torch.int32, "idx_out"); _req(count_out, torch.int64, "count_out")
    _check(load().sober_nonzero_i32(mu.data_ptr(), mu.numel(), idx_out.data_ptr(), count_out.data_ptr(), ws.data_ptr(),
                                    ws.numel() * ws.element_size(), _stream(mu)), "sober_nonzero_i32")


def level_chunks(n_rows, pos0, count, S) -> int:
    r = load().sober_level_chunks(n_rows, pos0, count, S)
    _check(min(r, 0), "sober_level_chunks")
    return r


def level_parts_mfma(n_rows, pos0, count, S) -> int:
    r = load().sober_level_parts_mfma(n_rows, pos0, count, S)
    _check(min(r, 0), "sober_level_parts_mfma")
    return r


def level_reduce(kind, rows, rows_norm, cand, cand_norm, dt, idx, idx_off, pos0, count, S, mu, wmul,
                 outputscale, n_chunks, partG, ldg, col0, partTot, tot_limit):
    _req(idx, torch.int32, "idx"); _req(mu, torch.float64, "mu")
    _check(load().sober_level_reduce(kind, rows.data_ptr(), _ptr(rows_norm), rows.shape[0], cand.data_ptr(),
                                     _ptr(cand_norm), dt, idx.data_ptr() + 4 * idx_off, pos0, count, S,
                                     mu.data_ptr(), _ptr(wmul), float(outputscale), n_chunks,
                                     partG.data_ptr(), ldg, col0, _ptr(partTot), tot_limit, _stream(mu)),
           "sober_level_reduce")


def aug_dim(d: int) -> int:
    """DA for the matrix-core level kernel, or -1 when d + 2 exceeds the compiled tile set."""
    r = load().sober_aug_dim(d)
    return r if r > 0 else -1


def augment_plan(X_nys, X_obs, X_cand, lengthscale, center, rows_aug, cand_aug):
    """sober_augment_plan: center <- column means of X_nys; rows_aug <- [X_nys; X_obs] (side 0), cand_aug <- X_cand (side 1)."""
    for t_, nm in ((X_nys, "X_nys"), (X_cand, "X_cand"), (X_obs, "X_obs")):
        if t_ is not None and (t_.dtype != torch.float64 or t_.stride(-1) != 1):
            raise SoberHipError(f"augment_plan: {nm} must be float64 with unit inner stride")
    _req(rows_aug, torch.float64, "rows_aug"); _req(cand_aug, torch.float64, "cand_aug"); _req(center, torch.float64, "center")
    n_obs = 0 if X_obs is None else X_obs.shape[0]
    _check(load().sober_augment_plan(X_nys.data_ptr(), X_nys.shape[0], X_nys.stride(0), _ptr(X_obs), n_obs,
                                     X_obs.stride(0) if n_obs else 0, X_cand.data_ptr(), X_cand.shape[0], X_cand.stride(0),
                                     X_nys.shape[1], lengthscale.data_ptr(), lengthscale.numel(), center.data_ptr(),
                                     rows_aug.data_ptr(), cand_aug.data_ptr(), rows_aug.shape[1], _stream(X_cand)),
           "sober_augment_plan")


def augment_points(X, lengthscale, center, side, out):
    n, d = X.shape
    _req(X, torch.float64, "X"); _req(out, torch.float64, "out"); _req(center, torch.float64, "center")
    _check(load().sober_augment_points(X.data_ptr(), n, d, X.stride(0), lengthscale.data_ptr(),
                                       lengthscale.numel(), center.data_ptr(), side, out.data_ptr(),
                                       out.shape[1], _stream(X)), "sober_augment_points")


def level_reduce_mfma(kind, rows, cand, da, idx, idx_off, pos0, count, S, mu, wmul, outputscale, n_chunks,
                      partG, ldg, col0, partTot, tot_limit):
    _req(idx, torch.int32, "idx"); _req(mu, torch.float64, "mu")
    _check(load().sober_level_reduce_mfma(kind, rows.data_ptr(), rows.shape[0], cand.data_ptr(), da,
                                          idx.data_ptr() + 4 * idx_off, pos0, count, S, mu.data_ptr(),
                                          _ptr(wmul), float(outputscale), n_chunks, partG.data_ptr(), ldg,
                                          col0, _ptr(partTot), tot_limit, _stream(mu)),
           "sober_level_reduce_mfma")


def sum_partials(partG, partTot, n_chunks, n_rows, ldg, S, extraG, extraTot, n_xchunks, n_xcols, G, tot):
    _check(load().sober_sum_partials(partG.data_ptr(), _ptr(partTot), n_chunks, n_rows, ldg, S,
                                     _ptr(extraG), _ptr(extraTot), n_xchunks, n_xcols, G.data_ptr(),
                                     G.stride(0), _ptr(tot), _stream(G)), "sober_sum_partials")


def dgemm(A, B, C_, transa=False, transb=False, alpha=1.0, beta=0.0):
    """C_[m,n] = alpha * op(A) @ op(B) + beta * C_ ; row-major 2-D tensors (unit inner stride)."""
    m, n = C_.shape
    k = A.shape[0] if transa else A.shape[1]
    for t, nm in ((A, "A"), (B, "B"), (C_, "C")):
        if t.dtype != torch.float64 or t.stride(1) != 1:
            raise SoberHipError(f"dgemm {nm}: need float64 with unit inner stride")
    ka = B.shape[1] if transb else B.shape[0]
    ma = A.shape[1] if transa else A.shape[0]
    na = B.shape[0] if transb else B.shape[1]
    if (ma, na, ka) != (m, n, k):
        raise SoberHipError(f"dgemm shape mismatch {A.shape} {B.shape} -> {C_.shape}")
    _check(load().sober_dgemm(int(transa), int(transb), m, n, k, float(alpha), A.data_ptr(), A.stride(0),
                              B.data_ptr(), B.stride(0), float(beta), C_.data_ptr(), C_.stride(0),
                              _stream(C_)), "sober_dgemm")


def barycentres(Xtr, n, S, tot, X_tmp):
    _check(load().sober_barycentres(Xtr.data_ptr(), Xtr.stride(0), n, S, _ptr(tot), X_tmp.data_ptr(),
                                    _stream(X_tmp)), "sober_barycentres")


def level_update(idx_cur, idx_off, pos0, count, S, E, keep_rank, w_star, tot, n_keep, mu, idx_new, new_pos0):
    _check(load().sober_level_update(idx_cur.data_ptr() + 4 * idx_off, pos0, count, S, E,
                                     keep_rank.data_ptr(), w_star.data_ptr(), tot.data_ptr(), n_keep,
                                     mu.data_ptr(), idx_new.data_ptr(), new_pos0, _stream(mu)),
           "sober_level_update")


def scatter_weights(idx_cur, sel, w, n_sel, mu, out_idx):
    _check(load().sober_scatter_weights(idx_cur.data_ptr(), sel.data_ptr(), w.data_ptr(), n_sel,
                                        mu.data_ptr(), out_idx.data_ptr(), _stream(mu)),
           "sober_scatter_weights")


def i64_to_i32(src, dst):
    _check(load().sober_i64_to_i32(src.data_ptr(), src.numel(), dst.data_ptr(), _stream(src)),
           "sober_i64_to_i32")


def car_pivot_host(Phi: torch.Tensor, mu: torch.Tensor, fast=None) -> int:
    """Host tensors (CPU, float64, contiguous).  Phi (N, m) is destroyed, mu updated in place."""
    if Phi.is_cuda or mu.is_cuda:
        raise SoberHipError("car_pivot_host takes host tensors")
    _req(Phi, torch.float64, "Phi"); _req(mu, torch.float64, "mu")
    N, m = Phi.shape
    fn = load().sober_car_pivot_host_fast if (fast or (fast is None and N > 200)) else load().sober_car_pivot_host
    r = fn(Phi.data_ptr(), N, m, mu.data_ptr())
    _check(min(r, 0), "sober_car_pivot_host")
    return r


_CAR_WS = {}      # per-device scratch for k_car's reflector vectors (caller-owned, reused)


def car_supported(N: int, m: int) -> bool:
    return bool(load().sober_car_supported(N, m))


def car_safe_supported(N: int, m: int) -> bool:
    return bool(load().sober_car_safe_supported(N, m))


def probe_rcp(x):
    """v_rcp_f64 of a float64 device tensor (test hook: the screened ratio test's seed)."""
    out = torch.empty_like(x)
    _check(load().sober_probe_rcp(x.data_ptr(), out.data_ptr(), x.numel(), _stream(x)), "sober_probe_rcp")
    return out


def car_device(X, mu_in, keep_rank, w_star, n_keep, mu_out, phi_out=None, multi_cu=False, mode=CAR_DEFAULT, big=False):
    """X (N, m-1) float64 device (unit inner stride), mu_in (N).  multi_cu=True: the multi-CU kernels of
    csrc/car_mc.hip whatever the size; big=True: the memory-resident kernels of csrc/car_big.hip whatever the size (test and
    timing hooks; sober_car_device picks by size).  mode: CAR_DEFAULT or CAR_SAFE (sober_car_device_ex: only launches that
    cannot give up)."""
    N, n = X.shape
    lib = load()
    if mode != CAR_DEFAULT and not multi_cu and not big:
        nbytes = lib.sober_car_ws_bytes(N, n + 1)
        ws = _CAR_WS.get(X.device)
        if ws is None or ws.numel() * 8 < nbytes:
            ws = _CAR_WS[X.device] = torch.empty(max(nbytes // 8, 1), dtype=torch.float64, device=X.device)
        _check(lib.sober_car_device_ex(X.data_ptr(), X.stride(0), N, n + 1, mu_in.data_ptr(), keep_rank.data_ptr(),
                                       w_star.data_ptr(), n_keep.data_ptr(), mu_out.data_ptr(), _ptr(phi_out),
                                       ws.data_ptr(), nbytes, int(mode), _stream(X)), "sober_car_device_ex")
        return
    nbytes = max(lib.sober_car_ws_bytes(N, n + 1), lib.sober_car_mc_ws_bytes(N, n + 1) if multi_cu else 0,
                 lib.sober_car_big_ws_bytes(N, n + 1) if big else 0)
    ws = _CAR_WS.get(X.device)
    if ws is None or ws.numel() * 8 < nbytes:
        ws = torch.empty(max(nbytes // 8, 1), dtype=torch.float64, device=X.device)
        _CAR_WS[X.device] = ws
    fn, name = (lib.sober_car_big_device, "sober_car_big_device") if big else \
        (lib.sober_car_mc_device, "sober_car_mc_device") if multi_cu else (lib.sober_car_device, "sober_car_device")
    _check(fn(X.data_ptr(), X.stride(0), N, n + 1, mu_in.data_ptr(), keep_rank.data_ptr(),
              w_star.data_ptr(), n_keep.data_ptr(), mu_out.data_ptr(), _ptr(phi_out),
              ws.data_ptr(), nbytes, _stream(X)), name)


def obj_set_sums(obj, mu, idx, pos0, count, S, E, out):
    _req(obj, torch.float64, "obj"); _req(mu, torch.float64, "mu"); _req(idx, torch.int32, "idx"); _req(out, torch.float64, "out")
    _check(load().sober_obj_set_sums(obj.data_ptr(), mu.data_ptr(), idx.data_ptr(), int(pos0), int(count), int(S), int(E),
                                     out.data_ptr(), _stream(mu)), "sober_obj_set_sums")


def null_vector(X, nfun, rank1, n_keep1, n1, null_row, status):
    """sober_null_vector: the null vector of [X[survivors, :nfun]^T; 1] by set (csrc/null_vector.hip)."""
    _check(load().sober_null_vector(X.data_ptr(), X.stride(0), X.shape[0], int(nfun), rank1.data_ptr(), n_keep1.data_ptr(),
                                    int(n1), null_row.data_ptr(), status.data_ptr(), _stream(X)), "sober_null_vector")


def second_elimination_rows(null_row, obj_row, w1, rank1, n_keep1, n1, keep_rank, w_star, n_keep, status=None):
    _check(load().sober_second_elimination_rows(null_row.data_ptr(), obj_row.data_ptr(), w1.data_ptr(), rank1.data_ptr(),
                                                n_keep1.data_ptr(), _ptr(status), int(n1), rank1.numel(), keep_rank.data_ptr(),
                                                w_star.data_ptr(), n_keep.data_ptr(), _stream(null_row)),
           "sober_second_elimination_rows")


def level_loop_obj(job, obj_job, R, idx_a, idx_b, first_sums_ready, stream):
    """sober_level_loop_obj: the queued chain of the acquisition-guided branch -> (level_R list, R_final, list is idx_b?)."""
    level_R = (_i64 * MAX_LEVELS)()
    n_levels, in_b, R_final = _i32(0), _i32(0), _i64(0)
    _check(load().sober_level_loop_obj(C.addressof(job), C.addressof(obj_job), R, idx_a.data_ptr(), idx_b.data_ptr(),
                                       int(bool(first_sums_ready)), MAX_LEVELS, level_R, C.byref(n_levels), C.byref(R_final),
                                       C.byref(in_b), stream), "sober_level_loop_obj")
    return list(level_R[:n_levels.value]), int(R_final.value), bool(in_b.value)


def second_elimination(phi, objp, w1, rank1, n1, keep_rank, w_star, n_keep):
    _check(load().sober_second_elimination(phi.data_ptr(), objp.data_ptr(), w1.data_ptr(), rank1.data_ptr(), int(n1),
                                           rank1.numel(), keep_rank.data_ptr(), w_star.data_ptr(), n_keep.data_ptr(),
                                           _stream(phi)), "sober_second_elimination")


def mc_selftest(x: torch.Tensor) -> torch.Tensor:
    out = torch.empty(128, dtype=torch.float64, device=x.device)
    _check(load().sober_mc_selftest(x.data_ptr(), out.data_ptr(), _stream(x)), "sober_mc_selftest")
    return out


def chol_max_n() -> int:
    return load().sober_chol_max_n()


def nystrom_max_n() -> int:
    """Largest N_nys of the device Nystrom route (the ladder's probes go panel by panel beyond chol_max_n())."""
    return load().sober_nystrom_max_n()


def cholesky(A, shift, info, min_pivot=None):
    """In place on the lower triangle of the square device matrix A (unit inner stride)."""
    n = A.shape[0]
    _check(load().sober_cholesky(A.data_ptr(), n, A.stride(0), float(shift), info.data_ptr(), _ptr(min_pivot),
                                 _stream(A)), "sober_cholesky")


def cholesky_inv(A, shift, info, min_pivot, xinv, ratio=None):
    """cholesky() that also leaves the inverted 32 x 32 diagonal blocks of L in xinv (ceil(n/32) * 1024 doubles) and,
    with `ratio`, smallest pivot / largest diagonal entry of the input there."""
    n = A.shape[0]
    if xinv.numel() < ((n + 31) // 32) * 1024:
        raise SoberHipError("cholesky_inv: xinv too small")
    _check(load().sober_cholesky_inv_ratio(A.data_ptr(), n, A.stride(0), float(shift), info.data_ptr(), _ptr(min_pivot),
                                           xinv.data_ptr(), _ptr(ratio), _stream(A)), "sober_cholesky_inv")


def trsm_blocks(Y, L, xinv, Q):
    m, q = Y.shape
    _check(load().sober_trsm_blocks(Y.data_ptr(), m, q, Y.stride(0), L.data_ptr(), L.stride(0), xinv.data_ptr(),
                                    Q.data_ptr(), Q.stride(0), _stream(Y)), "sober_trsm_blocks")


def cholesky_probe(src, shifts, work, info, min_pivot=None):
    n = src.shape[0]
    _check(load().sober_cholesky_probe_piv(src.data_ptr(), n, src.stride(0), shifts.data_ptr(), shifts.numel(),
                                           work.data_ptr(), info.data_ptr(), _ptr(min_pivot), _stream(src)),
           "sober_cholesky_probe_piv")


PROBE_NO_VERDICT = -7      # info of a rung whose workgroups lost each other in cholesky_probe_mc


def cholesky_probe_mc_ws_bytes(n: int, n_shifts: int) -> int:
    return int(load().sober_cholesky_probe_mc_ws_bytes(n, n_shifts))


def cholesky_probe_mc(src, shifts, work, info, min_pivot, ws):
    """cholesky_probe with 8 workgroups per rung; info == PROBE_NO_VERDICT for a rung means: probe again with
    cholesky_probe (or decide on the host)."""
    n = src.shape[0]
    _check(load().sober_cholesky_probe_mc(src.data_ptr(), n, src.stride(0), shifts.data_ptr(), shifts.numel(),
                                          work.data_ptr(), info.data_ptr(), _ptr(min_pivot), ws.data_ptr(),
                                          ws.numel() * ws.element_size(), _stream(src)), "sober_cholesky_probe_mc")


def abs_sym(C_, out, flag, dmax=None):
    """|C| = sqrt(C * C^T) elementwise + the symmetry flag; dmax (one zeroed device double): max diagonal of the result."""
    n = C_.shape[0]
    _check(load().sober_abs_sym_dmax(C_.data_ptr(), n, C_.stride(0), out.data_ptr(), out.stride(0), flag.data_ptr(),
                                     _ptr(dmax), _stream(C_)), "sober_abs_sym")


def kmeans_lloyd(X, K, iters, centroids, labels):
    N, d = X.shape
    _req(X, torch.float64, "X"); _req(centroids, torch.float64, "centroids"); _req(labels, torch.int32, "labels")
    nbytes = int(load().sober_kmeans_ws_bytes(N, d, K))
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=X.device)     # cluster sizes, sorted order, sort scratch
    _check(load().sober_kmeans_lloyd(X.data_ptr(), N, d, K, iters, centroids.data_ptr(), labels.data_ptr(),
                                     ws.data_ptr(), nbytes, _stream(X)), "sober_kmeans_lloyd")


def predict_fused_supported(kind: int, n_obs: int, dt: int) -> bool:
    return bool(load().sober_predict_fused_supported(int(kind), int(n_obs), int(dt)))


def predict_fused(kind, obs, obs_norm, cand, cand_norm, n, dt, outputscale, W, alpha, c0, kxx_const, noise, mean_out,
                  var_out, eta=0.0, lfi_out=None, log=False, eta_dev=None, root_tri=None):
    """csrc/predict.hip: mean / variance / pi over `n` prepared candidates in one launch (W symmetric); eta_dev: the
    threshold as a one-element device tensor (read by the kernel: no host read-back).  root_tri (a one-element int32 device
    tensor): `W` is then S^T, the transposed root of W = S S^T, and the word says whether it is lower triangular
    (sober_predict_fused_root)."""
    # (the kernel takes raw pointers: only W's row stride travels -- everything else must be what it assumes)
    if W.dtype != torch.float64 or W.dim() != 2 or W.stride(1) != 1:
        raise SoberHipError(f"predict_fused: W must be float64 with unit inner stride, got {W.dtype} strides {tuple(W.stride())}")
    if cand.dtype == torch.float64 and (cand.dim() != 2 or cand.stride(1) != 1 or cand.stride(0) != dt):
        raise SoberHipError(f"predict_fused: candidates must be rows of {dt} contiguous float64, got strides {tuple(cand.stride())}")
    for t_, nm in ((alpha, "alpha"), (mean_out, "mean_out"), (var_out, "var_out"), (lfi_out, "lfi_out"), (eta_dev, "eta_dev")):
        if t_ is not None:
            _req(t_, torch.float64, "predict_fused: " + nm)
    if alpha is not None and alpha.numel() != obs.shape[0]:
        raise SoberHipError(f"predict_fused: alpha has {alpha.numel()} entries for {obs.shape[0]} observations")
    if root_tri is not None:
        _req(root_tri, torch.int32, "predict_fused: root_tri")
        if W.shape[0] != W.shape[1] or W.shape[0] != obs.shape[0]:
            raise SoberHipError(f"predict_fused: the root must be {obs.shape[0]} x {obs.shape[0]}, got {tuple(W.shape)}")
        _check(load().sober_predict_fused_root(int(kind), obs.data_ptr(), _ptr(obs_norm), obs.shape[0], cand.data_ptr(),
                                               _ptr(cand_norm), int(n), int(dt), float(outputscale), W.data_ptr(), W.stride(0),
                                               root_tri.data_ptr(), _ptr(alpha), float(c0), float(kxx_const), float(noise),
                                               _ptr(mean_out), var_out.data_ptr(), float(eta), _ptr(eta_dev), _ptr(lfi_out),
                                               int(bool(log)), _stream(cand)), "sober_predict_fused_root")
        return
    _check(load().sober_predict_fused(int(kind), obs.data_ptr(), _ptr(obs_norm), obs.shape[0], cand.data_ptr(),
                                      _ptr(cand_norm), int(n), int(dt), float(outputscale), W.data_ptr(), W.stride(0),
                                      _ptr(alpha), float(c0), float(kxx_const), float(noise), _ptr(mean_out),
                                      var_out.data_ptr(), float(eta), _ptr(eta_dev), _ptr(lfi_out), int(bool(log)),
                                      _stream(cand)),
           "sober_predict_fused")


def predict_finish(KX, V, mean, kxx_const, norms, outputscale, noise, var_out, eta=0.0, lfi_out=None, log=False):
    n_obs, N = KX.shape
    _check(load().sober_predict_finish(KX.data_ptr(), V.data_ptr(), n_obs, N, KX.stride(0), _ptr(mean),
                                       float(kxx_const), _ptr(norms), float(outputscale), float(noise),
                                       var_out.data_ptr(), float(eta), _ptr(lfi_out), int(bool(log)),
                                       _stream(KX)), "sober_predict_finish")


def cleansing_weights(w, eps):
    _req(w, torch.float64, "weights")
    nbytes = load().sober_reduce_ws_bytes(w.numel())
    ws = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=w.device)
    _check(load().sober_cleansing_weights(w.data_ptr(), w.numel(), float(eps), ws.data_ptr(), nbytes,
                                          _stream(w)), "sober_cleansing_weights")


def wkde_draw(eps, comp, Xobs, L, lo, hi, x, inside):
    n, d = eps.shape
    _check(load().sober_wkde_draw(eps.data_ptr(), n, d, comp.data_ptr(), Xobs.data_ptr(), Xobs.stride(0),
                                  L.data_ptr(), lo.data_ptr() if lo is not None else None,
                                  hi.data_ptr() if hi is not None else None, x.data_ptr(),
                                  inside.data_ptr() if inside is not None else None, _stream(eps)),
           "sober_wkde_draw")


def level_gather(Kmat, idx, idx_off, pos0, count, S, mu, wmul, n_chunks, partG, ldg, col0, partTot, tot_limit):
    _req(idx, torch.int32, "idx"); _req(mu, torch.float64, "mu"); _req(Kmat, torch.float64, "Kmat")
    if Kmat.stride(1) != 1:
        raise SoberHipError("level_gather: Kmat needs unit inner stride")
    _check(load().sober_level_gather(Kmat.data_ptr(), Kmat.shape[1], Kmat.stride(0),
                                     idx.data_ptr() + 4 * idx_off, pos0, count, S, mu.data_ptr(), _ptr(wmul),
                                     n_chunks, partG.data_ptr(), ldg, col0, _ptr(partTot), tot_limit,
                                     _stream(partG)), "sober_level_gather")


def gspace_finish(K, corr, mug_cand, mug_rows):
    n, m = K.shape
    _check(load().sober_gspace_finish(K.data_ptr(), corr.data_ptr(), n, m, K.stride(0), corr.stride(0),
                                      mug_cand.data_ptr(), mug_rows.data_ptr(), _stream(K)), "sober_gspace_finish")


def car_ws_bytes(N: int, m: int) -> int:
    return load().sober_car_ws_bytes(N, m)


def level_moments(job: LevelJob, stream: int):
    _check(load().sober_level_moments(C.addressof(job), stream), "sober_level_moments")


def level_car(job: LevelJob, stream: int):
    _check(load().sober_level_car(C.addressof(job), stream), "sober_level_car")


def level_car_retry(job: LevelJob, stream: int) -> bool:
    """sober_level_car_retry: the step of the last sober_level_car again in CAR_SAFE mode (synchronised).  False: that
    mode does not cover the size."""
    rc = load().sober_level_car_retry(C.addressof(job), stream)
    if rc == E_EXCHANGE:
        return False
    _check(rc, "sober_level_car_retry")
    return True


E_NOPROGRESS = -4
MAX_LEVELS = 64


def level_loop(job: LevelJob, R: int, idx_a, idx_b, first_sums_ready: bool, events, stream: int, final: "FinalJob" = None):
    """sober_level_loop (sober_level_loop_final with `final`: the final direct level follows in the same call when the
    loop ends on one, final.done says so): -> (level_R list, R_final, final list is idx_b?, gave up?).  events: None
    or a flat list of 4 * MAX_LEVELS hipEvent_t handles (None entries allowed)."""
    level_R = (_i64 * MAX_LEVELS)()
    n_levels, in_b, R_final = _i32(0), _i32(0), _i64(0)
    ev = None
    if events is not None:
        ev = (_vp * (4 * MAX_LEVELS))(*events)
    if final is not None:
        rc = load().sober_level_loop_final(C.addressof(job), C.addressof(final), R, idx_a.data_ptr(), idx_b.data_ptr(),
                                           int(bool(first_sums_ready)), ev, MAX_LEVELS, level_R, C.byref(n_levels),
                                           C.byref(R_final), C.byref(in_b), stream)
    else:
        rc = load().sober_level_loop(C.addressof(job), R, idx_a.data_ptr(), idx_b.data_ptr(), int(bool(first_sums_ready)),
                                     ev, MAX_LEVELS, level_R, C.byref(n_levels), C.byref(R_final), C.byref(in_b), stream)
    if rc == E_NOPROGRESS:
        raise RuntimeError("recombination made no progress (the Caratheodory step cancelled nothing, "
                           "SOBER/_rchq.py:241-242); the reference would loop forever here")
    if rc != E_EXCHANGE:
        _check(rc, "sober_level_loop")
    # (E_EXCHANGE: the level at R_final is still due -- its Caratheodory step gave up and the safe launches do not
    #  cover its size; the state handed back is consistent and the caller goes on from there on the host route)
    return list(level_R[:n_levels.value]), int(R_final.value), bool(in_b.value), rc == E_EXCHANGE


ALLREDUCE_FN = C.CFUNCTYPE(_i32, _vp, _vp, _i64, _vp)      # sober_allreduce_fn(comm, buf, n, stream)


def level_loop_sharded(job: LevelJob, rank: int, world: int, bounds, idx_a, idx_b, first_sums_ready: bool,
                       allreduce_ptr, comm_ptr, R_stop: int, stream: int):
    """sober_level_loop_sharded: -> (level_R list, new bounds list, final list is idx_b?).  allreduce_ptr: address of
    a sober_allreduce_fn (sober_rccl_allreduce_ptr() or a ctypes callback cast to void*)."""
    level_R = (_i64 * MAX_LEVELS)()
    n_levels, in_b = _i32(0), _i32(0)
    b = (_i64 * (world + 1))(*bounds)
    rc = load().sober_level_loop_sharded(C.addressof(job), rank, world, b, idx_a.data_ptr(), idx_b.data_ptr(),
                                         int(bool(first_sums_ready)), allreduce_ptr, comm_ptr, int(R_stop), MAX_LEVELS,
                                         level_R, C.byref(n_levels), C.byref(in_b), stream)
    if rc == E_NOPROGRESS:
        raise RuntimeError("recombination made no progress (the Caratheodory step cancelled nothing, "
                           "SOBER/_rchq.py:241-242); the reference would loop forever here")
    _check(rc, "sober_level_loop_sharded")
    return list(level_R[:n_levels.value]), list(b), bool(in_b.value)


class RcclComm:
    """An RCCL communicator of our own over the ranks of a torch.distributed group (csrc/rccl_link.cpp): rank 0
    draws the unique id, the group's own collective carries it to the others."""

    def __init__(self, dist, group, device):
        import torch as _t
        lib = load()
        path = os.path.join(os.path.dirname(_t.__file__), "lib", "librccl.so")
        _check(lib.sober_rccl_load(path.encode() if os.path.exists(path) else b""), "sober_rccl_load")
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        buf = (C.c_char * 128)()
        if rank == 0:
            _check(lib.sober_rccl_unique_id(C.addressof(buf)), "sober_rccl_unique_id")
        on_dev = dist.get_backend(group) == "nccl"
        t = _t.frombuffer(bytearray(buf.raw), dtype=_t.uint8).clone()
        t = t.to(device) if on_dev else t
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ident = bytes(t.cpu().numpy().tobytes())
        self.handle = _vp()
        with _t.cuda.device(device):
            _check(lib.sober_rccl_comm_init(ident, rank, world, C.byref(self.handle)), "sober_rccl_comm_init")
        self.fn_ptr = lib.sober_rccl_allreduce_ptr()

    def close(self):
        if self.handle:
            load().sober_rccl_comm_destroy(self.handle)
            self.handle = _vp()


class PeerComm:
    """The one-shot direct-peer all-reduce of csrc/peer_reduce.hip over the ranks of a torch.distributed group (one
    process per GPU of ONE node): every rank creates its exchange region, the group's own all-gather carries the IPC
    handles, every rank maps the others' regions.  `self_check` runs a few calls with known data on every rank and
    lets the group agree on the verdict -- a node where peer mapping or cross-process visibility does not work ends
    up on the RCCL route instead."""

    def __init__(self, dist, group, device, n_max: int):
        import torch as _t
        lib = load()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.n_max, self.device, self.handle = int(n_max), device, _vp()
        self.fn_ptr = lib.sober_peer_allreduce_ptr()
        on_dev = dist.get_backend(group) == "nccl"
        hbuf = (C.c_char * 64)()
        with _t.cuda.device(device):
            rc = lib.sober_peer_create(self.rank, self.world, self.n_max, C.byref(self.handle), C.addressof(hbuf))
        mine = _t.frombuffer(bytearray(hbuf.raw), dtype=_t.uint8).clone()
        mine = _t.cat([mine, _t.tensor([0 if rc == 0 else 1], dtype=_t.uint8)])        # handle + "my create failed"
        mine = mine.to(device) if on_dev else mine
        outs = [_t.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(outs, mine, group=group)
        outs = [o.cpu() for o in outs]
        ok = rc == 0 and all(int(o[64]) == 0 for o in outs)
        if ok:
            allh = b"".join(bytes(o[:64].numpy().tobytes()) for o in outs)
            with _t.cuda.device(device):
                ok = lib.sober_peer_connect(self.handle, allh) == 0
        self.ok = self._agree(dist, group, on_dev, ok)
        if not self.ok:
            self.close()

    def _agree(self, dist, group, on_dev, ok: bool) -> bool:
        import torch as _t
        t = _t.tensor([1 if ok else 0], dtype=_t.int32)
        t = t.to(self.device) if on_dev else t
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return bool(int(t.item()) == 1)

    def allreduce(self, buf, stream: int):
        _check(load().sober_peer_allreduce_f64(self.handle, buf.data_ptr(), buf.numel(), stream), "sober_peer_allreduce_f64")

    def status(self, stream: int = None) -> int:
        """sober_peer_status on `stream` (default: the device's current stream -- the one the all-reduces were queued on, so
        that the clearing of the error word is ordered against them)."""
        if stream is None:
            import torch as _t
            stream = _t.cuda.current_stream(self.device).cuda_stream
        return int(load().sober_peer_status(self.handle, None, 0, stream))

    def self_check(self, dist, group) -> bool:
        """Four calls with rank- and round-dependent data; every rank must see the exact sums.  The ranks enter together
        (a barrier of the group first: a rank that is still loading the library must not look like a dead peer)."""
        import torch as _t
        lib = load()
        on_dev = dist.get_backend(group) == "nccl"
        ok = True
        try:
            lib.sober_peer_set_spin_limit(self.handle, 1 << 21)       # short waits here (~0.5 s): the ranks enter together
            dist.barrier(group=group)
            with _t.cuda.device(self.device):
                n = min(self.n_max, 4096)
                base = _t.arange(n, dtype=_t.float64, device=self.device)
                st = _t.cuda.current_stream(self.device)
                for rnd in range(4):
                    x = (base + rnd) * float(self.rank + 1)
                    self.allreduce(x, st.cuda_stream)
                    st.synchronize()
                    want = (base + rnd) * float(self.world * (self.world + 1) // 2)
                    ok = ok and self.status() == 0 and bool(_t.equal(x, want))
        except Exception:
            ok = False
        if self.handle:
            lib.sober_peer_set_spin_limit(self.handle, 1 << 27)       # production: a late peer is waited for (~30 s), like RCCL would
        self.ok = self._agree(dist, group, on_dev, ok)
        if not self.ok:
            self.close()
        return self.ok

    def close(self):
        if self.handle:
            load().sober_peer_destroy(self.handle)
            self.handle = _vp()


def level_final(job: LevelJob, rows, rows_norm, cand, cand_norm, dt, idx, R, N, row_offset, K, mu_live, out_idx,
                out_w, stream: int):
    _check(load().sober_level_final(C.addressof(job), rows.data_ptr(), _ptr(rows_norm), cand.data_ptr(),
                                    _ptr(cand_norm), dt, idx.data_ptr(), int(R), int(N), int(row_offset),
                                    K.data_ptr(), mu_live.data_ptr(), out_idx.data_ptr(), out_w.data_ptr(), stream),
           "sober_level_final")


def level_final_job(job: LevelJob, fin: "FinalJob", idx, R: int, stream: int):
    _check(load().sober_level_final_job(C.addressof(job), C.addressof(fin), idx.data_ptr(), int(R), stream), "sober_level_final_job")


def nystrom_flags_bytes(n_rungs: int, niter: int) -> int:
    return int(load().sober_nystrom_flags_bytes(n_rungs, niter))


def nystrom_basis(job: NystromJob, phase: int, stream: int):
    _check(load().sober_nystrom_basis(C.addressof(job), int(phase), stream), "sober_nystrom_basis")


def projection(Ut, mean, T, P):
    s_, M = Ut.shape
    _check(load().sober_projection(Ut.data_ptr(), s_, M, _ptr(mean), _ptr(T), 0 if T is None else T.shape[1], P.data_ptr(),
                                   _stream(P)), "sober_projection")


def record_event_pair(ev0, ev1, stream: int):
    """torch.cuda.Event pair (already recorded once, so that their handles exist) re-recorded back to back."""
    _check(load().sober_record_event_pair(ev0.cuda_event, ev1.cuda_event, stream), "sober_record_event_pair")


def jitter_ladder_auto(A, info, k_out):
    _check(load().sober_jitter_ladder_auto(A.data_ptr(), A.shape[0], A.stride(0), info.data_ptr(), info.numel(),
                                           k_out.data_ptr(), _stream(A)), "sober_jitter_ladder_auto")


def jitter_ladder(A, k):
    _check(load().sober_jitter_ladder(A.data_ptr(), A.shape[0], A.stride(0), int(k), _stream(A)), "sober_jitter_ladder")
