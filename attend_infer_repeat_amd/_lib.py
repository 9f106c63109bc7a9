"""ctypes binding of libair_hip.so -- every symbol declared in include/air_hip.h, nothing else.

The library is the product: if it is missing or fails to load, importing any compute path raises (there is no
CPU / PyTorch fallback).  Signatures mirror include/air_hip.h one to one.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libair_hip.so")

c_int, c_float, c_size_t, c_void_p, c_uint64 = (ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p,
                                                 ctypes.c_uint64)
P = c_void_p   # device pointers travel as void*
ABI_VERSION = 10        # == AIR_ABI_VERSION in include/air_hip.h (the stable contract, AIR_API)
ENGINE_ABI_VERSION = 5  # == AIR_ENGINE_ABI_VERSION (the engine plan entries, AIR_ENGINE_API)

class AirGemmDesc(ctypes.Structure):
    """mirror of `struct AirGemmDesc` (include/air_hip.h)"""
    _fields_ = [("ta", c_int), ("tb", c_int), ("M", c_int), ("N", c_int), ("K", c_int),
                ("A", c_void_p), ("lda", c_int), ("B", c_void_p), ("ldb", c_int), ("C", c_void_p), ("ldc", c_int),
                ("bias", c_void_p), ("epilogue", c_int), ("aux", c_void_p), ("ldaux", c_int), ("beta", c_float),
                ("colsum", c_void_p), ("precision", c_int),
                ("A2", c_void_p), ("a_bias", c_void_p), ("a_elu", c_int), ("a_out", c_void_p),
                ("A16", c_void_p), ("B16", c_void_p), ("C16", c_void_p)]


class AirRmspropSlice(ctypes.Structure):
    """mirror of `struct AirRmspropSlice` (include/air_hip.h)"""
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("ms", c_void_p), ("mg", c_void_p), ("mom", c_void_p),
                ("lo", c_size_t), ("hi", c_size_t), ("n_model", c_size_t), ("lr_dev", c_void_p),
                ("lr_mult_tail", c_float), ("decay", c_float), ("momentum", c_float), ("eps", c_float), ("grad_scale", c_float)]


class AirOptFold(ctypes.Structure):
    """mirror of `struct AirOptFold` (include/air_hip.h)"""
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("ms", c_void_p), ("mg", c_void_p), ("mom", c_void_p),
                ("n_model", c_size_t), ("lr_dev", c_void_p),
                ("lr_mult_tail", c_float), ("decay", c_float), ("momentum", c_float), ("eps", c_float), ("grad_scale", c_float),
                ("fold_mask", ctypes.c_uint), ("n_ranges", c_int), ("range_lo", c_size_t * 4), ("range_hi", c_size_t * 4),
                ("global_step_dev", c_void_p), ("rng_state_dev", c_void_p), ("rng_increment", c_uint64)]


class AirBatchGather(ctypes.Structure):
    """mirror of `struct AirBatchGather` (include/air_hip.h)"""
    _fields_ = [("dataset", ctypes.c_void_p), ("n_items", ctypes.c_longlong), ("item_floats", ctypes.c_int), ("shuffle", ctypes.c_int),
                ("B", ctypes.c_int), ("seed_dev", ctypes.c_void_p), ("step_dev", ctypes.c_void_p), ("obs", ctypes.c_void_p),
                ("idx_out", ctypes.c_void_p), ("copy_mask", ctypes.c_uint)]


class AirDxLayer(ctypes.Structure):
    """mirror of `struct AirDxLayer` (include/air_hip.h)"""
    _fields_ = [("w_bf16", ctypes.c_void_p), ("aux", ctypes.c_void_p), ("out", ctypes.c_void_p), ("out_bf16", ctypes.c_void_p),
                ("n_in", ctypes.c_int), ("n_out", ctypes.c_int), ("ldaux", ctypes.c_int), ("ldout", ctypes.c_int)]


class AirDxChain(ctypes.Structure):
    """mirror of `struct AirDxChain` (include/air_hip.h)"""
    _fields_ = [("g_in", ctypes.c_void_p), ("ld_in", ctypes.c_int), ("rows", ctypes.c_int), ("n_layers", ctypes.c_int),
                ("layer", AirDxLayer * 4)]


class AirIpcPeers(ctypes.Structure):
    """mirror of `struct AirIpcPeers` (include/air_hip.h)"""
    _fields_ = [("world", c_int), ("rank", c_int), ("grads", c_void_p * 8), ("params", c_void_p * 8), ("flags", c_void_p * 8)]


class AirGaussBwdEpi(ctypes.Structure):
    """mirror of `struct AirGaussBwdEpi` (include/air_hip.h)"""
    _fields_ = [("problem", c_int), ("pre", c_void_p), ("ld_pre", c_int), ("eps", c_void_p), ("raw_offset", c_float),
                ("p_loc", c_float), ("p_scale", c_float), ("loc", c_void_p), ("scale", c_void_p), ("dkl_row", c_void_p),
                ("dkl_scale", c_float), ("dpre", c_void_p), ("ld_dpre", c_int), ("D", c_int), ("guard_eps", c_float)]


# name -> (restype, argtypes); order and meaning exactly as in include/air_hip.h
SIGNATURES = {
    "air_abi_version": (c_int, []),
    "air_engine_abi_version": (c_int, []),
    "air_status_string": (ctypes.c_char_p, [c_int]),
    "air_build_digest": (ctypes.c_char_p, []),
    "air_st_read_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "air_st_read_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "air_st_write_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "air_st_write_bwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "air_canvas_unroll_fwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                      c_float, P]),
    "air_canvas_unroll_bwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                      c_float, c_float, P]),
    "air_canvas_unroll_bwd_dpresence": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                                c_float, c_float, P]),
    "air_canvas_unroll_bwd_nvil": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                           c_float, c_float, P, c_int, P, P, P, P, P, P, P, P]),
    "air_canvas_unroll_bands": (c_int, [c_int, c_int]),
    "air_canvas_unroll_fwd_banded": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                             c_float, c_float, P]),
    "air_imp_weight": (c_int, [P, c_int, P, P, c_float, P, P, P, c_int, c_int, P, P, c_float, P]),
    "air_nvil_parts": (c_int, [P, c_int, P, P, P, P, P, P, c_int, P, P]),
    "air_gemm": (c_int, [c_int, c_int, c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, P, c_int, P, c_int,
                         c_float, P, P, c_size_t, P]),
    "air_gemm_bf16": (c_int, [c_int, c_int, c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, P, c_int, P, c_int,
                              c_float, P, P, c_size_t, P]),
    "air_gemm_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "air_gemm_grouped": (c_int, [ctypes.POINTER(AirGemmDesc), c_int, P]),
    "air_gemm_grouped_gauss_bwd": (c_int, [ctypes.POINTER(AirGemmDesc), c_int, ctypes.POINTER(AirGaussBwdEpi), P, c_int, P, P, P, P, P, P,
                                           c_int, P, P, c_int, P, c_int, P]),
    "air_gemm_grouped_opt": (c_int, [ctypes.POINTER(AirGemmDesc), c_int, ctypes.POINTER(AirOptFold), P]),
    "air_linear_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "air_linear_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "air_what_sample_pack": (c_int, [P, c_int, P, c_float, c_float, c_float, P, P, P, P, c_int, P, P, P, P, P,
                                     c_int, c_int, c_int, c_int, c_float, P]),
    "air_attend_fwd": (c_int, [P, P, P, c_int, P, P, P, c_int, P, P, P, c_float, c_float, c_float, c_float, c_float,
                               P, P, P, P, P, c_float, c_float, P, P, P, P, P, P, P, P, P,
                               c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    "air_attend_bwd": (c_int, [P, P, P, P, P, P, c_float, c_float, c_float, c_float, c_float, P, P, P, c_int, P, c_float, P,
                               P, P, P, c_float, P, P, c_float, P, P, P, c_float, c_float, P,
                               c_int, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    "air_attend_bwd_dx": (c_int, [P, P, P, P, P, P, c_float, c_float, c_float, c_float, c_float, P, P, P, c_int, P, c_float, P,
                                  P, P, P, c_float, P, P, c_float, P, P, P, c_float, c_float, P,
                                  c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, c_int, P, P, P, c_int, c_int,
                                  c_int, c_float, P]),
    "air_lstm_step_fwd": (c_int, [P, P, P, c_int, P, c_int, P, P, P, c_int, c_int, c_float, c_int, P]),
    "air_lstm_step_fwd_prologue": (c_int, [P, P, P, c_int, P, c_int, P, P, P, c_int, c_int, c_float, c_int,
                                           P, c_size_t, P, c_size_t, P, P, c_int, ctypes.c_double, ctypes.c_double,
                                           ctypes.c_double, ctypes.c_double, ctypes.c_double, P, c_int, P, P, P]),
    "air_lstm_first_step_fwd": (c_int, [P, c_int, c_int, P, P, P, P, P, c_int, P, c_int, P, P, P, c_int, c_int, c_float, c_int,
                                        P, c_size_t, P, c_size_t, P, P, c_int, ctypes.c_double, ctypes.c_double,
                                        ctypes.c_double, ctypes.c_double, ctypes.c_double, P, c_int, P, P, P]),
    "air_lstm_step_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, P]),
    "air_lstm_step_bwd_opt": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, ctypes.POINTER(AirRmspropSlice), P]),
    "air_lstm_step_bwd_entry_fits": (c_int, [c_int, c_int]),
    "air_lstm_step_bwd_entry": (c_int, [P] * 16 + [c_int, c_int, ctypes.POINTER(AirRmspropSlice), P]),
    "air_lstm_pointwise_bwd_opt": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, ctypes.POINTER(AirRmspropSlice), P]),
    "air_lstm_pointwise_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_float, P]),
    "air_lstm_pointwise_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, P]),
    "air_gauss_sample_fwd": (c_int, [P, c_int, P, c_float, c_int, c_float, c_float, c_float, c_float, P, P, P, P,
                                     c_int, c_int, c_float, P]),
    "air_gauss_sample_bwd": (c_int, [P, c_int, P, c_float, c_int, c_float, c_float, c_float, c_float, P, P, P, P,
                                     P, c_float, P, c_int, c_int, c_int, c_float, P, c_int, P, P]),
    "air_what_head_parts": (c_int, [c_int]),
    "air_what_head_fwd": (c_int, [P, c_int, c_int, P, P, P, c_float, c_float, c_float, P, P, P, P, P, c_int, P, P, P, P, P,
                                  c_int, c_int, c_int, c_int, c_float, c_int, P]),
    "air_gauss_sample_bwd_nvil": (c_int, [P, c_int, P, c_float, c_int, c_float, c_float, c_float, c_float, P, P, P, P,
                                          P, c_float, P, c_int, c_int, c_int, P, c_int, P, P, P, P, P, P, c_int, c_float, P, P, c_int, P, P]),
    "air_canvas_unroll_fwd_bwd_fits": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "air_canvas_unroll_fwd_bwd": (c_int, [P, P, P, P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                          c_float, c_float, c_float, P]),
    "air_normal_kl_fwd": (c_int, [P, P, c_float, c_float, c_float, c_float, P, c_int, c_int, P]),
    "air_normal_kl_bwd": (c_int, [P, P, c_float, c_float, c_float, c_float, P, P, P, c_int, c_int, P]),
    "air_presence_fwd": (c_int, [P, P, P, c_float, c_float, c_int, P, P, c_int, c_int, P]),
    "air_presence_bwd": (c_int, [P, c_float, c_float, c_int, P, P, P, c_int, c_int, P]),
    "air_rec_loglik_fwd": (c_int, [P, P, c_float, c_float, P, c_int, c_int, P]),
    "air_rec_loglik_bwd": (c_int, [P, P, c_float, c_float, P, c_float, P, c_int, c_int, P]),
    "air_numsteps_fwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, P]),
    "air_numsteps_bwd": (c_int, [P, P, P, c_float, P, P, P, c_int, c_int, P]),
    "air_presence_numsteps_fwd": (c_int, [P, P, c_float, c_float, P, P, P, P, P, P, P, c_int, c_int, P]),
    "air_numsteps_presence_bwd": (c_int, [P, P, P, c_float, P, P, c_float, P, P, P, c_float, c_float, P, c_int, c_int, P]),
    "air_heads_fwd": (c_int, [P, c_int, P, c_float, c_int, c_float, c_float, c_float, c_float, P, P, P, P, c_int, c_int,
                              P, P, c_float, c_float, P, P, P, P, P, P, P, c_int, c_int, c_float, P]),
    "air_heads_bwd": (c_int, [P, c_int, P, c_float, c_int, c_float, c_float, c_float, c_float, P, P, P, P, P, c_float,
                              P, c_int, c_int, c_int, P, P, P, c_float, P, P, c_float, P, P, P, c_float, c_float, P,
                              c_int, c_int, c_float, P]),
    "air_step_prologue": (c_int, [P, c_size_t, P, c_size_t, P, P, c_int, ctypes.c_double, ctypes.c_double,
                                  ctypes.c_double, ctypes.c_double, ctypes.c_double, P, c_int, P, P, P, P, c_int,
                                  c_int, P]),
    "air_step_prologue_cvt": (c_int, [P, c_size_t, P, c_size_t, P, P, c_int, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_double, ctypes.c_double, P, c_int, P, P, P, P, c_int,
                                      c_int, P, P, c_size_t, P]),
    "air_step_prologue_gather_cvt": (c_int, [P, c_size_t, P, c_size_t, P, P, c_int, ctypes.c_double, ctypes.c_double,
                                             ctypes.c_double, ctypes.c_double, ctypes.c_double, P, c_int, P, P, P, P, c_int,
                                             c_int, P, P, P]),
    "air_batch_gather": (c_int, [P, ctypes.c_longlong, c_int, P, P, c_int, P, c_int, P, P]),
    "air_step_epilogue": (c_int, [P, P, P, P, P, c_size_t, c_size_t, P, c_float, c_float, c_float, c_float, c_float,
                                  P, P, c_uint64, P]),
    "air_lstm_step_fwd_bf16": (c_int, [P, P, P, P, c_int, P, c_int, P, P, P, P, c_int, c_int, c_float, P]),
    "air_lstm_step_bwd_bf16": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, P]),
    "air_lstm_pointwise_bwd_bf16": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, P]),
    "air_step_epilogue_shadow": (c_int, [P, P, P, P, P, c_size_t, c_size_t, P, c_float, c_float, c_float, c_float, c_float,
                                         P, P, c_uint64, P, P]),
    "air_f32_to_bf16": (c_int, [P, P, c_size_t, P]),
    "air_steps_prior": (c_int, [P, c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                ctypes.c_double, P, c_int, P]),
    "air_counter_add": (c_int, [P, ctypes.c_int64, P]),
    "air_nvil": (c_int, [P, P, P, P, P, P, c_int, P, P]),
    "air_l2_grad_add": (c_int, [P, P, ctypes.POINTER(c_size_t), ctypes.POINTER(c_size_t), c_int, c_float, P]),
    "air_baseline_pack": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "air_rmsprop_centered": (c_int, [P, P, P, P, P, c_size_t, P, c_float, c_float, c_float, c_float, c_float, P]),
    "air_rmsprop": (c_int, [P, P, P, P, P, c_size_t, P, c_float, c_float, c_float, c_float, c_int, c_float, P]),
    "air_rng_fill": (c_int, [P, c_size_t, P, c_size_t, P, P]),
    "air_rng_advance": (c_int, [P, c_uint64, P]),
    "air_fill": (c_int, [P, c_size_t, c_float, P]),
    "air_axpby": (c_int, [P, c_float, P, c_float, P, c_size_t, P]),
    "air_tile_rows": (c_int, [P, P, c_int, c_int, P]),
    "air_colsum": (c_int, [P, c_int, P, c_int, c_int, P]),
    "air_sum_leading": (c_int, [P, P, c_int, c_size_t, P]),
    "air_comm_unique_id": (c_int, [P]),
    "air_comm_init": (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, P]),
    "air_comm_available": (c_int, []),
    "air_comm_count": (c_int, [P, ctypes.POINTER(c_int)]),
    "air_comm_destroy": (c_int, [P]),
    "air_allreduce_sum": (c_int, [P, c_size_t, P, P]),
    "air_comm_last_error": (ctypes.c_char_p, []),
    "air_gemm_grouped_gather_fits": (c_int, [ctypes.POINTER(AirGemmDesc), c_int, ctypes.POINTER(AirBatchGather)]),
    "air_gemm_grouped_gather": (c_int, [ctypes.POINTER(AirGemmDesc), c_int, ctypes.POINTER(AirBatchGather), P]),
    "air_mlp_dx_chain_fits": (c_int, [c_int, c_int]),
    "air_mlp_dx_chain_bf16": (c_int, [ctypes.POINTER(AirDxChain), c_int, P]),
    "air_ipc_flags_alloc": (c_int, [ctypes.POINTER(c_void_p), c_size_t]),
    "air_ipc_flags_free": (c_int, [P]),
    "air_ipc_flags_zero": (c_int, [P, c_size_t, P]),
    "air_ipc_handle_get": (c_int, [P, P]),
    "air_ipc_handle_open": (c_int, [P, ctypes.POINTER(c_void_p)]),
    "air_ipc_handle_close": (c_int, [P]),
    "air_dp_ipc_barrier": (c_int, [ctypes.POINTER(AirIpcPeers), c_int, P, P, P]),
    "air_dp_ipc_barrier_wgs": (c_int, [ctypes.POINTER(AirIpcPeers), c_int, P, P, c_int, P]),
    "air_dp_ipc_rs_update_ag": (c_int, [ctypes.POINTER(AirIpcPeers), P, P, P, c_size_t, c_size_t, P, c_float, c_float, c_float, c_float,
                                        P, P, c_uint64, P]),
    "air_stream_wait_event": (c_int, [P, P]),
    "air_graph_begin_capture": (c_int, [P]),
    "air_graph_end_capture": (c_int, [P, ctypes.POINTER(c_void_p)]),
    "air_graph_launch": (c_int, [P, P]),
    "air_graph_destroy": (c_int, [P]),
    "air_event_create": (c_int, [ctypes.POINTER(c_void_p)]),
    "air_event_record": (c_int, [P, P]),
    "air_event_elapsed_ms": (c_int, [P, P, ctypes.POINTER(c_float)]),
    "air_event_destroy": (c_int, [P]),
}

_lib = None


class AirHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises loudly if the HIP extension is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AirHipError(
            f"{LIB_PATH} not found: the HIP extension is the product and there is no fallback. "
            "Build it with `python -m attend_infer_repeat_amd.build` (or __graft_entry__.build()).")
    # torch bundles its own libamdhip64; import it FIRST so that this library binds to the same HIP runtime instance
    # (loading /opt/rocm's copy first leaves the process with two runtimes and kernels fail with hipErrorNoDevice).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.air_abi_version() != ABI_VERSION or lib.air_engine_abi_version() != ENGINE_ABI_VERSION:
        raise AirHipError(f"libair_hip.so ABI version {lib.air_abi_version()} / engine {lib.air_engine_abi_version()} != binding "
                          f"{ABI_VERSION} / {ENGINE_ABI_VERSION}: rebuild (python -m attend_infer_repeat_amd.build)")
    # a binary compiled from other sources than the ones next to it (a checkout that changed csrc/ without a rebuild) would
    # be called with possibly changed argument lists -- ctypes cannot notice -- so it is refused
    from . import build as _build
    if os.path.isdir(_build.CSRC) and os.environ.get("AIR_SKIP_DIGEST_CHECK") != "1":
        built, have = lib.air_build_digest().decode(), _build.source_digest()
        if built != have:
            raise AirHipError(f"{LIB_PATH} was built from different sources (binary {built[:12]}, tree {have[:12]}): "
                              "rebuild with `python -m attend_infer_repeat_amd.build`")
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        msg = load().air_status_string(int(status))
        raise AirHipError(f"{what or 'air_hip call'} failed with status {status}: {msg.decode() if msg else '?'}")
