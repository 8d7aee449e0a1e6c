"""
ctypes binding of the C ABI (include/bsvi.h) — the only way the Python host layer reaches
the HIP kernels.  There is deliberately no fallback: if ``libbsvi.so`` is missing or no
MI355X is visible, evaluating a model raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbsvi.so")
ABI_VERSION = 11
OUT_HEADER = 4


class NativeError(RuntimeError):
    pass


class Sized(C.Structure):
    """A descriptor / argument struct of the C ABI: its first member is `struct_size`, the size of the struct as THIS binding
    declares it (include/bsvi.h: the library refuses a call whose struct_size differs from its own sizeof)."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(self)


class UniformEntry(C.Structure):
    _fields_ = [("src", C.c_uint32), ("transform", C.c_uint8), ("is_param", C.c_uint8),
                ("reserved", C.c_uint16), ("a", C.c_float), ("b", C.c_float)]


class Record(C.Structure):
    _fields_ = [("code_begin", C.c_uint32), ("code_end", C.c_uint32), ("n_elems", C.c_uint32),
                ("temp_base", C.c_uint32), ("n_temps", C.c_uint32), ("flags", C.c_uint32)]


class ProgramDesc(Sized):
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32), ("n_params", C.c_uint32), ("n_consts", C.c_uint32),
                ("n_obs", C.c_uint32), ("n_slots", C.c_uint32), ("n_noise", C.c_uint32),
                ("n_uniform", C.c_uint32), ("n_uniform_grad", C.c_uint32), ("n_records", C.c_uint32),
                ("n_code", C.c_uint32), ("estimator", C.c_uint32),
                ("uniform", C.c_void_p), ("records", C.c_void_p), ("code", C.c_void_p), ("consts", C.c_void_p),
                ("param_uniform_ptr", C.c_void_p), ("param_uniform_idx", C.c_void_p)]


class ElboArgs(Sized):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32),
                ("params_dev", C.c_void_p), ("obs_dev", C.c_void_p), ("noise_dev", C.c_void_p),
                ("seed", C.c_uint64), ("offset", C.c_uint64),
                ("n_samples_local", C.c_uint32), ("n_samples_global", C.c_uint32),
                ("sample_base", C.c_uint32), ("reserved", C.c_uint32),
                ("out_dev", C.c_void_p), ("samples_out_dev", C.c_void_p), ("noise_out_dev", C.c_void_p),
                ("fvalue_out_dev", C.c_void_p), ("workspace_dev", C.c_void_p), ("stream", C.c_void_p),
                ("offset_dev", C.c_void_p), ("f_weight_dev", C.c_void_p), ("q_weight_dev", C.c_void_p)]


class OptCfg(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("lr", C.c_float), ("momentum", C.c_float), ("dampening", C.c_float),
                ("weight_decay", C.c_float), ("nesterov", C.c_uint32), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("amsgrad", C.c_uint32), ("maximize", C.c_uint32)]


class DenseDesc(Sized):
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32), ("n_params", C.c_uint32), ("n_consts", C.c_uint32),
                ("n_uniform", C.c_uint32), ("n_uniform_grad", C.c_uint32),
                ("n_classes", C.c_uint32), ("n_features", C.c_uint32), ("dataset_size", C.c_uint32),
                ("batch_size", C.c_uint32), ("likelihood", C.c_uint32),
                ("q_loc_u", C.c_uint32), ("q_scale_u", C.c_uint32), ("prior_loc_u", C.c_uint32),
                ("prior_scale_u", C.c_uint32), ("q_loc_stride", C.c_uint32), ("q_scale_stride", C.c_uint32),
                ("prior_loc_stride", C.c_uint32), ("prior_scale_stride", C.c_uint32),
                ("lik_weight", C.c_float), ("prior_weight", C.c_float), ("entropy_weight", C.c_float),
                ("estimator", C.c_uint32), ("reserved", C.c_uint32),
                ("uniform", C.c_void_p), ("consts", C.c_void_p), ("param_uniform_ptr", C.c_void_p),
                ("param_uniform_idx", C.c_void_p), ("dataset", C.c_void_p), ("labels", C.c_void_p)]


class DenseArgs(Sized):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32),
                ("params_dev", C.c_void_p), ("noise_dev", C.c_void_p), ("indices_dev", C.c_void_p),
                ("seed", C.c_uint64), ("offset", C.c_uint64),
                ("n_samples_local", C.c_uint32), ("n_samples_global", C.c_uint32),
                ("sample_base", C.c_uint32), ("reserved", C.c_uint32),
                ("out_dev", C.c_void_p), ("noise_out_dev", C.c_void_p), ("indices_out_dev", C.c_void_p),
                ("fvalue_out_dev", C.c_void_p), ("workspace_dev", C.c_void_p), ("stream", C.c_void_p),
                ("f_weight_dev", C.c_void_p), ("q_weight_dev", C.c_void_p), ("logq_out_dev", C.c_void_p)]


DENSE_EXPORTS = {
    "bsvi_dense_create": (C.c_int, [C.POINTER(DenseDesc), C.POINTER(C.c_void_p)]),
    "bsvi_dense_destroy": (None, [C.c_void_p]),
    "bsvi_dense_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_uint32]),
    "bsvi_dense_exact_data": (C.c_int, [C.c_void_p]),
    "bsvi_dense_fwd_bwd": (C.c_int, [C.c_void_p, C.POINTER(DenseArgs)]),
    "bsvi_dense_finalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "bsvi_dense_step": (C.c_int, [C.c_void_p, C.POINTER(DenseArgs), C.POINTER(OptCfg), C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
}



EXPORTS = {
    "bsvi_program_create": (C.c_int, [C.POINTER(ProgramDesc), C.POINTER(C.c_void_p)]),
    "bsvi_program_destroy": (None, [C.c_void_p]),
    "bsvi_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_uint32]),
    "bsvi_elbo_fwd_bwd": (C.c_int, [C.c_void_p, C.POINTER(ElboArgs)]),
    "bsvi_finalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "bsvi_optimizer_step": (C.c_int, [C.POINTER(OptCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_uint32, C.c_void_p]),
    "bsvi_finalize_step": (C.c_int, [C.POINTER(OptCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                     C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsvi_finalize_step_counted": (C.c_int, [C.POINTER(OptCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p]),
    "bsvi_svi_step": (C.c_int, [C.c_void_p, C.POINTER(ElboArgs), C.POINTER(OptCfg), C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsvi_train_persistent": (C.c_int, [C.c_void_p, C.POINTER(ElboArgs), C.POINTER(OptCfg), C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "bsvi_train_persistent2": (C.c_int, [C.c_void_p, C.POINTER(ElboArgs), C.POINTER(OptCfg), C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "bsvi_train_persistent_exchange": (C.c_int, [C.c_void_p, C.POINTER(ElboArgs), C.POINTER(OptCfg), C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsvi_persistent_supported": (C.c_int, [C.c_void_p, C.c_uint32]),
    "bsvi_persistent_split_shares": (C.c_int, [C.c_void_p, C.c_uint32]),
    "bsvi_program_set_shares": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32]),
    "bsvi_train_persistent_split": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(ElboArgs),
                                              C.POINTER(OptCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "bsvi_program_source": (C.c_size_t, [C.POINTER(ProgramDesc), C.c_int, C.c_char_p, C.c_size_t]),
    "bsvi_jit_compile": (C.c_int, [C.c_char_p, C.POINTER(C.c_size_t)]),
    "bsvi_jit_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "bsvi_jit_last_origin": (C.c_int, []),
    "bsvi_jit_cache_dir": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "bsvi_jit_compiler_identity": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "bsvi_program_engine": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint32)]),
    "bsvi_max_lds_bytes": (C.c_int, [C.c_void_p]),
    "bsvi_query_geometry": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "bsvi_debug_math": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                  C.c_void_p]),
    "bsvi_debug_set_stamps": (None, [C.c_void_p]),
    "bsvi_last_error": (C.c_char_p, []),
    "bsvi_abi_version": (C.c_int, []),
    "bsvi_sizeof": (C.c_size_t, [C.c_int]),
    "bsvi_device_count": (C.c_int, []),
}
EXPORTS.update(DENSE_EXPORTS)


class BnnLayer(C.Structure):
    _fields_ = [("rows", C.c_uint32), ("cols", C.c_uint32), ("weight_row0", C.c_uint32), ("bias_row0", C.c_uint32),
                ("activation", C.c_uint32), ("reserved", C.c_uint32)]


class BnnDesc(Sized):
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32), ("n_params", C.c_uint32), ("n_consts", C.c_uint32),
                ("n_uniform", C.c_uint32), ("n_uniform_grad", C.c_uint32),
                ("n_layers", C.c_uint32), ("n_rows", C.c_uint32), ("n_features", C.c_uint32), ("dataset_size", C.c_uint32),
                ("batch_size", C.c_uint32), ("likelihood", C.c_uint32), ("estimator", C.c_uint32),
                ("lik_weight", C.c_float), ("prior_weight", C.c_float), ("entropy_weight", C.c_float),
                ("layers", C.c_void_p), ("row_uniform", C.c_void_p), ("uniform", C.c_void_p), ("consts", C.c_void_p),
                ("param_uniform_ptr", C.c_void_p), ("param_uniform_idx", C.c_void_p), ("dataset", C.c_void_p), ("labels", C.c_void_p)]


class BnnArgs(Sized):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32),
                ("params_dev", C.c_void_p), ("noise_dev", C.c_void_p), ("indices_dev", C.c_void_p),
                ("seed", C.c_uint64), ("offset", C.c_uint64),
                ("n_samples_local", C.c_uint32), ("n_samples_global", C.c_uint32), ("sample_base", C.c_uint32), ("reserved", C.c_uint32),
                ("out_dev", C.c_void_p), ("noise_out_dev", C.c_void_p), ("indices_out_dev", C.c_void_p),
                ("fvalue_out_dev", C.c_void_p), ("logq_out_dev", C.c_void_p), ("workspace_dev", C.c_void_p), ("stream", C.c_void_p),
                ("f_weight_dev", C.c_void_p), ("q_weight_dev", C.c_void_p)]


EXPORTS.update({
    "bsvi_bnn_create": (C.c_int, [C.POINTER(BnnDesc), C.POINTER(C.c_void_p)]),
    "bsvi_bnn_destroy": (None, [C.c_void_p]),
    "bsvi_bnn_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_uint32]),
    "bsvi_bnn_exact_data": (C.c_int, [C.c_void_p]),
    "bsvi_bnn_fwd_bwd": (C.c_int, [C.c_void_p, C.POINTER(BnnArgs)]),
    "bsvi_bnn_finalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "bsvi_bnn_step": (C.c_int, [C.c_void_p, C.POINTER(BnnArgs), C.POINTER(OptCfg), C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p]),
})


class MlpLayer(C.Structure):
    _fields_ = [("in_value", C.c_uint32), ("out_value", C.c_uint32), ("n_in", C.c_uint32), ("n_out", C.c_uint32),
                ("weight_off", C.c_uint32), ("bias_off", C.c_uint32), ("activation", C.c_uint32),
                ("post_add", C.c_float), ("split_col", C.c_uint32), ("activation2", C.c_uint32),
                ("post_add2", C.c_float), ("reserved", C.c_uint32)]


class AmortDesc(Sized):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32), ("abi_version", C.c_uint32), ("n_params", C.c_uint32),
                ("n_features", C.c_uint32), ("latent_dim", C.c_uint32), ("dataset_size", C.c_uint32),
                ("batch_size", C.c_uint32), ("n_enc_layers", C.c_uint32), ("n_dec_layers", C.c_uint32),
                ("enc_loc_value", C.c_uint32), ("enc_scale_value", C.c_uint32),
                ("enc_loc_col", C.c_uint32), ("enc_scale_col", C.c_uint32),
                ("dec_logits_value", C.c_uint32), ("likelihood", C.c_uint32),
                ("enc_layers", C.POINTER(MlpLayer)), ("dec_layers", C.POINTER(MlpLayer)),
                ("prior_loc", C.c_void_p), ("prior_scale", C.c_void_p), ("dataset", C.c_void_p),
                ("likelihood_scale", C.c_void_p), ("prior_loc_off", C.c_uint32), ("prior_scale_off", C.c_uint32),
                ("lik_scale_off", C.c_uint32), ("lik_scale_size", C.c_uint32), ("dec_scale_value", C.c_uint32), ("reserved1", C.c_uint32)]


class AmortArgs(Sized):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32),
                ("params_dev", C.c_void_p), ("noise_dev", C.c_void_p), ("indices_dev", C.c_void_p),
                ("seed", C.c_uint64), ("offset", C.c_uint64),
                ("n_samples_local", C.c_uint32), ("n_samples_global", C.c_uint32),
                ("sample_base", C.c_uint32), ("estimator", C.c_uint32),
                ("out_dev", C.c_void_p), ("noise_out_dev", C.c_void_p), ("indices_out_dev", C.c_void_p),
                ("fvalue_out_dev", C.c_void_p), ("logq_out_dev", C.c_void_p), ("workspace_dev", C.c_void_p),
                ("stream", C.c_void_p), ("f_weight_dev", C.c_void_p), ("q_weight_dev", C.c_void_p)]


EXPORTS.update({
    "bsvi_amort_create": (C.c_int, [C.POINTER(AmortDesc), C.POINTER(C.c_void_p)]),
    "bsvi_amort_destroy": (None, [C.c_void_p]),
    "bsvi_amort_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_uint32]),
    "bsvi_amort_exact_data": (C.c_int, [C.c_void_p]),
    "bsvi_amort_fwd_bwd": (C.c_int, [C.c_void_p, C.POINTER(AmortArgs)]),
    "bsvi_amort_bucket": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "bsvi_amort_set_bucket_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bsvi_amort_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "bsvi_debug_gemm": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                  C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                  C.c_float, C.c_uint32, C.c_void_p]),
})

EXPORTS.update({
    "bsvi_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bsvi_exchange_handle_bytes": (C.c_size_t, []),
    "bsvi_exchange_create": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "bsvi_exchange_export": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bsvi_exchange_connect": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bsvi_exchange_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "bsvi_exchange_status": (C.c_int, [C.c_void_p]),
    "bsvi_exchange_selftest_tagged": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "bsvi_minibatch_gather": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsvi_exchange_destroy": (None, [C.c_void_p]),
})

class MvnInsn(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("flag", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32), ("imm", C.c_float)]


class MvnDesc(Sized):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32), ("abi_version", C.c_uint32), ("dim", C.c_uint32), ("n_code", C.c_uint32), ("n_mats", C.c_uint32),
                ("n_slot_inputs", C.c_uint32), ("n_uniform_inputs", C.c_uint32), ("value_is_latent", C.c_uint32),
                ("loc_is_param", C.c_uint32),
                ("code", C.POINTER(MvnInsn)), ("mats", C.c_void_p), ("loc", C.c_void_p), ("value", C.c_void_p),
                ("uniform_inputs", C.c_void_p), ("loc_entries", C.c_void_p), ("weight", C.c_float), ("form", C.c_uint32),
                ("value_entries", C.c_void_p), ("value_is_param", C.c_uint32), ("reserved1", C.c_uint32)]


class MvnArgs(Sized):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32),
                ("params_dev", C.c_void_p), ("samples_dev", C.c_void_p), ("rows_out_dev", C.c_void_p),
                ("n_samples_local", C.c_uint32), ("value_row0", C.c_uint32), ("input_rows", C.c_uint32 * 8),
                ("stream", C.c_void_p)]


MVN_KIND = dict(MAT=0, INPUT=1, IMM=2, BIN=3, UN=4)
MVN_FORM = dict(covariance_matrix=0, scale_tril=1, precision_matrix=2)      # bsvi_mvn_form
EXPORTS.update({
    "bsvi_mvn_create": (C.c_int, [C.POINTER(MvnDesc), C.POINTER(C.c_void_p)]),
    "bsvi_mvn_destroy": (None, [C.c_void_p]),
    "bsvi_mvn_rows_out": (C.c_uint32, [C.POINTER(MvnDesc)]),
    "bsvi_mvn_eval": (C.c_int, [C.c_void_p, C.POINTER(MvnArgs)]),
    "bsvi_mvn_source": (C.c_size_t, [C.POINTER(MvnDesc), C.c_char_p, C.c_size_t]),
})


def mvn_desc(node):
    """lowering.ExternalMvn -> (bsvi_mvn_desc, the arrays it points into)"""
    code = (MvnInsn * len(node.code))(*[MvnInsn(kind=MVN_KIND[k], flag=f, a=a, b=b, imm=imm) for k, f, a, b, imm in node.code])
    mats = np.ascontiguousarray(node.mats, dtype=np.float32)
    loc = np.ascontiguousarray(node.loc, dtype=np.float32)
    value = np.ascontiguousarray(node.value if node.value is not None else np.zeros(node.dim), dtype=np.float32)
    uni = np.ascontiguousarray(node.uniform_inputs)
    loc_entries = getattr(node, "loc_entries", None)
    loc_entries = np.ascontiguousarray(loc_entries) if loc_entries is not None else None
    value_entries = getattr(node, "value_entries", None)
    value_entries = np.ascontiguousarray(value_entries) if value_entries is not None else None
    keep = dict(code=code, mats=mats, loc=loc, value=value, uni=uni, loc_entries=loc_entries, value_entries=value_entries)
    d = MvnDesc(abi_version=ABI_VERSION, dim=node.dim, n_code=len(node.code), n_mats=mats.shape[0] if mats.size else 0,
                n_slot_inputs=len(node.slot_inputs), n_uniform_inputs=len(uni), value_is_latent=int(node.value is None),
                value_is_param=int(value_entries is not None), value_entries=_ptr(value_entries) if value_entries is not None else None,
                loc_is_param=int(loc_entries is not None), loc_entries=_ptr(loc_entries) if loc_entries is not None else None,
                code=code, mats=_ptr(mats), loc=_ptr(loc), value=_ptr(value), uniform_inputs=_ptr(uni) if len(uni) else None,
                weight=float(node.weight), form=MVN_FORM[getattr(node, "form", "covariance_matrix")])
    return d, keep


class MvnNode:
    """Owns a ``bsvi_mvn*`` created from a lowering.ExternalMvn (the batched multivariate-normal kernel)."""

    def __init__(self, node):
        self.lib = load()
        d, self._keep = mvn_desc(node)
        handle = C.c_void_p()
        check(self.lib.bsvi_mvn_create(C.byref(d), C.byref(handle)))
        self.handle, self.node = handle, node
        assert int(self.lib.bsvi_mvn_rows_out(C.byref(d))) == node.n_rows_out

    def eval(self, params_ptr, samples_ptr, rows_out_ptr, n_local, stream):
        args = MvnArgs(params_dev=params_ptr, samples_dev=samples_ptr, rows_out_dev=rows_out_ptr, n_samples_local=n_local,
                       value_row0=self.node.value_row0, stream=stream)
        for k, row in enumerate(self.node.slot_inputs):
            args.input_rows[k] = row
        check(self.lib.bsvi_mvn_eval(self.handle, C.byref(args)))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.bsvi_mvn_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def mvn_source(node):
    lib = load()
    d, keep = mvn_desc(node)
    need = lib.bsvi_mvn_source(C.byref(d), None, 0)
    if need == 0:
        raise NativeError("bsvi_mvn_source: " + lib.bsvi_last_error().decode())
    buf = C.create_string_buffer(need)
    lib.bsvi_mvn_source(C.byref(d), buf, need)
    return buf.value.decode()


class ReduceDesc(Sized):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32), ("abi_version", C.c_uint32), ("rows", C.c_uint32), ("cols", C.c_uint32),
                ("n_code", C.c_uint32), ("n_mats", C.c_uint32), ("n_slot_inputs", C.c_uint32), ("n_uniform_inputs", C.c_uint32),
                ("n_data", C.c_uint32), ("drawn", C.c_uint32), ("reserved1", C.c_uint32),
                ("code", C.POINTER(MvnInsn)), ("mats", C.c_void_p), ("uniform_inputs", C.c_void_p), ("data_mean", C.c_void_p),
                ("data_scale", C.c_void_p), ("weight", C.c_float), ("reserved2", C.c_uint32)]


class ReduceArgs(Sized):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32),
                ("params_dev", C.c_void_p), ("samples_dev", C.c_void_p), ("rows_out_dev", C.c_void_p),
                ("n_samples_local", C.c_uint32), ("reserved1", C.c_uint32), ("input_rows", C.c_uint32 * 8),
                ("data_dev", C.c_void_p), ("seed", C.c_uint64), ("offset", C.c_uint64), ("stream", C.c_void_p)]


EXPORTS.update({
    "bsvi_reduce_create": (C.c_int, [C.POINTER(ReduceDesc), C.POINTER(C.c_void_p)]),
    "bsvi_reduce_destroy": (None, [C.c_void_p]),
    "bsvi_reduce_rows_out": (C.c_uint32, [C.POINTER(ReduceDesc)]),
    "bsvi_reduce_eval": (C.c_int, [C.c_void_p, C.POINTER(ReduceArgs)]),
    "bsvi_reduce_source": (C.c_size_t, [C.POINTER(ReduceDesc), C.c_char_p, C.c_size_t]),
})


def reduce_desc(node):
    """lowering.ExternalReduce -> (bsvi_reduce_desc, the arrays it points into)"""
    code = (MvnInsn * len(node.code))(*[MvnInsn(kind=MVN_KIND[k], flag=f, a=a, b=b, imm=imm) for k, f, a, b, imm in node.code])
    mats = np.ascontiguousarray(node.mats, dtype=np.float32)
    uni = np.ascontiguousarray(node.uniform_inputs)
    mean = np.ascontiguousarray(node.data_mean, dtype=np.float32)
    scale = np.ascontiguousarray(node.data_scale, dtype=np.float32) if node.drawn else None
    keep = dict(code=code, mats=mats, uni=uni, mean=mean, scale=scale)
    d = ReduceDesc(abi_version=ABI_VERSION, rows=node.rows, cols=node.cols, n_code=len(node.code), n_mats=mats.shape[0] if mats.size else 0,
                   n_slot_inputs=len(node.slot_inputs), n_uniform_inputs=len(uni), n_data=node.n_data, drawn=int(node.drawn),
                   code=code, mats=_ptr(mats) if mats.size else None, uniform_inputs=_ptr(uni) if len(uni) else None,
                   data_mean=_ptr(mean), data_scale=_ptr(scale) if scale is not None else None, weight=float(node.weight))
    return d, keep


class ReduceNode:
    """Owns a ``bsvi_reduce*`` created from a lowering.ExternalReduce (links with a reduction the per-sample program does not unroll)."""

    def __init__(self, node):
        self.lib = load()
        d, self._keep = reduce_desc(node)
        handle = C.c_void_p()
        check(self.lib.bsvi_reduce_create(C.byref(d), C.byref(handle)))
        self.handle, self.node = handle, node
        assert int(self.lib.bsvi_reduce_rows_out(C.byref(d))) == node.n_rows_out

    def eval(self, params_ptr, samples_ptr, rows_out_ptr, n_local, stream, seed=0, offset=0, data_ptr=None):
        args = ReduceArgs(params_dev=params_ptr, samples_dev=samples_ptr, rows_out_dev=rows_out_ptr, n_samples_local=n_local,
                          data_dev=data_ptr, seed=int(seed) & 0x7FFFFFFFFFFFFFFF, offset=int(offset), stream=stream)
        for k, row in enumerate(self.node.slot_inputs):
            args.input_rows[k] = row
        check(self.lib.bsvi_reduce_eval(self.handle, C.byref(args)))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.bsvi_reduce_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


# bsvi_struct_kind (include/bsvi.h) -> the ctypes mirror of the struct: load() checks every size against bsvi_sizeof()
STRUCT_KINDS = {0: UniformEntry, 1: Record, 2: ProgramDesc, 3: ElboArgs, 4: OptCfg, 5: DenseDesc, 6: DenseArgs, 7: MlpLayer,
                8: AmortDesc, 9: AmortArgs, 10: MvnInsn, 11: MvnDesc, 12: MvnArgs, 13: BnnLayer, 14: BnnDesc, 15: BnnArgs,
                16: ReduceDesc, 17: ReduceArgs}

_lib = None


def load():
    """dlopen libbsvi.so (built by brancher_amd/csrc/Makefile / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP runtime (soname libamdhip64.so.7).  Import torch first so
    # that libbsvi.so binds to that same runtime instance: two HIP runtimes in one process do
    # not share devices, streams or allocations.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise NativeError("{} not found: build it with `make -C brancher_amd/csrc` (hipcc --offload-arch=gfx950). "
                          "There is no CPU fallback.".format(LIB_PATH))
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in EXPORTS.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.bsvi_abi_version() != ABI_VERSION:
        raise NativeError("libbsvi.so ABI version {} != {}".format(lib.bsvi_abi_version(), ABI_VERSION))
    for kind, cls in STRUCT_KINDS.items():
        if lib.bsvi_sizeof(kind) != C.sizeof(cls):
            raise NativeError("{}: this binding declares {} bytes, libbsvi.so {} (include/bsvi.h and native.py disagree)".format(
                cls.__name__, C.sizeof(cls), lib.bsvi_sizeof(kind)))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise NativeError("bsvi error {}: {}".format(rc, load().bsvi_last_error().decode()))


def _ptr(arr):
    return arr.ctypes.data_as(C.c_void_p) if arr is not None and arr.size else None


def program_desc(program):
    """lowered ``Program`` -> (bsvi_program_desc, the arrays it points into)"""
    from brancher_amd.lowering import EST
    k = dict(
        uniform=np.ascontiguousarray(program.uniform), records=np.ascontiguousarray(program.records),
        code=np.ascontiguousarray(program.code, dtype=np.uint32),
        consts=np.ascontiguousarray(program.consts, dtype=np.float32),
        ptr=np.ascontiguousarray(program.param_uniform_ptr, dtype=np.uint32),
        idx=np.ascontiguousarray(program.param_uniform_idx, dtype=np.uint32))
    assert k["uniform"].dtype.itemsize == C.sizeof(UniformEntry)
    assert k["records"].dtype.itemsize == C.sizeof(Record)
    d = ProgramDesc(abi_version=ABI_VERSION, n_params=program.n_params, n_consts=k["consts"].size,
                    n_obs=program.obs.size, n_slots=program.n_slots, n_noise=program.n_noise,
                    n_uniform=len(k["uniform"]), n_uniform_grad=program.n_uniform_grad,
                    n_records=len(k["records"]), n_code=len(k["code"]), estimator=EST[program.estimator],
                    uniform=_ptr(k["uniform"]), records=_ptr(k["records"]), code=_ptr(k["code"]),
                    consts=_ptr(k["consts"]), param_uniform_ptr=_ptr(k["ptr"]), param_uniform_idx=_ptr(k["idx"]))
    return d, k


def specialised_source(program, variant=0):
    """The HIP translation unit libbsvi generates for a lowered program (bsvi_program_source): variant 0 is the
    training kernel, 1 the diagnostic one.  Host only — no device needed.  None when the program is left to the
    interpreter kernels."""
    lib = load()
    d, keep = program_desc(program)
    need = lib.bsvi_program_source(C.byref(d), variant, None, 0)
    if need == 0:
        return None
    buf = C.create_string_buffer(need)
    lib.bsvi_program_source(C.byref(d), variant, buf, need)
    return buf.value.decode()


def jit_compile(source):
    """hiprtc-compile a generated translation unit for gfx950 (no device needed); returns the code-object size."""
    n = C.c_size_t()
    check(load().bsvi_jit_compile(source.encode(), C.byref(n)))
    return n.value


JIT_ORIGINS = {0: "none", 1: "hiprtc", 2: "process cache", 3: "disk cache"}


def jit_load(source):
    """the code object of a generated translation unit through the caches (bsvi_jit_load): (bytes, origin name)"""
    n, origin = C.c_size_t(), C.c_int()
    check(load().bsvi_jit_load(source.encode(), C.byref(n), C.byref(origin)))
    return n.value, JIT_ORIGINS[origin.value]


def jit_cache_dir():
    lib = load()
    need = lib.bsvi_jit_cache_dir(None, 0)
    buf = C.create_string_buffer(need)
    lib.bsvi_jit_cache_dir(buf, need)
    return buf.value.decode()


def jit_compiler_identity():
    """what the code-object cache's key holds about the compiler that will run (bsvi_jit_compiler_identity)"""
    lib = load()
    need = lib.bsvi_jit_compiler_identity(None, 0)
    buf = C.create_string_buffer(need)
    lib.bsvi_jit_compiler_identity(buf, need)
    return buf.value.decode()


class NativeProgram:
    """Owns a ``bsvi_program*`` created from a lowered ``Program``."""

    def __init__(self, program):
        lib = load()
        if lib.bsvi_device_count() < 1:
            raise NativeError("no MI355X / HIP device visible: the engine cannot run (no CPU fallback)")
        d, self._keep = program_desc(program)
        handle = C.c_void_p()
        check(lib.bsvi_program_create(C.byref(d), C.byref(handle)))
        self.handle = handle
        self.lib = lib
        self.program = program
        self._elbo_share_sets = {}      # V -> [NativeProgram]: shares of the model's log-prob records
        self._elbo_shares_set = 0
        self._shares_wanted = 0
        self.ensure_shares(1)

    def workspace_bytes(self, n_local):
        return int(self.lib.bsvi_workspace_bytes(self.handle, n_local))

    def geometry(self, n_local):
        nb, nw, zg, lds = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        check(self.lib.bsvi_query_geometry(self.handle, n_local, C.byref(nb), C.byref(nw), C.byref(zg), C.byref(lds)))
        mode, lanes = zg.value & 0xFF, (zg.value >> 8) & 0xFF
        return dict(n_blocks=nb.value, n_waves=nw.value, zglobal=mode == 2,
                    storage=("lds+wave_sum", "lds+lane_acc", "global")[mode], lanes_per_wave=lanes or 64,
                    lds_bytes=lds.value)

    def engine(self, n_local, mode=0):
        """which kernels serve a call over n_local samples (mode 0 fwd_bwd, 1 svi_step, 2 persistent loop):
        dict(engine="specialised", n_blocks, n_threads, lds_bytes) or dict(engine="interpreter")"""
        nb, nt, lds = C.c_uint32(), C.c_uint32(), C.c_uint32()
        if self.lib.bsvi_program_engine(self.handle, n_local, mode, C.byref(nb), C.byref(nt), C.byref(lds)):
            return dict(engine="specialised", n_blocks=nb.value, n_threads=nt.value, lds_bytes=lds.value)
        return dict(engine="interpreter")

    def persistent_supported(self, n_local):
        return bool(self.lib.bsvi_persistent_supported(self.handle, n_local))

    def ensure_shares(self, n_local):
        """choose the widest split of the model's log-prob records for which every (sample group, share) workgroup of a
        launch over `n_local` samples still gets a CU of its own (bsvi_program_set_shares) ..."""
        available = getattr(self.program, "shares", {})
        if not available:
            return
        blocks = self.geometry(n_local)["n_blocks"] if n_local > 1 else 1
        self._shares_wanted = max([v for v in available if v * blocks <= 256] or [0])

    def attach_shares(self):
        """... and create / attach that split — in front of a launch that uses it (bsvi_elbo_fwd_bwd, bsvi_svi_step).  The in-kernel
        training loop does not: a `perform_inference` call that only trains never creates the (up to eight) share programs, 5 of the
        11 ms such a call took on a fresh model (profiles/r6/perform_inference_readme.txt)."""
        V = self._shares_wanted
        if V == self._elbo_shares_set:
            return
        available = self.program.shares
        if V >= 2 and V not in self._elbo_share_sets:
            self._elbo_share_sets[V] = [NativeProgram(self._share_program(code, records)) for code, records in available[V]]
        if V >= 2:
            arr = (C.c_void_p * V)(*[sp.handle for sp in self._elbo_share_sets[V]])
            check(self.lib.bsvi_program_set_shares(self.handle, arr, V))
        else:
            check(self.lib.bsvi_program_set_shares(self.handle, None, 0))
        self._elbo_shares_set = V

    def _share_program(self, code, records):
        import copy
        share = copy.copy(self.program)
        share.code, share.records, share.shares = code, records, {}
        return share

    def split_shares(self, n_local):
        """program shares for the multi-workgroup persistent trainer at this sample count: None, or a ctypes array of
        `bsvi_program*` created (once) from `Program.shares[V]` — same tables, every share its own code"""
        import copy
        V = int(self.lib.bsvi_persistent_split_shares(self.handle, n_local))
        parts = getattr(self.program, "shares", {}).get(V) if V > 1 else None
        if not parts:
            return None
        cache = self.__dict__.setdefault("_share_programs", {})
        if V not in cache:
            cache[V] = [NativeProgram(self._share_program(code, records)) for code, records in parts]
        arr = (C.c_void_p * V)(*[sp.handle for sp in cache[V]])
        return arr

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.bsvi_program_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def make_opt_cfg(optimizer, **kw):
    """torch.optim keyword arguments -> bsvi_opt_cfg (`brancher/optimizers.py:53-67` forwards
    ``**opt_params`` verbatim to ``getattr(torch.optim, name)``)."""
    name = optimizer if isinstance(optimizer, str) else getattr(optimizer, "__name__", str(optimizer))
    if name == "SGD":
        allowed = {"lr", "momentum", "dampening", "weight_decay", "nesterov", "maximize"}
        extra = set(kw) - allowed
        if extra:
            raise NotImplementedError("SGD options {} are not supported by the fused optimizer".format(sorted(extra)))
        if kw.get("nesterov") and (kw.get("momentum", 0) <= 0 or kw.get("dampening", 0) != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        return OptCfg(kind=0, lr=kw.get("lr", 1e-3), momentum=kw.get("momentum", 0.0),
                      dampening=kw.get("dampening", 0.0), weight_decay=kw.get("weight_decay", 0.0),
                      nesterov=int(bool(kw.get("nesterov", False))), maximize=int(bool(kw.get("maximize", False))))
    if name == "Adam":
        allowed = {"lr", "betas", "eps", "weight_decay", "amsgrad", "maximize"}
        extra = set(kw) - allowed
        if extra:
            raise NotImplementedError("Adam options {} are not supported by the fused optimizer".format(sorted(extra)))
        b1, b2 = kw.get("betas", (0.9, 0.999))
        return OptCfg(kind=1, lr=kw.get("lr", 1e-3), beta1=b1, beta2=b2, eps=kw.get("eps", 1e-8),
                      weight_decay=kw.get("weight_decay", 0.0), amsgrad=int(bool(kw.get("amsgrad", False))),
                      maximize=int(bool(kw.get("maximize", False))))
    raise NotImplementedError("optimizer {!r}: the fused device optimizer implements torch.optim.SGD and "
                              "torch.optim.Adam".format(name))
