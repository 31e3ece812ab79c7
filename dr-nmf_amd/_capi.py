"""ctypes binding of libdrnmf.so (C ABI in include/drnmf.h).

The library is the product path: there is NO fallback.  If the shared object is missing or a
call fails, an exception is raised.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdrnmf.so")

OK = 0
COMM_ID_BYTES = 128          # DRNMF_COMM_ID_BYTES
DIV_ED, DIV_KL, DIV_BETA = 0, 1, 2
MATRIX_F32, MATRIX_BF16X3 = 0, 1     # DRNMF_MATRIX_*


class CellDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("B", "T", "F", "N", "K", "n_D", "n_alph", "alph_len", "n_lam",
                 "return_all_hidden", "operand_f16", "divergence")]


class DenseDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("B", "T", "F", "N", "K", "connect_input", "activation", "return_all_hidden",
                 "operand_f16")]


ACTIVATIONS = {"linear": 0, "relu": 1, "tanh": 2, "sigmoid": 3, "softplus": 4, "hard_sigmoid": 5}

_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
_DP = C.POINTER(CellDesc)
_DDP = C.POINTER(DenseDesc)

# name -> (restype, argtypes); mirrors include/drnmf.h one to one
SIGNATURES = {
    "drnmf_version": (_i32, []),
    "drnmf_create": (_i32, [C.POINTER(_vp), _i32]),
    "drnmf_destroy": (_i32, [_vp]),
    "drnmf_create_unbound": (_i32, [C.POINTER(_vp)]),
    "drnmf_last_error": (C.c_char_p, [_vp]),
    "drnmf_params_bytes": (_sz, [_DP]),
    "drnmf_prepare_params": (_i32, [_vp, _DP, _vp, _vp, _vp, _vp, _vp]),
    "drnmf_cell_workspace_bytes": (_sz, [_DP]),
    "drnmf_cell_launches_per_frame": (_i32, [_DP]),
    "drnmf_cell_forward": (_i32, [_vp, _DP, _vp, _f32, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _sz,
                                  _vp]),
    "drnmf_cell_forward_stateful": (_i32, [_vp, _DP, _vp, _f32, _vp, _vp, _f32, _f32, _f32, _vp, _vp,
                                           _vp, _vp, _sz, _vp]),
    "drnmf_cell_forward_ista": (_i32, [_vp, _DP, _vp, _f32, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _sz,
                                       _vp]),
    "drnmf_cell_profile": (_i32, [_vp, _DP, _vp, _f32, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _sz,
                                  _vp, _i32, C.POINTER(C.c_float)]),
    "drnmf_dense_params_bytes": (_sz, [_DDP]),
    "drnmf_dense_prepare_params": (_i32, [_vp, _DDP, _vp, _vp, _vp, _vp, _vp, _vp]),
    "drnmf_dense_workspace_bytes": (_sz, [_DDP]),
    "drnmf_dense_cell_forward": (_i32, [_vp, _DDP, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                        _vp]),
    "drnmf_dense_backward_workspace_bytes": (_sz, [_DDP]),
    "drnmf_dense_cell_backward": (_i32, [_vp, _DDP, _vp, _f32] + [_vp] * 13 + [_sz, _vp]),
    "drnmf_dense_cell_forward_dropout": (_i32, [_vp, _DDP, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "drnmf_dense_cell_backward_dropout": (_i32, [_vp, _DDP, _vp, _f32] + [_vp] * 14 + [_sz, _vp]),
    "drnmf_dense_cell_forward_dropout_stateful": (_i32, [_vp, _DDP, _vp, _f32] + [_vp] * 7 + [_sz, _vp]),
    "drnmf_dense_cell_backward_stateful": (_i32, [_vp, _DDP, _vp, _f32] + [_vp] * 13 + [_sz, _vp]),
    "drnmf_padded_f": (_i32, [_i32]),
    "drnmf_head_forward": (_i32, [_vp, _i64, _i32, _i32, _vp, _i64, _i32, _vp, _vp, _i32, _vp,
                                  _vp, _vp, _vp, _vp]),
    "drnmf_loss_head_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "drnmf_snmf_cost_head_backward": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _i64, _i32, _vp, _vp,
                                             _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _sz,
                                             _vp]),
    "drnmf_loss_head_backward": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _i32,
                                        _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                        _vp]),
    "drnmf_cell_backward_workspace_bytes": (_sz, [_DP]),
    "drnmf_cell_backward": (_i32, [_vp, _DP, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _vp, _sz,
                                   _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "drnmf_cell_backward_stateful": (_i32, [_vp, _DP, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _sz,
                                            _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "drnmf_cell_backward_ista": (_i32, [_vp, _DP, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _sz, _vp, _sz,
                                        _vp, _vp, _vp, _vp, _vp]),
    "drnmf_cell_backward_ista_stateful": (_i32, [_vp, _DP, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _sz, _vp, _sz,
                                                 _vp, _vp, _vp, _vp, _vp]),
    "drnmf_cell_backward_profile": (_i32, [_vp, _DP, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _vp,
                                           _sz, _vp, _sz, _vp, _vp, _vp, _vp, _vp,
                                           C.POINTER(C.c_float)]),
    "drnmf_adam_step": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _f32, _vp]),
    "drnmf_sumsq": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "drnmf_adam_step_flat": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32,
                                    _f32, _i32, _f32, _vp, _vp]),
    "drnmf_adam_step_flat_counted": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double,
                                            C.c_double, C.c_double, _f32, _f32, _i32, _f32, _vp, _vp, _vp,
                                            _vp]),
    "drnmf_check_status": (_i32, [_vp]),
    "drnmf_status_take_device": (_i32, [_vp, _vp, _vp]),
    "drnmf_reload_env": (_i32, []),
    "drnmf_set_matrix_mode": (_i32, [_vp, _i32]),
    "drnmf_get_matrix_mode": (_i32, [_vp]),
    "drnmf_persist_admitted": (_i32, [_vp]),
    "drnmf_persist_admit_reason": (C.c_char_p, [_vp]),
    "drnmf_host_report_ring": (_i32, [_vp, C.POINTER(C.POINTER(C.c_float)), C.POINTER(_i32)]),
    "drnmf_ista_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "drnmf_ista_forward": (_i32, [_vp, _i64, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _vp, _vp,
                                  _vp, _vp, _sz, _vp]),
    "drnmf_mu_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "drnmf_mu_forward": (_i32, [_vp, _i64, _i32, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp,
                                _vp, _sz, _vp]),
    "drnmf_snmf_train_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "drnmf_snmf_train_init": (_i32, [_vp, _i64, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "drnmf_snmf_train_step": (_i32, [_vp, _i64, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _i32, _vp,
                                     _vp, _sz, _vp]),
    "drnmf_stft_frames": (_i32, [_i64, _i32, _i32]),
    "drnmf_stft_mag": (_i32, [_vp, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "drnmf_stft": (_i32, [_vp, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "drnmf_istft_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "drnmf_istft_masked": (_i32, [_vp, _i32, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz,
                                  _vp]),
    "drnmf_snr": (_i32, [_vp, _i32, _i64, _vp, _vp, _vp, _vp]),
    "drnmf_divide_a_by_aplusb": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "drnmf_add": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "drnmf_loss_forward_workspace_bytes": (_sz, [_i64]),
    "drnmf_loss_forward": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32,
                                  _f32, _vp, _vp, _sz, _vp]),
    "drnmf_wav_int16_workspace_bytes": (_sz, []),
    "drnmf_wav_int16": (_i32, [_vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "drnmf_sdr_workspace_bytes": (_sz, [_i32, _i64, _i32]),
    "drnmf_sdr_corr": (_i32, [_vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "drnmf_comm_unique_id": (_i32, [_vp, _vp]),
    "drnmf_comm_init": (_i32, [_vp, _vp, _i32, _i32]),
    "drnmf_comm_destroy": (_i32, [_vp]),
    "drnmf_comm_info": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "drnmf_allreduce_grads": (_i32, [_vp, _vp, _i64, _vp]),
    "drnmf_broadcast_params": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "drnmf_sdr_project": (_i32, [_vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
}

_lib = None
_handles = {}


class DrnmfError(RuntimeError):
    pass


def lib():
    """Load libdrnmf.so (once).  Raises ImportError with the build command if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libdrnmf.so not found at %s -- build it with `python dr-nmf_amd/build.py` "
                "(or __graft_entry__.build()).  There is no CPU fallback." % LIB_PATH)
        # torch first: it ships its own libamdhip64, and the process must hold ONE HIP runtime.
        # Loaded before torch, libdrnmf.so would pull in /opt/rocm's copy and the second runtime
        # to initialise reports "no ROCm-capable device".
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def handle(device=0):
    """One library handle per device per process."""
    if device not in _handles:
        L = lib()
        h = _vp()
        rc = L.drnmf_create(C.byref(h), int(device))
        if rc != OK:
            raise DrnmfError("drnmf_create(device=%d) failed (%d): %s" %
                             (device, rc, L.drnmf_last_error(None).decode()))
        _handles[device] = h
    return _handles[device]


def check(rc, h, what):
    if rc != OK:
        msg = lib().drnmf_last_error(h).decode(errors="replace")
        exc = ValueError if rc in (-1, -2, -4) else DrnmfError
        raise exc("%s failed (%d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else _vp(t.data_ptr())
