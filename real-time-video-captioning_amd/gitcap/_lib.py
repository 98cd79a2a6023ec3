"""ctypes binding of libgitcap.so (include/gitcap.h).  There is no fallback: if the HIP
library is missing or a call fails this module raises."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

from .config import CGitCapConfig
from .student_config import CStudentConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgitcap.so")

# every symbol include/gitcap.h declares (tests/test_cabi.py checks the list against the header)
SYMBOLS = {
    "gitcap_abi_version": (c_int, []),
    "gitcap_create": (c_int, [POINTER(CGitCapConfig), c_int, POINTER(c_void_p)]),
    "gitcap_destroy": (None, [c_void_p]),
    "gitcap_last_error": (c_char_p, [c_void_p]),
    "gitcap_load_tensor": (c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int]),
    "gitcap_finalize_weights": (c_int, [c_void_p]),
    "gitcap_hidden_states_enable": (c_int, [c_void_p, c_int]),
    "gitcap_hidden_states_read": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "gitcap_set_weight_storage": (c_int, [c_void_p, c_int]),
    "gitcap_set_compute": (c_int, [c_void_p, c_int]),
    "gitcap_set_kv_cache": (c_int, [c_void_p, c_int]),
    "gitcap_set_fp8_scale": (c_int, [c_void_p, c_float]),
    "gitcap_fp8_saturations": (c_int, [c_void_p, POINTER(c_int64), c_int]),
    "gitcap_weight_bytes": (c_int, [c_void_p, POINTER(c_int64)]),
    "gitcap_encode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "gitcap_set_visual": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gitcap_text_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "gitcap_greedy": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gitcap_encode_raw": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "gitcap_greedy_raw": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gitcap_greedy_submit": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, POINTER(c_int)]),
    "gitcap_greedy_wait": (c_int, [c_void_p, c_int, c_void_p]),
    "gitcap_greedy_raw_submit": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                         POINTER(c_int)]),
    "gitcap_beam_search_raw_submit": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_float, c_int,
                                              c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_int)]),
    "gitcap_dbg_enc_tap": (c_int, [c_void_p, c_void_p]),
    "gitcap_host_copy": (c_int, [c_void_p, c_void_p, c_int64]),
    "gitcap_poll_errors": (c_int, [c_void_p]),
    "gitcap_preprocess": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "gitcap_beam_topk": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gitcap_beam_search": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "gitcap_beam_search_submit": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_void_p,
                                          c_void_p, c_void_p, POINTER(c_int)]),
    "gitcap_beam_search_wait": (c_int, [c_void_p, c_int, c_void_p]),
    "gitcap_reorder_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gitcap_profile_enable": (c_int, [c_void_p, c_int]),
    "gitcap_profile_read": (c_int, [c_void_p, c_int, POINTER(ctypes.c_double), POINTER(c_int64),
                                    POINTER(ctypes.c_double), POINTER(ctypes.c_double)]),
    "gitcap_dbg_gemm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gitcap_dbg_gemm_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                   c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gitcap_dbg_gemm_f8": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "gitcap_dbg_config": (c_int, [c_int, c_int]),
    "gitcap_dbg_attn_full": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gitcap_dbg_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gitcap_workspace_bytes": (c_int, [c_void_p, POINTER(c_int64)]),
    # student decoder (gitcap/student.py)
    "gitcap_student_create": (c_int, [POINTER(CStudentConfig), c_int, POINTER(c_void_p)]),
    "gitcap_student_destroy": (None, [c_void_p]),
    "gitcap_student_last_error": (c_char_p, [c_void_p]),
    "gitcap_student_load_tensor": (c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int]),
    "gitcap_student_finalize": (c_int, [c_void_p]),
    "gitcap_student_set_memory": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "gitcap_student_forward_decoder": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "gitcap_student_greedy": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gitcap_student_beam_search": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
}

_lib = None


def load() -> ctypes.CDLL:
    """dlopen libgitcap.so.  torch must already be imported so that the HIP runtime the
    library binds to (SONAME libamdhip64.so.7) is the one torch loaded."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C real-time-video-captioning_amd/csrc`). gitcap has no CPU/PyTorch fallback.")
    import torch  # noqa: F401  (loads libamdhip64 first)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class GitcapError(RuntimeError):
    pass


class GitcapExchangeTimeout(GitcapError):
    """GITCAP_ERR_EXCHANGE: a fused GEMM + LayerNorm launch gave up waiting for its sibling workgroups.  The handle has
    switched to unfused launches; results produced since the last clean check must be recomputed (include/gitcap.h)."""


ERR_EXCHANGE = -5


def check(lib, handle, rc: int, what: str) -> None:
    if rc != 0:
        msg = lib.gitcap_last_error(handle)
        cls = GitcapExchangeTimeout if rc == ERR_EXCHANGE else GitcapError
        raise cls(f"{what} failed (status {rc}): {msg.decode() if msg else '?'}")
