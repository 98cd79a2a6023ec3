"""The C-ABI library loads without a GPU, exports every symbol include/gitcap.h declares, and the
product path fails loudly (no CPU fallback) when no HIP device is present."""
import ctypes
import os
import re

import pytest
import torch

from gitcap import _lib
from gitcap.config import CGitCapConfig, git_tiny

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "gitcap.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gitcap_[a-z_]+)\s*\(", txt)))


def test_header_symbols_are_bound_and_exported():
    names = _declared()
    assert len(names) >= 12
    assert sorted(_lib.SYMBOLS) == names, "gitcap/_lib.py:SYMBOLS must list exactly what include/gitcap.h declares"
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    lib.gitcap_abi_version.restype = ctypes.c_int
    assert lib.gitcap_abi_version() == 1


def test_library_contains_gfx950_code_object():
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"gemm_bf16_kernel" in blob and b"attn_full_kernel" in blob


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly():
    lib = _lib.load()
    cc = CGitCapConfig.from_config(git_tiny(2), 1, 2, 8, 1)
    h = ctypes.c_void_p()
    rc = lib.gitcap_create(ctypes.byref(cc), 0, ctypes.byref(h))
    assert rc < 0 and not h
    assert b"no HIP device" in lib.gitcap_last_error(None)
    from gitcap.model import GitCaptioner
    with pytest.raises(_lib.GitcapError):
        GitCaptioner(git_tiny(2))


def test_bad_config_is_rejected_before_touching_the_device():
    lib = _lib.load()
    cc = CGitCapConfig.from_config(git_tiny(2), 1, 2, 8, 1)
    cc.enc_heads = 3            # head_dim != 64
    h = ctypes.c_void_p()
    assert lib.gitcap_create(ctypes.byref(cc), 0, ctypes.byref(h)) == -1
    assert b"head_dim" in lib.gitcap_last_error(None)
    assert lib.gitcap_create(None, 0, ctypes.byref(h)) == -1
