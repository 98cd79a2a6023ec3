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
    return sorted(set(re.findall(r"\b(gitcap_[a-z0-9_]+)\s*\(", txt)))


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


def _struct_fields(name):
    """Field names of `struct <name> { ... };` in include/gitcap.h, in declaration order."""
    txt = open(os.path.join(ROOT, "include", "gitcap.h")).read()
    body = re.search(r"struct\s+%s\s*\{(.*?)\}\s*\w*\s*;" % name, txt, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            out += [f.strip() for f in decl.split(None, 1)[1].split(",")]
    return out


def test_ctypes_structs_mirror_the_header_field_order():
    from gitcap.student_config import CStudentConfig
    assert [f for f, _ in CStudentConfig._fields_] == _struct_fields("gitcap_student_config")
    assert [f for f, _ in CGitCapConfig._fields_] == _struct_fields("gitcap_config")


def test_student_abi_rejects_bad_configs_and_has_no_cpu_path():
    from gitcap.student_config import CStudentConfig, StudentConfig, student_tiny
    lib = _lib.load()
    h = ctypes.c_void_p()
    bad = CStudentConfig.from_config(StudentConfig(d_model=512, n_head=8, d_ffn=1024), 4, 16)     # no kernel instantiation
    assert lib.gitcap_student_create(ctypes.byref(bad), 0, ctypes.byref(h)) == -1 and not h
    assert b"d_model" in lib.gitcap_student_last_error(None)
    long_ = CStudentConfig.from_config(student_tiny(), 4, 64)                                        # > 63 tokens
    assert lib.gitcap_student_create(ctypes.byref(long_), 0, ctypes.byref(h)) == -1
    assert lib.gitcap_student_create(None, 0, ctypes.byref(h)) == -1
    if not torch.cuda.is_available():
        ok = CStudentConfig.from_config(student_tiny(), 4, 16)
        assert lib.gitcap_student_create(ctypes.byref(ok), 0, ctypes.byref(h)) < 0 and not h
        assert b"no CPU fallback" in lib.gitcap_student_last_error(None)
        from gitcap.student import StudentCaptioner
        with pytest.raises(_lib.GitcapError):
            StudentCaptioner(cfg=student_tiny())


def test_speed_switches_are_process_wide_and_documented():
    """gitcap_dbg_config (the run-time form of the GITCAP_NO_* switches) needs no device: every key the header documents returns its
    previous value and restores; an unknown key is refused.  The product kernels behind keys 10 and 11 must be in the code object."""
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "gitcap.h")).read()
    for key in range(12):
        if key in (2, 3, 6):                 # thresholds / a poll count, not on-off switches
            old = lib.gitcap_dbg_config(key, 1)
            assert old >= 0 and lib.gitcap_dbg_config(key, old) == 1
            continue
        if key == 4:                         # retired in round 6 (224-row GEMM tiles): accepted, no effect
            assert lib.gitcap_dbg_config(4, 0) == 0 and lib.gitcap_dbg_config(4, 1) == 0
            continue
        old = lib.gitcap_dbg_config(key, 0)
        assert old in (0, 1), key
        assert lib.gitcap_dbg_config(key, old) == 0 and lib.gitcap_dbg_config(key, old) == old
        assert re.search(r"\b%d:" % key, hdr) or ("key %d" % key) in hdr, f"include/gitcap.h does not document key {key}"
    assert lib.gitcap_dbg_config(12, 0) < 0
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"skinny_head_kernel" in blob and b"skinny_rows3_kernel" in blob
