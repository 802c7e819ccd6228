"""CPU checks of the drop-in boundary: the shared library loads and exports every symbol that
include/ilvlm_hip.h declares; bad arguments are rejected with a message instead of a launch."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ilvlm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ilvlm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from ilvlm_amd import lib
    handle = lib.load()
    names = header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(handle, n), "libilvlm_hip.so does not export %s" % n
    # and the ctypes table binds every one of them (except the two argument-less queries)
    assert set(names) - {"ilvlm_version", "ilvlm_last_error"} == set(lib.SIGNATURES)
    assert handle.ilvlm_version() == 300


def test_ctypes_table_has_the_headers_argument_counts():
    """every prototype of include/ilvlm_hip.h against the ctypes argtypes table: same number of parameters (a missing or
    extra argument would shift every pointer after it)"""
    from ilvlm_amd import lib
    text = open(os.path.join(ROOT, "include", "ilvlm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = dict((m.group(1), m.group(2)) for m in re.finditer(r"\b(ilvlm_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S))
    assert set(lib.SIGNATURES) <= set(protos)
    for name, args in lib.SIGNATURES.items():
        params = protos[name].strip()
        n = 0 if params in ("", "void") else len([x for x in params.split(",") if x.strip()])
        assert n == len(args), "%s: header declares %d parameters, ctypes table has %d" % (name, n, len(args))


def test_bad_arguments_are_rejected_without_a_launch():
    from ilvlm_amd import lib
    h = lib.load()
    epi = lib.GemmEpilogue()
    rc = h.ilvlm_gemm(lib.BF16, 0, 0, 16, 16, 16, None, 16, None, 16, None, 16, C.byref(epi), 1, None)
    assert rc == -1 and b"null" in h.ilvlm_last_error()
    rc = h.ilvlm_layernorm_fwd(None, 0, None, None, None, 0, None, None, 4, 768, 1e-5, 0, 0, None)
    assert rc == -1
    rc = h.ilvlm_sparsemax_fwd(None, None, 1, 10, None)
    assert rc == -1
    with pytest.raises(RuntimeError):
        lib.check(rc, "sparsemax_fwd")


def test_ops_refuse_cpu_tensors():
    import torch
    from ilvlm_amd import ops
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(8, 8), torch.zeros(8, 8), torch.zeros(8, 8))
