"""The C-ABI library loads on a machine without a GPU, exports every symbol include/lexls_hip.h declares, and refuses to
compute without a device (there is no CPU fallback in the product).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def header_functions():
    text = open(os.path.join(ROOT, "include", "lexls_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lexls_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from lexls_amd import capi
    lib = capi.lib()
    declared = header_functions()
    assert len(declared) >= 30
    missing = [name for name in declared if not hasattr(lib, name)]
    assert not missing, f"declared in include/lexls_hip.h but not exported: {missing}"
    assert sorted(capi.SYMBOLS) == declared, "lexls_amd/capi.py SYMBOLS is out of sync with the header"


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from lexls_amd import capi
    lib = capi.lib()
    n = C.c_int(-1)
    assert lib.lexls_device_count(C.byref(n)) == 4  # LEXLS_ERR_NO_DEVICE
    h = C.c_void_p()
    dims = (C.c_uint32 * 2)(2, 2)
    assert lib.lexls_lse_create(C.byref(h), 0, 1, 4, 2, dims) == 4
    assert b"no HIP device" in lib.lexls_last_error()
    import lexls_amd
    with pytest.raises(capi.LexlsError):
        lexls_amd.BatchedLexLSE(1, 4, [2, 2])


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    offenders = []
    for base, _, files in os.walk(os.path.join(ROOT, "lexls_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                if re.search(r"\boracle\b", open(os.path.join(base, f), errors="ignore").read().replace("oracle/lexlse_oracle.h", "").replace("see oracle", "")):
                    offenders.append(os.path.join(base, f))
    for base, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            if "#include \"lexlse_oracle.h\"" in open(os.path.join(base, f), errors="ignore").read():
                offenders.append(os.path.join(base, f))
    # comments that merely cite the contract file are allowed; imports / includes / dlopen of the oracle are not
    real = [p for p in offenders if re.search(r"(import\s+oracle|from\s+oracle|liblexls_oracle|#include\s+[\"<].*oracle)", open(p, errors="ignore").read())]
    assert not real, real
