"""CPU: the C-ABI library loads and exports every symbol include/gmx.h declares.
No compute calls (there is no GPU in the build container)."""
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "gmx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(gmx_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported():
    import gmx
    if not os.path.exists(gmx.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = gmx.lib()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), "libgmx.so does not export %s" % s
    assert sorted(gmx.EXPORTS) == syms, "gmx.py binding list out of sync with include/gmx.h"


def test_no_cpu_fallback_without_device():
    """On a box without a GPU the product path must fail loudly, not compute on the CPU."""
    import numpy as np

    import gmx
    if gmx.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(gmx.GmxError):
        gmx.Graph.upload(np.array([0, 1, 1], np.int32), np.array([1], np.int32))
    with pytest.raises(gmx.GmxError):
        gmx.Graph.rmat(64, 1024)


def test_product_does_not_reference_oracle():
    """Nothing under green-marl_amd/ or include/ may import, link or name the oracle."""
    bad = []
    for base in ("green-marl_amd", "include"):
        for dp, dn, fn in os.walk(os.path.join(ROOT, base)):
            if "build" in dp.split(os.sep):
                continue
            for f in fn:
                if f.endswith((".py", ".hip", ".cc", ".h", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"pyoracle|liboracle|gm_oracle|oracle/", txt) and "oracle lives in" not in txt:
                        bad.append(os.path.join(dp, f))
                    elif re.search(r"import pyoracle|liboracle\.so|gm_oracle\.h", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
