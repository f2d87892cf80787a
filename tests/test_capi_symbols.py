"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/txq.h
declares, and refuses to compute without a GPU (no silent fallback)."""
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "txq.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(txq_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from tetrex_amd import capi
    L = capi.lib()
    declared = _declared_symbols()
    assert declared, "no declarations found in include/txq.h"
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(capi.SYMBOLS) == declared


def test_no_cpu_fallback_without_gpu():
    from tetrex_amd import capi
    if capi.device_count_safe() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.TxqError) as e:
        capi.init()
    assert e.value.code == -4  # TXQ_ERR_STATE
    import numpy as np
    with pytest.raises(capi.TxqError):
        capi.Index.upload_ibf(64, 8, 2, np.zeros(8, dtype=np.uint64))


def test_header_cites_reference_seams():
    text = open(os.path.join(ROOT, "include", "txq.h")).read()
    for cite in ["include/index_base.h:104-107", "include/index_ibf.h:146-150", "include/index_hibf.h:132-147",
                 "include/otf_collector.h:341-393"]:
        assert cite in text
