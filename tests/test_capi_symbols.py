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


def test_product_library_has_no_result_changing_experiment_switch():
    """TXQ_HIBF_STORE bits 4/5 (no row gathers / no stores: wrong masks, for timing experiments) are compiled only with
    -DTXQ_EXPERIMENTS (`make EXPERIMENTS=1`, tools/ab_hibf*.sh); the product library does not know the variable at all.
    Every variable it does read is listed in include/txq.h."""
    lib = open(os.path.join(ROOT, "tetrex_amd", "libtxq.so"), "rb").read()
    names = set(m.decode() for m in re.findall(rb"TXQ_[A-Z0-9_]{3,}", lib))
    assert "TXQ_HIBF_STORE" not in names and "TXQ_HIBF_STORE_KIND" in names
    header = open(os.path.join(ROOT, "include", "txq.h")).read()
    knobs = {n for n in names if not n.startswith("TXQ_ERR") and n not in ("TXQ_OK", "TXQ_HIP")}
    missing = sorted(n for n in knobs if n not in header)
    assert not missing, missing
