"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/ppcsr.h declares.
No compute is attempted without a GPU; creating an engine must fail loudly (no CPU fallback)."""
import os
import re

import pytest

from helpers import ROOT, load_pkg


@pytest.fixture(scope="module")
def lib():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ppcsr_build", os.path.join(ROOT, "parallel-packed-csr_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    path = b.build_engine()
    return load_pkg().load_library(path)


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "ppcsr.h")).read()
    declared = set(re.findall(r"\b(p{2,3}csr_[a-z_]+)\s*\(", hdr))
    assert len(declared) >= 35
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert set(load_pkg().EXPORTED) <= declared


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pkg = load_pkg()
    with pytest.raises(pkg.PpcsrError):
        pkg.PCSR(10)


def test_bucket_ops_is_stable_and_local(lib, streams):
    import numpy as np
    from oracle_lib import OraclePPPCSR
    pkg = load_pkg()
    ops = streams.random_stream(1003, 5000, seed=3, p_delete=0.3)
    b, counts = pkg.bucket_ops(1003, 8, ops)
    pp = OraclePPPCSR(1003, True, 1, 8)
    owner = np.array([pp.get_partition(int(s)) for s in ops[:, 0]])
    off = 0
    for k in range(8):
        sub = ops[owner == k].copy()
        sub[:, 0] -= np.uint32(pp.partition_start(k))
        assert counts[k] == len(sub)
        np.testing.assert_array_equal(b[off:off + len(sub)], sub)
        off += len(sub)
