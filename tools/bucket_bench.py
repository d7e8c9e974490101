#!/usr/bin/env python3
"""Owner bucketing of a device-resident block (k_bucket_hist / _scan / _scatter): time per call for 1 M and 10 M updates, 8 partitions."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_pkg, load_streams  # noqa: E402
import importlib.util  # noqa: E402

pkg, st = load_pkg(), load_streams()
spec = importlib.util.spec_from_file_location("ppcsr_exchange", os.path.join(ROOT, "parallel-packed-csr_amd", "exchange.py"))
ex = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ex)
for m in (1_000_000, 10_000_000):
    ops = st.random_stream(10_000_000, m, seed=3)
    t = torch.from_numpy(ops.view(np.int32)).cuda()
    for _ in range(2):
        out, counts = ex.bucket_ops_device(t, 10_000_000, 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out, counts = ex.bucket_ops_device(t, 10_000_000, 8)
    torch.cuda.synchronize()
    print(f"{m} updates, 8 partitions: {(time.perf_counter() - t0) / 10 * 1e6:.0f} us per call (incl. output allocation)", flush=True)
