import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import load_pkg, slide_off_end_stream
from oracle_lib import Oracle
pkg = load_pkg()
ops = slide_off_end_stream()
for mode in (0, 1):
    e = pkg.PCSR(4096)
    e.set_option("mode", mode)
    o = Oracle(4096)
    for i in range(0, len(ops), 8):
        print("mode", mode, "ops", i, flush=True)
        if i >= 40 and os.environ.get("REPRO_STOP"):
            break
        e.apply(ops[i:i + 8]); o.apply(ops[i:i + 8])
        ok = e.geometry() == o.geometry() and np.array_equal(e.state()[0], o.state()[0]) and np.array_equal(e.state()[1], o.state()[1])
        print("   parity", ok, "invariants", e.check_invariants(), flush=True)
    print(mode, e.geometry() == o.geometry(), np.array_equal(e.state()[0], o.state()[0]), flush=True)
