import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from helpers import load_pkg, load_streams
pkg, st = load_pkg(), load_streams()
s, d = st.rmat_edges(20, 10_000_000, seed=1)
e = pkg.PCSR(1 << 20)
e.bulk_build(st.adds(s, d))
for k in range(5):
    lv, bms = e.bfs(0, with_ms=True)
print("bfs ms", bms)
