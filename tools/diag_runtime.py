import sys, os, ctypes
sys.path.insert(0, "tests")
import numpy as np
order = sys.argv[1]
def maps():
    s = set()
    for l in open("/proc/self/maps"):
        if "amdhip" in l or "hsa-runtime" in l:
            s.add(l.split()[-1])
    return sorted(s)
if order == "torch_first":
    import torch
    x = torch.zeros(4).cuda()
    print("torch ok", maps())
    from helpers import load_pkg
    pkg = load_pkg()
    e = pkg.PCSR(100)
    e.add_edge(1, 2, 1)
    print("engine ok", e.edge_exists(1, 2), maps())
else:
    from helpers import load_pkg
    pkg = load_pkg()
    e = pkg.PCSR(100)
    e.add_edge(1, 2, 1)
    print("engine ok", e.edge_exists(1, 2), maps())
    import torch
    try:
        x = torch.zeros(4).cuda()
        print("torch ok", maps())
    except Exception as ex:
        print("torch FAIL", ex, maps())
import importlib.util
spec = importlib.util.spec_from_file_location("ex", "parallel-packed-csr_amd/exchange.py")
ex = importlib.util.module_from_spec(spec); spec.loader.exec_module(ex)
try:
    t = torch.from_numpy(np.arange(3000, dtype=np.int32).reshape(-1, 3) % 77).cuda()
    out, counts = ex.bucket_ops_device(t, 77, 8)
    print("bucket ok", counts.tolist())
except Exception as exn:
    L = ex._hip_lib()
    L.ppcsr_last_error.restype = ctypes.c_char_p
    print("bucket FAIL", exn, L.ppcsr_last_error())
