#!/bin/bash
# where the CLI's phase-2 time goes (test_cli_pppcsr_drives_all_partitions_at_once compares it with the C-ABI call from python)
cd "$(dirname "$0")/.."
python - <<'PY'
import sys, os
sys.path.insert(0, "tests")
from helpers import load_streams
import pandas as pd
st = load_streams()
scale, m, u = 18, 2_000_000, 500_000
s, d = st.rmat_edges(scale, m, seed=1)
n0 = 1 << scale
core = st.adds(st.permute_labels(s, n0), d)
s2, d2 = st.rmat_edges(scale, u, seed=2)
upd = st.adds(st.permute_labels(s2, n0), d2)
pd.DataFrame(core[:, :2]).to_csv("/tmp/cli_core.txt", sep=" ", header=False, index=False)
pd.DataFrame(upd[:, :2]).to_csv("/tmp/cli_upd.txt", sep=" ", header=False, index=False)
PY
PPCSR_CLI_TIMING=1 parallel-packed-csr_amd/host/ppcsr_cli -threads=8 -size=500000 -insert -pppcsrnuma -partitions_per_domain=8 -gpus=1 -core_graph=/tmp/cli_core.txt -update_file=/tmp/cli_upd.txt 2>&1 | grep -v "^Thread\|^Done\|^Edges"
