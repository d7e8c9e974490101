timeout -k 10 300 python3 tools/_z.py
python bench.py --no-cpu-baseline --no-ref-cli --no-secondary --steps 3 > gpurun_out/w.json 2> gpurun_out/w.err || { tail -3 gpurun_out/w.err; exit 1; }
python - <<PY
import json
d = json.loads(open('gpurun_out/w.json').read().strip().splitlines()[-1])
print('config2', round(d['value']/1e6,1), d['engine']['rounds'], d['parity_checked'])
PY
