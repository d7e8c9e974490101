python bench.py --no-cpu-baseline --no-ref-cli --steps 3 > gpurun_out/w.json 2> gpurun_out/w.err || { tail -3 gpurun_out/w.err; exit 1; }
python - <<PY
import json
d = json.loads(open('gpurun_out/w.json').read().strip().splitlines()[-1])
print('config2', round(d['value']/1e6,1), d['engine']['rounds'], d['roofline']['kernel_ms'], d['roofline']['avg_launch_us'], d['parity_checked'])
for k in ('config3_mixed','config5_shape_zipf'):
    print(k, round(d[k]['value']/1e6,2), d[k]['engine']['rounds'], d[k]['engine']['replan_factor'], d[k]['parity_checked'])
PY
