for o in "opt_horizon=6144" "opt_horizon=12288" "opt_horizon=6144" "opt_horizon=12288"; do
python bench.py --config 4 --no-cpu-baseline --no-ref-cli --no-secondary --no-check --steps 2 --opt $o > gpurun_out/w4.json 2> gpurun_out/w4.err || { tail -3 gpurun_out/w4.err; exit 1; }
python - <<PY
import json
d = json.loads(open('gpurun_out/w4.json').read().strip().splitlines()[-1])
print('$o', 'config4', round(d['value']/1e6,1), d['ms_per_step'], d['engine']['rounds'], d['engine']['replan_factor'])
PY
done
