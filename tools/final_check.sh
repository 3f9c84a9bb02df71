python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('metric','value','unit','n_gpus','ms_per_step','dtype','vs_baseline','scaling')}); print(d['roofline']); print(d['cpu_baseline']); print(d['config'])"
