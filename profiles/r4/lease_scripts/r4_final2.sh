cd $GRAFT_REPO_ROOT
timeout 1800 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
timeout 600 python3 bench.py 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value']), j['ms_per_step'], j['roofline']['frac'], j['roofline']['executed']['frac'], j['blocking_call_ms']['trees_per_call'], j['parity']['max_dll'], j['parity']['max_dgrad'], j['cpu_baseline']['value'])"
