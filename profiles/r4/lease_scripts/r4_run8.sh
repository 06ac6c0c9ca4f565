cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_gp.py tests/test_nni.py tests/test_tp.py -m gpu -x -q 2>&1 | tail -5
for f in 1 0; do for d in ds1 seeded; do
BITO_AMD_GP_FUSED_SWEEP=$f python3 bench.py --workload gp --gp-dag $d --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fused=$f', '$d', 'ms/step %.3f'%d['ms_per_step'], d['config']['ms_by_schedule'])"
done; done
