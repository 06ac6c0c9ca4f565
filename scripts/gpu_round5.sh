#!/bin/bash
# usage: scripts/gpu_round5.sh [tag]      (on the GPU box through gpurun; about half an hour)
# Everything round 5 wants from an MI355X in ONE lease, every step under its own timeout, every output kept under
# gpurun_out/<tag>/ (a step that fails does not stop the next): the evidence run (full -m gpu suite, smoke, the driver's
# bench command), device Brent against the checker's iterates, Path B's bench line with the scheduled executor and with
# the sequential one, the headline's rocprofv3 profile (kernel trace + separate FETCH_SIZE / WRITE_SIZE passes), the
# codon and config-4 bench lines (cache-miss loops, distinct-model sweep), the eight-slot host time line.
cd $GRAFT_REPO_ROOT
T=${1:-r5}
O=gpurun_out/$T
mkdir -p $O
step() { echo "=== $1 ($(date +%T))"; }
step "evidence run"
bash scripts/gpu_verify.sh $T/verify 2>&1 | tail -12
step "Brent traces"
timeout 600 python3 scripts/gpu_gp_brent_trace.py > $O/brent_trace.log 2>&1; tail -12 $O/brent_trace.log
step "Path B bench: scheduled and sequential"
for dag in ds1 seeded; do
  timeout 400 python3 bench.py --workload gp --gp-dag $dag --steps 20 --warmup 3 --cpu-seconds 10 > $O/gp_${dag}_bench.json 2> $O/gp_${dag}_bench.err
  BITO_AMD_GP_SCHEDULE=0 timeout 400 python3 bench.py --workload gp --gp-dag $dag --steps 20 --warmup 3 --no-cpu-baseline > $O/gp_${dag}_sequential_bench.json 2> $O/gp_${dag}_sequential_bench.err
  python3 - <<PY
import json
for name in ("gp_${dag}_bench", "gp_${dag}_sequential_bench"):
    try:
        j = json.loads(open("$O/" + name + ".json").read().strip().splitlines()[-1])
        print(name, "ms/step %.3f" % j["ms_per_step"], j["config"]["ms_by_schedule"], j.get("parity", {}).get("after_one_sweep"))
    except Exception as err:
        print(name, "no line:", err, open("$O/" + name + ".err").read()[-400:])
PY
done
step "Path B kernel stats"
bash scripts/profile_gp_bench.sh $T/gp_ds1 ds1 > $O/profile_gp_ds1.log 2>&1; tail -3 $O/profile_gp_ds1.log | cut -c1-300
step "headline profile"
bash scripts/profile_round.sh $T/call > $O/profile_call.log 2>&1; tail -4 $O/profile_call.log | cut -c1-600
step "codon"
timeout 600 python3 bench.py --workload codon --steps 10 --warmup 2 --cpu-seconds 10 > $O/codon_bench.json 2> $O/codon_bench.err; tail -c 900 $O/codon_bench.json
step "config 4"
timeout 600 python3 bench.py --workload config4 --steps 6 --warmup 2 --cpu-seconds 10 > $O/config4_bench.json 2> $O/config4_bench.err; tail -c 600 $O/config4_bench.json
step "eight-slot host time line"
timeout 300 python3 scripts/gpu_slots_timeline.py 8 6400 > $O/slots8_timeline.log 2>&1; tail -14 $O/slots8_timeline.log
step "weibull+6 site gradient on three slots: time line of the second pass"
BITO_AMD_TRACE_CALL=1 timeout 300 python3 - > $O/second_pass_timeline.log 2>&1 <<'PY'
import numpy as np
import bito_amd
from bito_amd import _capi, workloads
w = workloads.ds1_gtr_weibull4(9)
w.site = "weibull+6"
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights, devices=[0, 0, 0])
for _ in range(3):
    out = eng.gradients(w.parent_ids, w.branch_lengths, w.params, flags=_capi.GRAD_SITE_MODEL)
print(eng.kernel_name(), float(out["site_model"].sum()))
PY
grep -c "second pass" $O/second_pass_timeline.log
step "done"
