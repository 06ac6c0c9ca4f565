#!/bin/bash
# usage: scripts/gpu_round6.sh [tag] [all | r5 | r6]      (on the GPU box through gpurun; about 45 minutes a part)
# Everything rounds 5 and 6 owe an MI355X in ONE lease, every step under its own timeout, every output kept under
# gpurun_out/<tag>/ (a step that fails does not stop the next).  First what round 5 scripted (scripts/gpu_round5.sh: the
# evidence run -- full -m gpu suite, smoke, the driver's bench command --, device Brent against the checker's iterates,
# Path B scheduled and sequential, the headline's rocprofv3 profile, codon and config-4 bench lines, the eight-slot time
# line), then round 6's changes one against the other, each a change whose purpose is speed and whose parity is held on
# the CPU (emulated) already:
#   * gs_matrices_kernel with a column's list of Q in registers       against  -DGS_DP_COLUMN=0 (round 3's loop)
#   * gs_eigen_kernel (scalar pair arithmetic, v_readlane, masked zeroing): 4096 trees with a (kappa, omega) row each,
#     kernel stats -- round 4 measured 0.94 ms per distinct model
#   * walk_hbm_cat_kernel with four-tip subtrees rebuilt in the step (BITO_AMD_HBM_FOLD=2)  against  round 4's walk (the default):
#     config 4 and the 64 / 100 / 128-taxon sizes, with the FETCH_SIZE / WRITE_SIZE passes of both
#   * small calls with set-up, step tables and images as one launch (BITO_AMD_SMALL_PREPARE=1) and a tree's final sums by its
#     last run of tiles (BITO_AMD_PIPE_LAST_UNIT=1: two launches per small call in all)  against  the default's four
#   * Path B with sixteen waves per optimiser workgroup (BITO_AMD_GP_OPT_WAVES=16)  against  four (the default)
# usage: scripts/gpu_round6.sh [tag] [all | r5 | r6]   -- two leases of about 45 minutes: `r5` (round 5's list: the evidence run
# first) and `r6` (round 6's forms against what they replace); `all` runs one behind the other in one call
cd $GRAFT_REPO_ROOT
T=${1:-r6}
PART=${2:-all}
O=gpurun_out/$T
mkdir -p $O
step() { echo "=== $1 ($(date +%T))"; }
if [ "$PART" != r6 ]; then bash scripts/gpu_round5.sh $T 2>&1 | tail -60; fi
if [ "$PART" = r5 ]; then exit 0; fi
step "codon: image kernel forms"
bash scripts/build_gs_variants.sh dp_round3 "-DGS_DP_COLUMN=0" > $O/build_dp_round3.log 2>&1
for lib in bito_amd/libbito_amd.so bito_amd/variants/dp_round3.so; do
  name=codon_$( [ "$lib" = bito_amd/libbito_amd.so ] && echo column_lists || echo round3_loop )
  BITO_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout 600 python3 bench.py --workload codon --steps 10 --warmup 2 --no-cpu-baseline --no-resident > $O/${name}_bench.json 2> $O/${name}_bench.err
  tail -c 400 $O/${name}_bench.json; echo
done
step "codon: kernel stats, one row for all trees and a row per tree"
cd /tmp && export TMPDIR=/tmp
for K in 1 0; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/codon_models_${K}_stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --workload codon --distinct-models $K --steps 6 --warmup 2 --no-cpu-baseline --no-resident > $GRAFT_REPO_ROOT/$O/codon_models_${K}_stats.log 2>&1
  grep -E "gs_eigen_kernel|gs_matrices_kernel|gs_walk_kernel|gs_model_kernel" $GRAFT_REPO_ROOT/$O/codon_models_${K}_stats/s_kernel_stats.csv | cut -c1-200
done
cd $GRAFT_REPO_ROOT
step "config 4 and mid sizes: four-tip subtrees folded (BITO_AMD_HBM_FOLD=2) against pitchforks only (the default)"
for fold in 2 1; do
  BITO_AMD_HBM_FOLD=$fold timeout 900 python3 bench.py --workload config4 --steps 6 --warmup 2 --no-cpu-baseline > $O/config4_fold${fold}_bench.json 2> $O/config4_fold${fold}_bench.err
  tail -c 500 $O/config4_fold${fold}_bench.json; echo
  BITO_AMD_HBM_FOLD=$fold timeout 600 python3 scripts/gpu_hbm_sizes.py 41 64 100 128 > $O/hbm_sizes_fold${fold}.log 2>&1; tail -6 $O/hbm_sizes_fold${fold}.log
  BITO_AMD_HBM_FOLD=$fold bash scripts/profile_config4.sh $T/config4_fold${fold} > $O/profile_config4_fold${fold}.log 2>&1; tail -4 $O/profile_config4_fold${fold}.log | cut -c1-400
done
step "Path B: one, two, eight and sixteen waves per optimiser workgroup against four (the default)"
for dag in ds1 seeded; do
  for waves in 1 2 8 16; do
    BITO_AMD_GP_OPT_WAVES=$waves timeout 400 python3 bench.py --workload gp --gp-dag $dag --steps 20 --warmup 3 --cpu-seconds 3 > $O/gp_${dag}_waves${waves}_bench.json 2> $O/gp_${dag}_waves${waves}_bench.err
  done
  python3 - <<PY
import json
for name in ("gp_${dag}_bench", "gp_${dag}_waves1_bench", "gp_${dag}_waves2_bench", "gp_${dag}_waves8_bench", "gp_${dag}_waves16_bench"):
    try:
        j = json.loads(open("$O/" + name + ".json").read().strip().splitlines()[-1])
        print(name, "ms/step %.3f" % j["ms_per_step"], j["config"]["ms_by_schedule"], j.get("parity", {}).get("after_one_sweep", {}).get("max_d_branch_length"))
    except Exception as err:
        print(name, "no line:", err)
PY
done
step "small calls: one set-up launch (BITO_AMD_SMALL_PREPARE=1), no final-sums launch (BITO_AMD_PIPE_LAST_UNIT=1), both, neither (the default: four launches)"
for fused in 11 10 01 00; do  # (set-up as one launch, final sums by a tree's last unit)
  BITO_AMD_SMALL_PREPARE=${fused:0:1} BITO_AMD_PIPE_LAST_UNIT=${fused:1:1} timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --large-batch 0 > $O/small_calls_prepare${fused}_bench.json 2> $O/small_calls_prepare${fused}_bench.err
  python3 - <<PY
import json
try:
    j = json.loads(open("$O/small_calls_prepare${fused}_bench.json").read().strip().splitlines()[-1])
    print("BITO_AMD_SMALL_PREPARE, BITO_AMD_PIPE_LAST_UNIT = $fused", "value %.0f trees/s" % j["value"], "blocking_call_ms", j["blocking_call_ms"]["trees_per_call"], "cache hit", j["blocking_call_ms"]["trees_per_call_model_cache_hit"])
except Exception as err:
    print("no line:", err)
PY
done
step "done"
