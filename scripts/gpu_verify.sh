#!/bin/bash
# usage: scripts/gpu_verify.sh <tag>      (on the GPU box through gpurun)
# The round's evidence run: the full `-m gpu` suite, smoke() and the driver's bench command, every line of their output
# kept under gpurun_out/<tag>/ (pytest.log, smoke.log, bench.json, bench.err, loaded_libs.txt, build.txt).  Copy what
# is to be judged into profiles/ as r<round>_gputest_<sha>.log / r<round>_bench_<sha>.json.
cd $GRAFT_REPO_ROOT
T=${1:-verify}
O=gpurun_out/$T
mkdir -p $O
{ echo "device code: sha256 of the built library and of its sources"; sha256sum bito_amd/libbito_amd.so oracle/*.so bito_amd/csrc/*.hip bito_amd/csrc/*.cpp bito_amd/csrc/*.hpp bito_amd/csrc/walk_pipe_gen.inc; rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing"; } > $O/build.txt 2>&1
timeout 2400 python3 -m pytest tests -m gpu -x -q -rA 2>&1 | tee $O/pytest.log | tail -5
python3 - > $O/smoke.log 2>&1 <<'PY'
import __graft_entry__ as g
g.smoke()
print("smoke ok")
print("mapped:", sorted({l.split()[-1] for l in open("/proc/self/maps") if l.rstrip().endswith(".so") and ("bito" in l or "oracle" in l)}))
PY
tail -8 $O/smoke.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json
