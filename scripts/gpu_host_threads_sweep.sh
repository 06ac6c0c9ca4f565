#!/bin/bash
# blocking-call time against the number of host threads of a call (engine.cpp / host_pool.hpp), the chunk plan
# that goes with them and the copy-engine alternatives of worker.cpp (scripts/gpu_call_latency.py on three batch sizes)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $*"; env "$@" python scripts/gpu_call_latency.py 1600 3200 6400 | sed 's/, log_likelihoods.*//'; }
run BITO_AMD_HOST_THREADS=1
run BITO_AMD_HOST_THREADS=8
run BITO_AMD_HOST_THREADS=8 BITO_AMD_INPUTS_COPY_MIN=0
run BITO_AMD_HOST_THREADS=8 BITO_AMD_INPUTS_COPY_MIN=1000 BITO_AMD_CHUNK_FIRST=512
run BITO_AMD_HOST_THREADS=8 BITO_AMD_INPUTS_COPY_MIN=1000 BITO_AMD_CHUNK_FIRST=768
run BITO_AMD_HOST_THREADS=8 BITO_AMD_INPUTS_COPY_MIN=1000 BITO_AMD_CHUNK_FIRST=1536
run BITO_AMD_HOST_THREADS=8
