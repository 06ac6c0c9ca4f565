#!/bin/bash
# blocking-call time against the chunking parameters of engine.cpp (scripts/gpu_call_latency.py on two batch sizes)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for first in 256 512; do
for growth in 3 4 6; do
for cap in 2048 4096 8192; do
  echo "first=$first growth=$growth cap=$cap"
  BITO_AMD_CHUNK_FIRST=$first BITO_AMD_CHUNK_GROWTH=$growth BITO_AMD_CHUNK_CAP=$cap BITO_AMD_CHUNK_LANES=8 python scripts/gpu_call_latency.py 1600 6400 | sed 's/, log_likelihoods.*//'
done; done; done
