#!/bin/bash
# blocking-call time against the chunking parameters of engine.cpp (scripts/gpu_call_latency.py on two batch sizes)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for taper in 0 1; do
for streams in 1 2; do
for first in 256 512; do
for growth in 2 3; do
for cap in 1024 2048 4096; do
  echo "taper=$taper streams=$streams first=$first growth=$growth cap=$cap"
  BITO_AMD_CHUNK_TAPER=$taper BITO_AMD_CHUNK_WALK_STREAMS=$streams BITO_AMD_CHUNK_FIRST=$first BITO_AMD_CHUNK_GROWTH=$growth BITO_AMD_CHUNK_CAP=$cap BITO_AMD_CHUNK_LANES=8 python scripts/gpu_call_latency.py 1600 6400 | sed 's/, log_likelihoods.*//'
done; done; done; done; done
