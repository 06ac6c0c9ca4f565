#!/bin/bash
# The CPU oracle (oracle/*.c) under AddressSanitizer + UndefinedBehaviorSanitizer, through the whole CPU test suite
# (the GPU side has no sanitizer on this pool: sanitizers run on the CPU build only).  Builds instrumented copies of the
# three oracle libraries in a scratch directory, swaps them in for the run, restores the plain ones.
# usage: scripts/oracle_sanitizers.sh     (from the repository root; a few minutes)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S=$(mktemp -d /tmp/oracle_asan.XXXX)
K=$(mktemp -d /tmp/oracle_keep.XXXX)
cp "$ROOT"/oracle/*.c "$ROOT"/oracle/*.h "$ROOT"/oracle/Makefile "$S"/
make -s -C "$S" CFLAGS="-O1 -g -march=x86-64-v3 -ffp-contract=off -fPIC -std=c11 -fsanitize=address,undefined -fno-omit-frame-pointer" 2>/dev/null
cp "$ROOT"/oracle/*.so "$K"/
trap 'cp "$K"/*.so "$ROOT"/oracle/' EXIT
cp "$S"/*.so "$ROOT"/oracle/
cd "$ROOT"
# (test_host_pool...: that test runs a ThreadSanitizer binary of its own, which cannot start under this preload)
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python3 -m pytest tests -q -m "not gpu" -p no:cacheprovider \
  --deselect tests/test_host.py::test_host_pool_runs_every_part_once_and_is_race_free
