cd $GRAFT_REPO_ROOT
BITO_AMD_TRACE_CALL=1 timeout 120 python3 scripts/gpu_call_timeline.py 100 2>&1 | tail -12
BITO_AMD_TRACE_CALL=1 timeout 120 python3 scripts/gpu_call_timeline.py 1 2>&1 | tail -6
bash scripts/gpu_call_trace.sh r4_trace100 100 5 > /dev/null 2>&1
python3 scripts/summarize_call_trace.py gpurun_out/r4_trace100 | tail -14
