"""Host-side time line of ONE blocking gradients call on an engine over several device slots (round 4):
`slots` x `trees per slot` config-3 trees on devices [0] * slots -- one GPU named several times, every slot served like a
device of its own by its own issuing thread, so the host side of an N-GPU call is what runs.  Prints per slot when each
chunk was staged / issued / had its results back (BITO_AMD_TRACE_CALL=1 lines of the last call, re-sorted by slot) and
the two checks: every slot's first chunk issued within 0.3 ms of the call's start, every slot's second chunk issued
before that slot's first chunk has its results back.
usage: BITO_AMD_TRACE_CALL=1 python scripts/gpu_slots_timeline.py [slots=8] [trees per slot=6400] 2> trace.txt"""
import os
import re
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if os.environ.get("BITO_AMD_TRACE_CALL") != "1":  # re-run with the trace on and read it back
    env = dict(os.environ, BITO_AMD_TRACE_CALL="1")
    proc = subprocess.run([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True)
    sys.stdout.write(proc.stdout)
    calls = proc.stderr.split("---- call")
    last = calls[-1]
    rows = re.findall(r"\s*([\d.]+) us\s+(staged|issued|results of|copied out) chunk (\d+) \(slot (\d+) lane (\d+), (\d+) trees\)", last)
    slots = {}
    for us, what, chunk, slot, lane, trees in rows:
        slots.setdefault(int(slot), {}).setdefault(int(lane), {"trees": int(trees)})[what] = float(us)
    ok_first = ok_second = True
    for slot in sorted(slots):
        lanes = slots[slot]
        line = f"slot {slot}:"
        for lane in sorted(lanes):
            c = lanes[lane]
            line += (f"  chunk {lane} ({c['trees']} trees) staged {c.get('staged', -1):7.1f} issued {c.get('issued', -1):7.1f} "
                     f"results {c.get('results of', -1):8.1f} us")
        print(line)
        ok_first &= lanes[0].get("issued", 1e9) <= 300.0
        if 1 in lanes:
            ok_second &= lanes[1].get("issued", 1e9) <= lanes[0].get("results of", 0.0)
    m = re.search(r"call \d+: ([\d.]+) ms", last)
    print(f"the call: {m.group(1) if m else '?'} ms; every slot's first chunk issued within 0.3 ms: {ok_first}; "
          f"every slot's second chunk issued before its first chunk's results were back: {ok_second}")
    sys.exit(0 if proc.returncode == 0 else proc.returncode)

import numpy as np

import bito_amd
from bito_amd import workloads

slots = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per_slot = int(sys.argv[2]) if len(sys.argv) > 2 else 6400
T = slots * per_slot
w = workloads.ds1_gtr_weibull4(-(-T // 100)).subset(T)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights,
                      devices=[0] * slots)
pid = np.ascontiguousarray(w.parent_ids, dtype=np.int32)
par = np.ascontiguousarray(w.params)
bls = [np.ascontiguousarray(w.branch_lengths), np.ascontiguousarray(w.branch_lengths * 1.03125)]
ll, grad = np.zeros(T), np.zeros((T, 2 * w.taxon_count - 1))
for k in range(8):
    if k >= 6:
        sys.stderr.write("---- call %d\n" % k)
    t0 = time.perf_counter()
    eng.gradients_into(pid, bls[k & 1], par, ll, grad)
    if k >= 6:
        sys.stderr.write("call %d: %.3f ms\n" % (k, (time.perf_counter() - t0) * 1e3))
print(f"{slots} slots x {per_slot} trees on GPU 0, host threads per slot: one issuing thread each; "
      f"finite results: {bool(np.all(np.isfinite(ll)) and np.all(np.isfinite(grad)))}")
