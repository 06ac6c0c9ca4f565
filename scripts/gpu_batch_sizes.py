"""bench.py's config-3 step at several batch sizes (x100 DS1 topologies per pass)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for k in [int(a) for a in sys.argv[1:]] or [16, 64, 160, 400]:
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--replicas", str(k), "--steps", "20", "--warmup", "3",
                          "--no-cpu-baseline"], capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"{100 * k} trees per pass: {d['value']:.0f} trees/s, {d['ms_per_step']:.3f} ms per step, walk kernel "
              f"{d['roofline']['avg_kernel_ms']:.3f} ms, {d['roofline']['frac']:.3f} of the FP64 matrix peak")
    except Exception:  # noqa: BLE001
        print(k, out.stderr[-400:])
