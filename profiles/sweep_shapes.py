#!/usr/bin/env python3
"""A/B of the kernel shapes (DN_WAVES=1|2|3) over fleet sizes: python profiles/sweep_shapes.py > gpurun_out/sweep.txt
Each cell is us per vector step (fused K=64 / single-step launches replayed from a hipGraph)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = [int(x) for x in sys.argv[1:]] or [4096, 16384, 32768, 49152, 65536, 98304, 131072, 262144, 524288, 1048576, 2097152]
print(f"{'drones':>9} | {'1w fused':>9} {'2w fused':>9} {'3w fused':>9} | {'1w single':>9} {'2w single':>9}   (us per vector step)")
for n in sizes:
    steps = max(128, min(8192, (1 << 28) // n // 64 * 64))
    ab = 64 if n <= 262144 else 8
    row = {}
    for w in ("1", "2", "3"):
        env = dict(os.environ, DN_WAVES=w)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--num-envs", str(n), "--steps",
                              str(steps), "--warmup", str(max(ab, steps // 8 // ab * ab)), "--action-batches", str(ab), "--no-ppo-rollout"],
                             capture_output=True, text=True, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            row[w] = (float("nan"), float("nan"))
            sys.stderr.write(out.stderr[-2000:])
            continue
        d = json.loads(line[-1])
        row[w] = (d["roofline"]["us_per_vector_step"], d["other_launch_shapes"]["graph"]["us_per_vector_step"])
    print(f"{n:9d} | {row['1'][0]:9.3f} {row['2'][0]:9.3f} {row['3'][0]:9.3f} | {row['1'][1]:9.3f} {row['2'][1]:9.3f}", flush=True)
