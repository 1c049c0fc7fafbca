#!/usr/bin/env python3
"""Quick A/B of builds on selected bench.py cases (desynchronised episodes), alternating, in one GPU call.
    python tools/ab_quick.py reach-joint,pnp-joint ab/a.so ab/b.so [--rounds 2]"""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
cases = sys.argv[1].split(","); args = sys.argv[2:]
rounds = 2
if "--rounds" in args:
    i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
for r in range(rounds):
    for lib in args:
        out = []
        for c in cases:
            steps = {"reach-joint": 400, "reach-IK": 100, "pnp-joint": 300, "pnp-IK": 30, "pnp-joint-grasp": 30}.get(c, 100)
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--case", c, "--steps", str(steps), "--warmup", "60",
                                "--no-cpu-baseline", "--no-secondary"], env=dict(os.environ, MCG_LIB=os.path.abspath(lib)),
                               capture_output=True, text=True)
            import json
            try: out.append(f"{c} {json.loads(p.stdout.strip().splitlines()[-1])['ms_per_step']:.4f}")
            except Exception: out.append(f"{c} FAILED {p.stderr[-200:]}")
        print(lib, "  ".join(out), "ms/step", flush=True)
