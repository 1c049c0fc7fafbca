#!/usr/bin/env python3
"""Regenerate tests/golden/oracle_regression.json: outputs of THIS REPO'S CPU ORACLE on fixed seeded inputs.

These are regression vectors of the oracle (they catch accidental changes of the restated algorithm); they are NOT
outputs of the reference -- MuJoCo cannot run here (parity unpinned).  Run:  python tools/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from tests.common import make_oracle  # noqa: E402


def rollout(has_object, controller, steps, n=4, seed=2024):
    ora = make_oracle(n, has_object=has_object, controller_type=controller, reward_type="dense", seed=seed, n_threads=1)
    obs, ag, dg = ora.reset(seed=seed)
    rng = np.random.default_rng(seed)
    out = {"reset_obs": obs.tolist(), "reset_goal": dg.tolist(), "steps": []}
    for t in range(steps):
        a = rng.uniform(-1, 1, (n, ora.act_dim)).astype(np.float32)
        o = ora.step(a)
        out["steps"].append({"action": a.tolist(), "obs": o["obs"].tolist(), "reward": o["reward"].tolist()})
    return out


def main():
    data = {"_note": "oracle regression vectors (this repo's C restatement, NOT MuJoCo); tools/make_golden.py",
            "reach_joint": rollout(False, "joint", 2), "reach_ik": rollout(False, "IK", 1),
            "pnp_joint": rollout(True, "joint", 2), "reach_mocap": rollout(False, "mocap", 2)}
    path = os.path.join(ROOT, "tests", "golden", "oracle_regression.json")
    with open(path, "w") as f:
        json.dump(data, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
