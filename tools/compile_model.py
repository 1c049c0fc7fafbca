#!/usr/bin/env python3
"""Offline: compile the reference MJCF into this repo's JSON model tables.

    python tools/compile_model.py [--assets /root/reference/mycobotgym/envs/assets]

Reads the reference's MJCF / STL data files where they lie (never copied into the repo) and
writes ``mycobotgym_amd/assets/*.json``.  Re-run only when the compiler changes; the JSON
tables are committed because /root/reference does not exist on the GPU box.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

from mycobotgym_amd.model.mjcf import MjcfCompiler, save_model  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "mycobotgym_amd", "assets")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--assets", default="/root/reference/mycobotgym/envs/assets")
    args = ap.parse_args()
    jobs = [
        # (output name, xml, mesh rule, dropped bodies)
        ("mycobot280", "mycobot280.xml", "legacy", ()),              # PickAndPlace, joint/IK actuators
        ("mycobot280_reach", "mycobot280.xml", "legacy", ("object0",)),  # Reach: cube unobserved -> dropped
        ("mycobot280_exactmesh", "mycobot280.xml", "exact", ()),
        ("mycobot280_reach_exactmesh", "mycobot280.xml", "exact", ("object0",)),
        ("mycobot280_mocap", "mycobot280_mocap.xml", "legacy", ()),   # mocap controller: + mocap body, weld, finger actuator only
        ("mycobot280_mocap_reach", "mycobot280_mocap.xml", "legacy", ("object0",)),
        ("mycobot280_mocap_exactmesh", "mycobot280_mocap.xml", "exact", ()),
        ("mycobot280_mocap_reach_exactmesh", "mycobot280_mocap.xml", "exact", ("object0",)),
    ]
    os.makedirs(OUT, exist_ok=True)
    for name, xml, rule, drop in jobs:
        model = MjcfCompiler(os.path.join(args.assets, xml), mesh_inertia=rule, drop_bodies=drop).compile()
        path = os.path.join(OUT, name + ".json")
        save_model(model, path)
        print(f"{name}: nbody={model['nbody']} nq={model['nq']} nv={model['nv']} nu={model['nu']} "
              f"ngeom={model['ngeom']} neq={model['neq']} -> {os.path.relpath(path)}")


if __name__ == "__main__":
    main()
