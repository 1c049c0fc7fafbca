#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the MyCobot Reach rollout hot path on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one env.step() of every environment (one launch of the fused step kernel = controller + 20 physics
sub-steps + observation + reward + termination + TimeLimit + auto-reset).  Workload = BASELINE.json configs[1]:
Reach, 8192 envs per GPU, free-space dynamics, `joint` controller (SURVEY 8(d) "Config 2", primary), dense reward,
actions ~ U(-1,1) float32 already resident in HBM, auto-reset on.  Weak scaling: every rank owns 8192 envs keyed by
global env id; there is no collective on the step path (RCCL is used once, after the timed region, to reduce the
episode statistics for logging).  The IK controller (100 sub-steps per step) is reported as a secondary figure.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ENVS_PER_GPU = 8192
# Algorithmic (compulsory) HBM bytes per env-step, SURVEY.md 8(d): Reach with the cube's state dropped:
# B = 2*S + A + O, S = 46 doubles + counters, A = 28 B, O = 135 B  ->  939 B.   (DESIGN.md "Bytes per env-step")
BYTES_PER_ENV_STEP = 939
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 lanes/clk x 2 FLOP x 2.4 GHz; the 16 lanes/clk (4.02 clk per wave64
                                # v_fma_f64 per SIMD) is measured: tools/microbench/issue_rate.hip, profiles/r01/issue_rate.log


def cpu_baseline(controller: str, budget_envs: int, steps: int):
    """The CPU oracle (kind "port": this repo's C restatement, NOT MuJoCo) on the host cores this job may use."""
    import numpy as np
    from tests.common import make_oracle
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, 16)     # a one-GPU box's CPU share is 16 cores
    ora = make_oracle(budget_envs, controller_type=controller, reward_type="dense", seed=0, n_threads=cores)
    ora.reset(seed=0)
    rng = np.random.default_rng(0)
    acts = [rng.uniform(-1, 1, (budget_envs, ora.act_dim)).astype(np.float32) for _ in range(4)]
    ora.step(acts[0])
    t0 = time.perf_counter()
    for t in range(steps):
        ora.step(acts[t % 4])
    dt = time.perf_counter() - t0
    return {"value": budget_envs * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{budget_envs} envs x {steps} steps, Reach/{controller}, OpenMP over envs, {dt:.1f}s wall; "
                      "this repo's C float64 restatement, not MuJoCo (unavailable offline)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--controller", default="joint", choices=["joint", "IK", "mocap"])
    ap.add_argument("--task", default="reach", choices=["reach", "pnp"],
                    help="reach = BASELINE configs[1] (the headline metric); pnp = configs[2], PickAndPlace with contacts")
    ap.add_argument("--dr", action="store_true", help="PickAndPlace with per-reset domain randomisation (configs[4])")
    ap.add_argument("--scripted-grasp", action="store_true",
                    help="PickAndPlace from the 'gripper closing over the cube' state (pad contacts in the timed window)")
    ap.add_argument("--envs-per-gpu", type=int, default=N_ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box only): ranks share device 0 and rendezvous over gloo, because RCCL refuses
    # two ranks on one device; the real multi-GPU run is one rank per GPU over RCCL ("nccl")
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    red_dev = "cpu" if share else None        # gloo reduces host tensors
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    dev = torch.device("cuda", local_rank)
    from mycobotgym_amd import MyCobotVecEnv

    n = args.envs_per_gpu
    K, W = args.steps, args.warmup

    def run(controller, steps, warmup):
        pnp = args.task == "pnp"
        envs = MyCobotVecEnv(n, has_object=pnp, controller_type=controller, reward_type="dense", device=dev,
                             seed=0, env_id_offset=rank * n,
                             domain_randomization={"mass": (0.5, 2.0), "friction": (0.5, 1.5)} if (pnp and args.dr) else None,
                             max_episode_steps=10 ** 9 if args.scripted_grasp else 50)
        envs.reset(seed=0)
        g = torch.Generator(device=dev); g.manual_seed(1234 + rank)
        pool = torch.rand(16, n, envs.action_dim, device=dev, generator=g) * 2 - 1     # resident action batches
        if pnp and args.scripted_grasp and controller == "joint":
            from mycobotgym_amd.scenarios import grasp_state
            st = grasp_state(n, seed=rank)
            act = torch.as_tensor(st.pop("action"), device=dev)
            envs.set_state(**st)
            pool = act.unsqueeze(0).repeat(16, 1, 1).contiguous()
        for t in range(warmup):
            envs.step_async(pool[t % 16])
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        # HIP events on the launch stream (mcg_step enqueues on torch's current stream, so torch events see it)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for t in range(steps):
            envs.step_async(pool[t % 16])
        ev1.record()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        kernel_ms = ev0.elapsed_time(ev1) / steps      # average launch-to-launch duration over the timed region
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=red_dev or dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        # logging path: episode statistics reduced over ranks with RCCL, off the timed region
        b = envs._buf
        stats = torch.stack([b["ep_return"].sum(), b["ep_length"].double().sum(), b["is_success"].double().sum()])
        if world > 1:
            if red_dev: stats = stats.to(red_dev)
            dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        envs.close()
        return dt, kernel_ms, stats.tolist()

    dt, kernel_ms, stats = run(args.controller, K, W)
    total_envs = n * world
    value = total_envs * K / dt
    substeps = 100 if args.controller == "IK" else 20
    out = {
        "metric": "env-steps/sec (whole node), MyCobot Reach, N_envs=8192/GPU" if args.task == "reach" else
                  "env-steps/sec (whole node), MyCobot PickAndPlace, N_envs=8192/GPU",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"MyCobot Reach, {n} envs/GPU, no contacts (free-space dynamics), controller={args.controller}, "
                                if args.task == "reach" else
                                f"MyCobot PickAndPlace, {n} envs/GPU, contacts on (cube-table/ground/pads, pyramidal condim 4), "
                                f"{'domain randomisation, ' if args.dr else ''}{'scripted grasp, ' if args.scripted_grasp else ''}controller={args.controller}, ") +
                               f"{substeps} physics sub-steps per env-step, dense reward, auto-reset, TimeLimit 50",
                   "envs_per_gpu": n, "total_envs": total_envs, "controller": args.controller,
                   "parallelism": f"env-sharded x{world}, no step-path collective"},
        "physics_substeps_per_sec": value * substeps,
    }
    if rank == 0:
        algo_bytes = BYTES_PER_ENV_STEP * n
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None; src = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json" if args.task == "reach" else "pmc_latest_pnp.json")
        flops = None
        if os.path.exists(pmc):
            with open(pmc) as f:
                pj = json.load(f)
            if (pj.get("controller") == args.controller and pj.get("n_envs") == n and pj.get("task", "reach") == args.task
                    and not args.scripted_grasp):       # the counters were collected on the default workload of this task
                traffic = pj.get("hbm_bytes_per_launch"); src = "profiles/" + os.path.basename(pmc) + ": " + pj.get("note", "")
                flops = pj.get("f64_flops_per_launch")
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": src,
                           "kernel": "step_reach_kernel", "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algo_bytes,
                           "note": "nominal roofline only: with all sub-steps fused the path moves ~1 KB per env-step and is "
                                   "bound by dependent FP64 VALU issue, not by HBM (SURVEY 8(d)); see DESIGN.md"}
        if args.task == "pnp":
            out["roofline"]["kernel"] = "step_pnp_kernel"
            out["roofline"]["algorithmic_bytes_per_launch"] = 1363 * n       # PickAndPlace B = 1363 B per env-step
            out["roofline"]["achieved"] = 1363 * n / (kernel_ms * 1e-3) / 1e9
            out["roofline"]["frac"] = out["roofline"]["achieved"] / HBM_PEAK_GBPS
        if flops:
            # secondary (the binding) roofline, SURVEY 8(d): FP64 vector FLOP/s.  Peak = 1024 SIMDs x 16 lanes/clk x 2 x 2.4 GHz;
            # the 16 lanes/clk is measured here (tools/microbench/issue_rate.hip: 4.02 clk per wave64 v_fma_f64 per SIMD).
            tf = flops / (kernel_ms * 1e-3) / 1e12
            out["roofline"]["valu_f64"] = {"achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": tf / FP64_VECTOR_PEAK_TFLOPS, "flops_per_launch": flops,
                                           "source": "SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64 x active lanes, " + (src or "")}
        if not args.no_secondary and world == 1 and args.task == "reach":
            other = "joint" if args.controller == "IK" else "IK"
            dt2, k2, _ = run(other, max(K // 5, 20), max(W // 5, 5))
            out["secondary"] = {"controller": other, "env_steps_per_sec": n * max(K // 5, 20) / dt2, "kernel_ms": k2,
                                "physics_substeps_per_sec": n * max(K // 5, 20) / dt2 * (100 if other == "IK" else 20)}
        if not args.no_cpu_baseline and world == 1 and args.task == "reach":
            out["cpu_baseline"] = cpu_baseline(args.controller, 8192, 40)
        out["episode_stats"] = {"sum_return": stats[0], "sum_length": stats[1], "sum_success": stats[2]}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
