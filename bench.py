#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the MyCobot rollout hot path on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one env.step() of every environment (one launch of the fused step kernel = controller + 20 physics
sub-steps + observation + reward + termination + TimeLimit + auto-reset).  Headline workload = BASELINE.json
configs[1]: Reach, 8192 envs per GPU, free-space dynamics, `joint` controller (SURVEY 8(d) "Config 2", primary), dense
reward, actions ~ U(-1,1) float32 already resident in HBM, auto-reset on.

Episodes are DESYNCHRONISED by default: every env starts at its own random point of its 50-step episode, as in any training
run after its first few hundred steps, so a short timed window is representative (with lock-stepped episodes all 8192 envs
reset on the same step and a 20-step window measures one phase of the episode only; `--lockstep` keeps that mode and the
default run reports it next to the headline figure).

Weak scaling: every rank owns 8192 envs keyed by global env id; there is no collective on the step path (RCCL is used
once, after the timed region, to reduce the episode statistics for logging: mycobotgym_amd.sharding.reduce_episode_stats).

The default 1-GPU run also times, as `secondary` entries with their own `roofline`: Reach with the IK controller (100
sub-steps per step), PickAndPlace (configs[2]: cube resting on the table, random actions), PickAndPlace during a scripted
grasp (pad contacts: the coupled robot + cube solve), and PickAndPlace with per-reset domain randomisation (configs[4]'s
per-GPU workload).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ENVS_PER_GPU = 8192
# Algorithmic (compulsory) HBM bytes per env-step, SURVEY.md 8(d): B = 2*S + A + O.  Reach with the cube's state dropped:
# S = 46 doubles + counters, A = 28 B, O = 135 B -> 939 B; PickAndPlace: 1363 B.   (DESIGN.md "Bytes per env-step")
ALGO_BYTES = {"reach": 939, "pnp": 1363}
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 lanes/clk x 2 FLOP x 2.4 GHz; the 16 lanes/clk (4.02 clk per wave64
                                # v_fma_f64 per SIMD) is measured: tools/microbench/issue_rate.hip, profiles/r01/issue_rate.log

# name -> (task, controller, domain randomisation, scripted grasp)
CASES = {
    "reach-joint": ("reach", "joint", False, False),
    "reach-IK": ("reach", "IK", False, False),
    "reach-mocap": ("reach", "mocap", False, False),
    "pnp-joint": ("pnp", "joint", False, False),
    "pnp-IK": ("pnp", "IK", False, False),
    "pnp-mocap": ("pnp", "mocap", False, False),
    "pnp-joint-dr": ("pnp", "joint", True, False),
    "pnp-joint-grasp": ("pnp", "joint", False, True),
}
DEFAULT_SECONDARY = ("reach-IK", "pnp-joint", "pnp-joint-grasp", "pnp-joint-dr", "pnp-IK", "pnp-mocap")


def describe(case, n, lockstep):
    task, controller, dr, grasp = CASES[case]
    sub = 100 if controller == "IK" else 20
    head = (f"MyCobot Reach, {n} envs/GPU, no contacts (free-space dynamics), " if task == "reach" else
            f"MyCobot PickAndPlace, {n} envs/GPU, contacts on (cube-table/ground/pads, pyramidal condim 4), "
            + ("per-reset domain randomisation (cube mass x U(0.5,2), sliding friction x U(0.5,1.5)), " if dr else "")
            + ("scripted grasp (every env closing its gripper on the cube: pad contacts in the timed window, no TimeLimit), " if grasp else ""))
    return head + (f"controller={controller}, {sub} physics sub-steps per env-step, dense reward, auto-reset, TimeLimit 50, "
                   + ("lock-stepped episodes" if lockstep else "desynchronised episodes (per-env random initial elapsed in [0,50))"))


def cpu_info():
    model = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    quota = None                         # the box's CPU share is a cgroup bandwidth quota, invisible to the affinity mask
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:               # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()
            if q != "max": quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f: q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f: per = int(f.read())
            if q > 0: quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        usable = max(1, min(usable, int(quota + 0.5)))
    return {"nproc": os.cpu_count() or 1, "usable_cores": usable, "cgroup_cpu_quota": quota, "cpu_model": model}


def cpu_baseline(controller: str):
    """The CPU oracle (kind "port": this repo's C restatement, NOT MuJoCo -- unavailable offline) on the host cores this
    job may use, at 1 thread and at all usable cores; bounded samples of the headline workload."""
    import numpy as np
    from tests.common import make_oracle
    info = cpu_info()

    def timed(n_envs, steps, threads):
        ora = make_oracle(n_envs, controller_type=controller, reward_type="dense", seed=0, n_threads=threads)
        ora.reset(seed=0)
        rng = np.random.default_rng(0)
        ora.set_state(elapsed=rng.integers(0, 50, n_envs).astype(np.int32))       # desynchronised, like the GPU run
        acts = [rng.uniform(-1, 1, (n_envs, ora.act_dim)).astype(np.float32) for _ in range(4)]
        ora.step(acts[0])
        t0 = time.perf_counter()
        for t in range(steps):
            ora.step(acts[t % 4])
        dt = time.perf_counter() - t0
        return n_envs * steps / dt, dt

    cores = min(info["usable_cores"], 16)       # a one-GPU box's CPU share is 16 cores, whatever the affinity mask says (256 on the pool's hosts)
    v1, dt1 = timed(512, 60, 1)
    vn, dtn = timed(8192, 60, cores)
    return {"value": vn, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"8192 envs x 60 steps, Reach/{controller}, OpenMP over envs on {cores} threads, {dtn:.1f}s wall; "
                      "this repo's C float64 restatement, not MuJoCo (unavailable offline)",
            "one_thread": {"value": v1, "unit": "env-steps/s", "cores": 1, "sample": f"512 envs x 60 steps, {dt1:.1f}s wall"},
            **info}


def load_pmc(case, n):
    """Counter figures for `case` from profiles/pmc_latest.json (written by tools/summarize_profile.py from rocprofv3 --pmc
    passes of this same command); None when the counters were collected on another size."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        pj = json.load(f).get("cases", {}).get(case)
    if not pj or pj.get("n_envs") != n:
        return None
    from mycobotgym_amd.build import source_hash
    if pj.get("src_sha256") != source_hash():        # counters of other kernels than the ones this run times: not this run's traffic
        return {"stale": True, "note": "profiles/pmc_latest.json[" + case + "] was collected on kernel sources " + str(pj.get("src_sha256"))[:12]
                + "..., this build is " + source_hash()[:12] + "...: re-run tools/profile_case.sh"}
    return pj


def roofline(case, n, kernel_ms):
    task = CASES[case][0]
    algo = ALGO_BYTES[task] * n
    achieved = algo / (kernel_ms * 1e-3) / 1e9
    pj = load_pmc(case, n)
    stale = pj.get("note") if pj and pj.get("stale") else None
    if stale: pj = None
    src = ("profiles/pmc_latest.json[" + case + "]: " + pj.get("note", "")) if pj else stale
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
           "traffic": pj.get("hbm_bytes_per_launch") if pj else None, "traffic_source": src,
           "traffic_over_algorithmic": (pj["hbm_bytes_per_launch"] / algo) if pj else None,
           "sq_wait_any_frac": pj.get("sq_wait_any_frac") if pj else None,
           "scratch_bytes_per_lane": pj.get("scratch_bytes_per_lane") if pj else None,
           "kernel": "step_reach_kernel" if task == "reach" else "step_pnp_kernel", "kernel_ms": kernel_ms,
           "algorithmic_bytes_per_launch": algo,
           "note": "nominal roofline only: with all sub-steps fused the path moves ~1 KB per env-step and is bound by "
                   "dependent FP64 VALU issue, not by HBM (SURVEY 8(d)); see DESIGN.md"}
    if pj and pj.get("f64_flops_per_launch"):
        # secondary (the binding) roofline, SURVEY 8(d): FP64 vector FLOP/s
        tf = pj["f64_flops_per_launch"] / (kernel_ms * 1e-3) / 1e12
        out["valu_f64"] = {"achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VECTOR_PEAK_TFLOPS,
                           "flops_per_launch": pj["f64_flops_per_launch"],
                           "source": "SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64 x active lanes, " + (src or "")}
    return out


def _rehearsal_engine(rank):
    """Stand-in for MyCobotVecEnv under BENCH_REHEARSE_PLUMBING=1 (see main): same constructor / reset / set_state / step_async /
    _buf / close surface on CPU tensors; a step sleeps (longer on higher ranks, so that the MAX over ranks is visible).  Not an engine."""
    import torch

    class Rehearsal:
        def __init__(self, n, has_object=False, controller_type="joint", device=None, env_id_offset=0, **kw):
            self.n, self.action_dim, self.env_id_offset = n, 7, env_id_offset
            self._buf = {"ep_return": torch.full((n,), float(rank + 1), dtype=torch.float64), "ep_length": torch.full((n,), 50, dtype=torch.int32),
                         "is_success": torch.zeros(n, dtype=torch.bool), "truncated": torch.ones(n, dtype=torch.bool)}
        def reset(self, seed=None): pass
        def set_state(self, **kw): pass
        def step_async(self, a): time.sleep(1e-3 * (rank + 1))
        def close(self): pass
    return Rehearsal


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--case", default=None, choices=sorted(CASES), help="workload by name (overrides --task/--controller/--dr/--scripted-grasp)")
    ap.add_argument("--controller", default="joint", choices=["joint", "IK", "mocap"])
    ap.add_argument("--task", default="reach", choices=["reach", "pnp"],
                    help="reach = BASELINE configs[1] (the headline metric); pnp = configs[2], PickAndPlace with contacts")
    ap.add_argument("--dr", action="store_true", help="PickAndPlace with per-reset domain randomisation (configs[4])")
    ap.add_argument("--scripted-grasp", action="store_true",
                    help="PickAndPlace from the 'gripper closing over the cube' state (pad contacts in the timed window)")
    ap.add_argument("--lockstep", action="store_true", help="all episodes start together (every env resets on the same step)")
    ap.add_argument("--envs-per-gpu", type=int, default=N_ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()
    case = args.case
    if case is None:
        case = args.task + "-" + args.controller + ("-dr" if args.dr else "") + ("-grasp" if args.scripted_grasp else "")
        if case not in CASES:
            raise SystemExit(f"no such workload: {case} (have {sorted(CASES)})")

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box only): ranks share device 0 and rendezvous over gloo, because RCCL refuses
    # two ranks on one device; the real multi-GPU run is one rank per GPU over RCCL ("nccl")
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    # BENCH_REHEARSE_PLUMBING=1 (tests/test_bench_plumbing.py, CPU only): everything of this function EXCEPT the engine -- rank / world
    # parsing, the --gpus check, shard offsets, barrier + synchronise brackets, the MAX-reduced wall time, the logging collective, JSON on
    # rank 0 only -- over gloo, with a stand-in that sleeps instead of launching kernels.  Its line carries "rehearsal": true and no value:
    # it measures nothing and exists so that the first real 8-GPU launch cannot fail on plumbing.
    rehearse = os.environ.get("BENCH_REHEARSE_PLUMBING") == "1"
    if share:
        local_rank = 0
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            if share:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    red_dev = "cpu" if (share or rehearse) else None        # gloo reduces host tensors
    dev = torch.device("cpu") if rehearse else torch.device("cuda", local_rank)
    sync = (lambda: None) if rehearse else (lambda: torch.cuda.synchronize(dev))
    from mycobotgym_amd.sharding import reduce_episode_stats, shard
    if rehearse:
        MyCobotVecEnv = _rehearsal_engine(rank)
    else:
        from mycobotgym_amd import MyCobotVecEnv

    n = args.envs_per_gpu
    K, W = args.steps, args.warmup
    env_offset, total_envs = shard(rank, world, n)

    def run(case, steps, warmup, lockstep):
        task, controller, dr, grasp = CASES[case]
        pnp = task == "pnp"
        envs = MyCobotVecEnv(n, has_object=pnp, controller_type=controller, reward_type="dense", device=dev,
                             seed=0, env_id_offset=env_offset,
                             domain_randomization={"mass": (0.5, 2.0), "friction": (0.5, 1.5)} if dr else None,
                             max_episode_steps=10 ** 9 if grasp else 50)
        envs.reset(seed=0)
        g = torch.Generator(device=dev); g.manual_seed(1234 + rank)
        if not lockstep and not grasp:      # every env at its own point of its episode
            envs.set_state(elapsed=torch.randint(0, 50, (n,), device=dev, generator=g, dtype=torch.int32))
        pool = torch.rand(16, n, envs.action_dim, device=dev, generator=g) * 2 - 1     # resident action batches
        if grasp:
            from mycobotgym_amd.scenarios import grasp_state
            st = grasp_state(n, seed=rank)
            act = torch.as_tensor(st.pop("action"), device=dev)
            envs.set_state(**st)
            pool = act.unsqueeze(0).repeat(16, 1, 1).contiguous()
        for t in range(warmup):
            envs.step_async(pool[t % 16])
        sync()
        if world > 1:
            dist.barrier()
        sync()
        # HIP events on the launch stream (mcg_step enqueues on torch's current stream, so torch events see it)
        if not rehearse:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if not rehearse: ev0.record()
        for t in range(steps):
            envs.step_async(pool[t % 16])
        if not rehearse: ev1.record()
        sync()
        if world > 1:
            dist.barrier()
        sync()
        dt = time.perf_counter() - t0
        kernel_ms = (dt * 1e3 if rehearse else ev0.elapsed_time(ev1)) / steps      # average launch-to-launch duration over the timed region
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=red_dev or dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        # logging path: episode statistics of the last step reduced over ranks (RCCL), off the timed region
        b = envs._buf
        stats = reduce_episode_stats(b["ep_return"], b["ep_length"], b["is_success"], b["truncated"], device=red_dev)
        envs.close()
        return dt, kernel_ms, stats

    def entry(case, steps, warmup, lockstep):
        dt, kernel_ms, stats = run(case, steps, warmup, lockstep)
        sub = 100 if CASES[case][1] == "IK" else 20
        v = total_envs * steps / dt
        return {"case": case, "workload": describe(case, n, lockstep), "env_steps_per_sec": v, "ms_per_step": dt / steps * 1e3,
                "steps": steps, "warmup": warmup, "physics_substeps_per_sec": v * sub,
                "roofline": roofline(case, n, kernel_ms) if (rank == 0 and not rehearse) else None, "episode_stats_last_step": stats}

    def api_step_cost(case, steps):
        """The packaged Gymnasium-style step() (fresh copies of every output, info dict) against the raw launch: what a caller of
        MyCobotVecEnv.step pays on top of the kernel."""
        task, controller, dr, grasp = CASES[case]
        envs = MyCobotVecEnv(n, has_object=task == "pnp", controller_type=controller, reward_type="dense", device=dev, seed=0)
        envs.reset(seed=0)
        g = torch.Generator(device=dev); g.manual_seed(7)
        pool = torch.rand(16, n, envs.action_dim, device=dev, generator=g) * 2 - 1
        out = {}
        for name, fn in (("raw_step_async_ms", lambda a: envs.step_async(a)), ("step_copy_false_ms", lambda a: envs.step(a, copy=False)),
                         ("step_ms", lambda a: envs.step(a))):
            for t in range(20): fn(pool[t % 16])
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            for t in range(steps): fn(pool[t % 16])
            torch.cuda.synchronize(dev); out[name] = (time.perf_counter() - t0) / steps * 1e3
        envs.close()
        return out

    main_e = entry(case, K, W, args.lockstep)
    task, controller = CASES[case][0], CASES[case][1]
    out = {
        "metric": "env-steps/sec (whole node), MyCobot Reach, N_envs=8192/GPU" if task == "reach" else
                  "env-steps/sec (whole node), MyCobot PickAndPlace, N_envs=8192/GPU",
        "value": main_e["env_steps_per_sec"], "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": main_e["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": main_e["workload"], "case": case, "envs_per_gpu": n, "total_envs": total_envs,
                   "controller": controller, "episodes": "lockstep" if args.lockstep else "desynchronised",
                   "parallelism": f"env-sharded x{world}, no step-path collective"},
        "physics_substeps_per_sec": main_e["physics_substeps_per_sec"],
        "roofline": main_e["roofline"], "episode_stats_last_step": main_e["episode_stats_last_step"],
    }
    if rehearse:
        out.update(value=None, rehearsal=True, data="none: plumbing rehearsal without the engine (BENCH_REHEARSE_PLUMBING=1)", roofline=None,
                   rehearsal_detail={"rank0_env_offset": env_offset, "total_envs": total_envs, "max_over_ranks_wall_s": main_e["ms_per_step"] * K / 1e3})
    if world == 1 and not args.no_secondary and case == "reach-joint" and not args.lockstep and not rehearse:
        # bounded so that the default run still finishes within a few minutes (the grasp case runs ~10 ms per step)
        k2, w2 = max(min(K, 200) // 2, 20), max(min(W, 100) // 2, 5)
        out["lockstep"] = {k: v for k, v in entry(case, K, W, True).items() if k != "roofline"}
        # random-policy contact workloads: at least 60 untimed steps first, so that every env has passed a reset at its own time and the
        # share of arms lying on the table is the stationary one (straight after reset() nothing touches anything: too flattering)
        warm_of = lambda c: 5 if c == "pnp-joint-grasp" else (max(w2, 60) if c in ("pnp-IK", "pnp-mocap") else w2)
        out["secondary"] = [entry(c, k2 if c != "pnp-joint-grasp" else min(k2, 40), warm_of(c), False) for c in DEFAULT_SECONDARY]
        out["api_step_cost"] = api_step_cost(case, 200)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1 and task == "reach" and not rehearse:
            out["cpu_baseline"] = cpu_baseline(controller)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
