#!/usr/bin/env python3
"""Development aid (-DMCG_COOP_DEBUG build): the first cooperative solve of every environment of workgroup 0 under the two routings
(one environment per wave / two), compared entry by entry at every sub-step of a scripted grasp: first Newton system H, g, candidate x,
result a.  Stops at the first sub-step where they differ.

    MCG_LIB=ab/pair_dbg.so python tools/coop_debug.py [sub-steps]
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from tests.common import make_pair, sync_oracle_to, step_errors, make_oracle
from tests.test_gpu_pickandplace import _grasp_state
from mycobotgym_amd import _abi, MyCobotVecEnv

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n = 32
L = _abi.load()
L.mcg_debug_coop_dump.argtypes = [C.POINTER(C.c_double), C.c_int]
kw = dict(has_object=True, controller_type="joint", reward_type="dense", seed=5, frame_skip=1, max_episode_steps=10 ** 9)
os.environ["MCG_COOP_PAIR"] = "0"
es, ora = make_pair(n, **kw)
os.environ["MCG_COOP_PAIR"] = "1"
ep = MyCobotVecEnv(n, **kw)
for e in (es, ep): e.reset(seed=5)
ora.reset(seed=5)
_grasp_state(ora, n)
a = np.clip(np.tile(np.concatenate([ora.get_state()["ctrl"][0, :6], [1.0]]).astype(np.float32), (n, 1)), -1, 1)
np.set_printoptions(linewidth=250, precision=3)
def dump():
    buf = (C.c_double * (32 * 512))(); L.mcg_debug_coop_dump(buf, 0); return np.array(buf).reshape(32, 512)
for t in range(steps):
    sync_oracle_to(es, ora); sync_oracle_to(ep, ora)
    L.mcg_debug_coop_dump(None, 1)
    os_, _, _, _, _ = es.step(torch.as_tensor(a)); torch.cuda.synchronize(); S = dump()
    L.mcg_debug_coop_dump(None, 1)
    op_, _, _, _, _ = ep.step(torch.as_tensor(a)); torch.cuda.synchronize(); P = dump()
    o = ora.step(a)
    errs = np.abs(os_["observation"].cpu().numpy() - o["obs"]).max(axis=1); errp = np.abs(op_["observation"].cpu().numpy() - o["obs"]).max(axis=1)
    ncon = [int(ora.data(i).get("ncon", (1,), np.int32)[0]) for i in range(n)]
    worst = 0; bad = []
    for e in range(n):
        da = np.abs(S[e][380:398] - P[e][380:398]).max()
        if da > 1e-8 or errp[e] > 1e-9: bad.append(e)
    print(f"t {t}: obs err single {errs.max():.1e} pair {errp.max():.1e}; ncon {sorted(set(ncon))}; bad {bad}")
    if bad:
        for e in bad[:3]:
            ds, dp = S[e], P[e]
            dH = np.abs(ds[:324] - dp[:324]).reshape(18, 18)
            print(f" env {e}: ncon(oracle, after) {ncon[e]}; |H| {np.abs(ds[:324]).max():.2e} dH {dH.max():.2e} at {np.unravel_index(dH.argmax(), dH.shape)}; dg {np.abs(ds[324:342] - dp[324:342]).max():.2e}; da_in {np.abs(ds[360:378] - dp[360:378]).max():.1e}; dx {np.abs(ds[342:360] - dp[342:360]).max():.2e}; d(result) {np.abs(ds[380:398] - dp[380:398]).max():.2e}")
            print("  x single", ds[342:360]); print("  x pair  ", dp[342:360]); print("  a single", ds[380:398]); print("  a pair  ", dp[380:398])
            for it in range(8):
                ts, tp = ds[400 + 8 * it:408 + 8 * it], dp[400 + 8 * it:408 + 8 * it]
                if ts[0] or tp[0]: print(f"  it {it}: single nact {ts[0]:.0f} same {ts[1]:.0f} ls {ts[2]:.0f} alpha {ts[3]:.12g} s0 {ts[4]:.6g} quad {ts[5]:.6g} | pair nact {tp[0]:.0f} same {tp[1]:.0f} ls {tp[2]:.0f} alpha {tp[3]:.12g} s0 {tp[4]:.6g} quad {tp[5]:.6g}")
            for b in range(9):
                print(f"   ls eval {b}: single alp {ds[464 + 3 * b]:.12g} f {ds[465 + 3 * b]:.6g} sl {ds[466 + 3 * b]:.6g} | pair alp {dp[464 + 3 * b]:.12g} f {dp[465 + 3 * b]:.6g} sl {dp[466 + 3 * b]:.6g}")
            print("  dH row max:", dH.max(axis=1)); print("  g single", ds[324:342]); print("  g pair  ", dp[324:342])
        break
