// mcg_coop.hpp -- the coupled robot + cube solve of PickAndPlace, one environment per 32 lanes (two per wave on the 64-lane waves).
//
// Replaces, for an environment in which a contact reaches the robot (finger pad / finger link on the cube, pad or arm mesh on the
// table / the ground), MuJoCo's mj_fwdConstraint over the 18 dofs (P9; call sites /root/reference/mycobotgym/envs/mycobot.py:170,193;
// contact geoms mycobot280_main.xml:81,87,105-175,195-199,222-225,260-265).
//
// Why it exists.  The lane-parallel coupled solve (mcg_cube.hpp: one env per lane) makes all 32 environments of a wave walk the
// longest contact list for the largest Newton iteration count whenever ONE of them touches something, and every instruction of that
// walk waits on the previous one's memory round trip (one wave per SIMD): 355 us per coupled sub-step whatever the number of touching
// environments.  Here the environments are routed one by one: the robot wave commits its speculative contact-free sub-step for the
// environments that touch nothing, and each touching environment is solved by ONE WAVE WORKING ON IT ALONE -- the four waves of the
// workgroup (robot, cube, M, RNE; all idle at that point of the sub-step) take the flagged environments in turn.
//
// The solve is the same primal Newton iteration as the oracle's (oracle/mco_physics.c: fwd_constraint), written densely over GENERIC
// rows:    minimise  1/2 a^T H0 a - g0^T a + sum_r 1/2 D_r min(0, J_r a - aref_r)^2,    a = (robot 12, cube 6),
// H0 = blockdiag(M + J_eq^T D J_eq, cube inertia), rows r = joint limits (10) and the pyramid rows of the contacts (6 or 4 each).
//   * lanes = ROWS for everything per row: the row's Jacobian (18 numbers in registers, built once per solve from the contact's frame and
//     the twist columns of the joints in its chain), residual, active bit (ballot), line-search terms;
//   * lanes = MATRIX ROWS (lane i < 18 holds row i of H) for the assembly H = H0 + sum_active D J J^T (the active rows pass through a
//     16-row LDS window), the L D L^T factorisation (pivot column broadcast lane by lane with v_readlane: no memory round trip on the
//     dependent chain) and the two triangular solves (L^T through one LDS transpose);
//   * a candidate x = H^-1 g that keeps the assumed active set IS the minimiser; otherwise an exact line search on the piecewise
//     quadratic (Newton on phi', two 32-lane DPP reductions per evaluation) and the next iteration.
// Any contact between any two of {static geom, cube, arm body b, finger body of a side} is a row here: there are no contact classes
// and no per-class accumulators (what kept the gripper base - cube pair out of round 2's lane-parallel solve: it is enabled now).
#pragma once

#include "mcg_mesh.hpp"

namespace mcg {

// ---- exchange slots of the four-wave kernel (all per-lane columns).
// Clip-polygon area, first 32 slots (the collision pass clips there, before barrier S2; afterwards:) flags and the cube's hand-over.
// XCH_FLAG: 0 = no contact reaches the robot | 2 = one does (a pad or a mesh on the cube, the table or the ground), or the cube alone
// holds more contacts than its own solve has row slots for: the environment's 18 dofs go to the cooperative solve.
constexpr int XCH_FLAG = LDS_POLY, XCH_T0 = LDS_POLY + 1, XCH_T1 = LDS_POLY + 2, XCH_NCON = LDS_POLY + 3;
constexpr int XCH_CB = LDS_POLY + 4, XCH_QL7 = LDS_POLY + 23, XCH_DR = LDS_POLY + 30;      // cube: pos 3, quat 4, vel 6, warm 6; lagged pose 7; DR scales 2
static_assert(XCH_DR + 2 <= LDS_POLY + 32, "exchange area exceeds the clip-polygon slots");
// ... second 32 slots, which no pass writes: q(t), qd(t) of the robot for the other waves -- written at the end of a sub-step, read after
// S1 by the M, RNE and cube waves and between S4 and S5 by the cooperative solves -- and the workgroup's hand-out counter of that phase
// (an unsigned in the first word of lane 0's slot).
constexpr int XCH_Q = LDS_POLY + 32, XCH_QD = XCH_Q + NB, COOP_CTR_SLOT = XCH_QD + NB;
// XCH_T1 (robot wave -> cube wave, read after S1) / XCH_BADC (cube wave -> robot wave, read after S4): the other body failed mj_checkPos /
// mj_checkVel, mj_resetData resets both
constexpr int XCH_BADC = COOP_CTR_SLOT + 1;
// XCH_ACT0 / XCH_ACT1: the active set the environment's last cooperative solve ended with, as raw bits -- [signature of the contact list |
// rows 0-31], [rows 32-63 | rows 64-95] -- cleared at the start of an env-step.  The next sub-step's solve starts from it (see coop_guess).
constexpr int XCH_ACT0 = XCH_BADC + 1, XCH_ACT1 = XCH_ACT0 + 1;
static_assert(XCH_ACT1 + 1 <= LDS_POLY + 62, "exchange area (slots 62, 63: a stage-clock census)");
// Row area (LDS_ROW .. LDS_ROW + 144 slot rows of PNP_LANES doubles):
//   between S1 and S1c: the frames of the bodies that carry candidate meshes (MP_FRAME, mcg_cube.hpp);
//   after S2: [0, 144) the cube wave's own solves (12 slots per contact, list positions 0..11: a lane whose list is longer is flagged);
//   between barriers S4 and S5, when every lane-parallel solve is over: [0, 96) cooperative workspace, 768 doubles per wave.
// The inputs of a lane's cooperative solve, parked by the robot wave (PubHook) -- right after S1b when the workgroup has a mesh phase to
// overlap with (then for every lane: the flags do not exist yet), else after S2 for the flagged lanes -- live in slots that are free from
// S1b to S5: g0 in the robot's q(t) slots (read for the last time before S1b, rewritten when the sub-step ends), the warm start in the
// bias-force slots (the robot wave has taken them; the RNE wave rewrites them after the next S1), the limit rows in the cube-alone
// solve's mask slots (a flagged lane has no such solve; an unflagged lane's solve initialises its masks) and four slots of their own.
constexpr int PUB_G0 = XCH_Q, PUB_WARM = PNP_SLOTS /* = XCH_FS */, PUB_SD = LDS_ACT, PUB_AREF_A = LDS_ACT + 10, PUB_AREF_B = PNP_SLOTS_ALL;
MCG_DEV constexpr int pub_aref(int j) { return j < 6 ? PUB_AREF_A + j : PUB_AREF_B + (j - 6); }
constexpr int PNP_SLOTS_LDS = PUB_AREF_B + 4;
static_assert(MAXCON >= 16, "the limit rows borrow sixteen mask slots");
constexpr int COOP_WS_ROW = LDS_ROW, COOP_WS_DOUBLES = 768;
static_assert(4 * COOP_WS_DOUBLES <= 96 * PNP_LANES, "cooperative workspace exceeds the first 96 rows of the row area");
constexpr int COOP_NV = 18, COOP_WIN = 16, COOP_WSTRIDE = 20;          // window: 16 active rows x (D, D aref, J[18])
static_assert(COOP_WIN * COOP_WSTRIDE <= COOP_WS_DOUBLES && COOP_NV * COOP_NV + COOP_NV <= COOP_WS_DOUBLES, "window / matrix buffer");
constexpr int COOP_ROWS = 10 + 6 * MAXCON, COOP_SETS = (COOP_ROWS + PNP_LANES - 1) / PNP_LANES;      // limits first, then the contacts
#ifndef MCG_COOP_FULL_STEPS
#define MCG_COOP_FULL_STEPS 1
#endif

// ---- cross-lane helpers (32 active lanes = two DPP rows of 16)
MCG_DEV real coop_rdlane(real v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
template <int CTRL> MCG_DEV real coop_dpp(real v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
MCG_DEV real coop_sum16(real v) {      // sum over each 16-lane DPP row, in every lane of the row
  v += coop_dpp<0xB1>(v); v += coop_dpp<0x4E>(v); v += coop_dpp<0x141>(v); v += coop_dpp<0x140>(v);
  return v;
}
MCG_DEV real coop_sum32(real v) {      // sum over the 32 active lanes, the same (uniform) number in every lane
  v += coop_dpp<0xB1>(v);              // quad_perm [1,0,3,2]
  v += coop_dpp<0x4E>(v);              // quad_perm [2,3,0,1]
  v += coop_dpp<0x141>(v);             // row_half_mirror
  v += coop_dpp<0x140>(v);             // row_mirror: every lane of a 16-lane row holds the row's sum
  return coop_rdlane(v, 0) + coop_rdlane(v, 16);
}
template <int K> MCG_DEV real coop_bcast16(real v) {      // lane K of each 16-lane DPP row to all lanes of that row: one v_mov_b64_dpp
  return __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xF, 0xF, false);      // (old = v: every lane is written, nothing to preset)
}
MCG_DEV void coop_lds_sync() {         // LDS traffic between lanes of one wave: order the compiler (the hardware keeps a wave's LDS ops in order)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- the first iteration's active set (coop_guess).  The oracle's Newton iteration starts from the rows violated at the warm start
// (last sub-step's acceleration).  Those are a poor predictor here: a contact row's residual at its solution is -f / D, micrometres per
// second squared, and aref moves by far more than that from one sub-step to the next, so the sign of a stiff row's residual at the warm
// start is close to a coin toss and the iteration needs 3-5 rounds to find the set again.  What does persist while a contact persists is
// WHICH rows carry force.  So a solve leaves its final active set behind, and the next sub-step's solve of the same environment -- if
// its contact list has the same pairs in the same order -- assembles its first system over that set.  The stopping rule is untouched: a
// candidate x = H^-1 g is accepted only if the rows it violates are exactly the rows it was assembled over, which is the KKT condition
// of the (strictly convex) problem; a wrong guess costs one iteration and the iteration carries on from its candidate as before.
MCG_DEV unsigned coop_sig_step(unsigned sig, int type) { return sig * 31u + (unsigned)type + 1u; }
MCG_DEV real coop_pack(unsigned hi, unsigned lo) { return __hiloint2double((int)hi, (int)lo); }
#ifndef MCG_COOP_GUESS
#define MCG_COOP_GUESS 1
#endif

// Stage clocks of the cooperative solve (-DMCG_STAGE_CLOCKS): accumulated in registers, added to the global table once per coop_phase
// (one atomic per tick from a thousand waves would itself be the largest stage).
#ifdef MCG_STAGE_CLOCKS
struct CoopClocks { unsigned long long last, t[ST_CO_IDLE - ST_CO_SETUP]; unsigned n[8]; };      // n: solves, iterations, line searches, active rows, evaluations, solves of >= 8 iterations, solves that hit the cap, 12-dof solves
#define COOP_COUNT(k, v) do { CK.n[k] += (unsigned)(v); } while (0)
#define COOP_TICK(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); CK.t[(k) - ST_CO_SETUP] += t_ - CK.last; CK.last = t_; } while (0)
#else
struct CoopClocks {};
#define COOP_TICK(k) do {} while (0)
#define COOP_COUNT(k, v) do {} while (0)
#endif

MCG_DEV void coop_flush_clocks(const CoopClocks& CK) {
#ifdef MCG_STAGE_CLOCKS
  if ((threadIdx.x & 63) == 0) {
    if (CK.n[0]) atomicAdd(&g_wg_stat[(blockIdx.x & 4095) * 4 + 1], (unsigned long long)CK.n[0]);
    for (int k = 0; k < ST_CO_IDLE - ST_CO_SETUP; k++) if (CK.t[k]) atomicAdd(&g_stage_clocks[ST_CO_SETUP + k], CK.t[k]);
    const int slot[8] = {CN_COUPLED, CN_COUPLED_IT, CN_COUPLED_LS, CN_COOP_ROWS, CN_COOP_LSEVAL, CN_COOP_LONG, CN_COOP_CAP, CN_COOP_12};
    for (int k = 0; k < 8; k++) if (CK.n[k]) atomicAdd(&g_stage_clocks[ST_COUNT + slot[k]], (unsigned long long)CK.n[k]);
  }
#else
  (void)CK;
#endif
}

#ifdef MCG_COOP_DEBUG
// development aid: the first solve of every environment of workgroup 0 leaves its first Newton system, candidate and result here
__device__ double g_coop_dbg[32 * 512];
__device__ int g_coop_dbg_done[32];
#endif

// ---- one row of the coupled problem (shared by the two solvers).  Row r < 10: the limit of joint r (from the robot wave's parked numbers);
// else pyramid row (r - 10) % 6 of contact (r - 10) / 6.  A contact's Jacobian row is built from its frame and the TWIST COLUMNS tc[j] of
// the joints in the chain of the pair's robot body: J_r[j] = +- e_r . c_j, e_r = [d ; lever x d + tau n], c_j = [(anchor_j - p0) x axis_j ;
// axis_j] (p0 = the cube's centre), and [d ; Rc^T (...)] in the cube's six dofs.  Any pair type is a row: there are no contact classes.
template <class CSYS>
MCG_DEV void coop_build_row(const PnpScratch ME, const CSYS& CS, const real* cpos, const real (*tc)[6], const real* qd, const real* vc, int r, int ncon,
                            real* J, real& D, real& ar, int& rowcls) {
  const bool is_lim = r < 10;
  // limit row: J = sg e_j, D, aref from the robot wave
  const int jl = sel(is_lim, r, 0);
  const real sD = ME.ld(PUB_SD + jl), al = ME.ld(sel(jl < 6, PUB_AREF_A + jl, PUB_AREF_B + jl - 6));
  const real sgn = sel(sD > 0, 1.0, sel(sD < 0, -1.0, 0.0));
  // contact row
  const int rc = sel(is_lim, 0, r - 10);
  const int c = sel(rc / 6 < MAXCON, rc / 6, MAXCON - 1), p = rc % 6;
  const bool is_con = !is_lim && (rc / 6 < ncon);
  const int b = LDS_CON + c * CON_STRIDE;
  real lev[3], n[3], t1[3], t2[3];
  _Pragma("unroll") for (int k = 0; k < 3; k++) { lev[k] = ME.ld(b + k) - cpos[k]; n[k] = ME.ld(b + 3 + k); }
  const real Dc = ME.ld(b + CON_D), kterm = ME.ld(b + CON_KTERM), ftype = ME.ld(b + CON_TYPE);
  MCG_FENCE();                                                                 // the row's loads are issued together
  { const bool unit = is_con; real nn[3] = {sel(unit, n[0], 0.0), sel(unit, n[1], 0.0), sel(unit, n[2], 1.0)}; make_frame(nn, t1, t2); }      // (rows that are no contact: a defined frame)
  const int type = sel(is_con, (int)ftype, 0);
  const bool padc = type == PAIR_PADR_CUBE || type == PAIR_PADL_CUBE, mcube = pair_mesh_cube(type);
  const bool mstat = pair_mesh_static(type), tabp = (type == PAIR_TABLE_PADR || type == PAIR_TABLE_PADL);
  const bool has_cube = pair_has_cube(type);
  const int rb = pair_robot_body(type);                                        // the robot body of the pair (-1: none)
  const real rsign = sel(has_cube, -1.0, 1.0);                                 // robot geom is geom1 against the cube, geom2 against a static geom
  real mu[3]; real Bc;
  _Pragma("unroll") for (int k = 0; k < 3; k++)
    mu[k] = sel(mstat, CS.mu_tl[k], sel(tabp, CS.mu_tp[k], sel(mcube, CS.mu_mc[k], sel(padc, CS.mu_pc[k], CS.mu_tc[k]))));
  Bc = sel(mstat, CS.B_tl, sel(tabp, CS.B_tp, sel(mcube, CS.B_mc, sel(padc, CS.B_pc, CS.B_tc))));
  const int kf = p >> 1;
  const real m = sel((p & 1), -1.0, 1.0) * sel3(kf, mu[0], mu[1], mu[2]);
  const bool absent = mstat && kf == 2;                                        // condim 3: no torsional pair of rows
  real dlin[3], eang[3];
  _Pragma("unroll") for (int k = 0; k < 3; k++) dlin[k] = n[k] + sel(kf == 0, m * t1[k], sel(kf == 1, m * t2[k], 0.0));
  cross(lev, dlin, eang);
  const real tau = sel(kf == 2, m, 0.0);
  _Pragma("unroll") for (int k = 0; k < 3; k++) eang[k] = fma(tau, n[k], eang[k]);
  const bool live = is_con && !absent;
  // robot columns: joints of the chain
  _Pragma("unroll") for (int j = 0; j < NB; j++) {
    const bool member = live && joint_in_chain(j, rb);
    const real dj = dlin[0] * tc[j][0] + dlin[1] * tc[j][1] + dlin[2] * tc[j][2] + eang[0] * tc[j][3] + eang[1] * tc[j][4] + eang[2] * tc[j][5];
    J[j] = sel(member, rsign * dj, 0.0);
  }
  // cube columns: [dlin ; Rc^T eang]
  _Pragma("unroll") for (int k = 0; k < 3; k++) {
    J[12 + k] = sel(live && has_cube, dlin[k], 0.0);
    J[15 + k] = sel(live && has_cube, CS.Rc[k] * eang[0] + CS.Rc[3 + k] * eang[1] + CS.Rc[6 + k] * eang[2], 0.0);
  }
  real vel = 0;
  _Pragma("unroll") for (int j = 0; j < NB; j++) vel = fma(J[j], qd[j], vel);
  _Pragma("unroll") for (int k = 0; k < 6; k++) vel = fma(J[12 + k], vc[k], vel);
  D = sel(live, Dc, 0.0); ar = sel(live, -Bc * vel - kterm, 0.0);
  // limit row over the top
  _Pragma("unroll") for (int j = 0; j < NB; j++) J[j] = sel(is_lim, sel(j == jl, sgn, 0.0), J[j]);
  D = sel(is_lim, fabs(sD), D); ar = sel(is_lim, al, ar);
  rowcls = sel(is_lim, 0, sel(has_cube, 2, 1));
}

// the twist columns of the twelve joints about the cube's centre, from the world axes / anchors the collision pass left (LDS_WJ); the
// gripper's six are posed only when the gripper is near something (stale LDS otherwise): `grip` = a contact of the list rides on them
MCG_DEV void coop_twists(const PnpScratch ME, const real* cpos, bool grip, real (*tc)[6]) {
  _Pragma("unroll") for (int j = 0; j < NB; j++) {
    real ax[3], d[3], v[3];
    _Pragma("unroll") for (int k = 0; k < 3; k++) { ax[k] = ME.ld(LDS_WJ + j * 6 + k); d[k] = ME.ld(LDS_WJ + j * 6 + 3 + k) - cpos[k]; }
    cross(d, ax, v);
    const bool ok = (j < 6) || grip;
    _Pragma("unroll") for (int k = 0; k < 3; k++) { tc[j][k] = sel(ok, v[k], 0.0); tc[j][3 + k] = sel(ok, ax[k], 0.0); }
  }
}

constexpr unsigned coop_row_mask(const Pattern& P, int i) { unsigned m = 0; for (int j = 0; j <= i; j++) m |= P.nz[i][j] ? (1u << j) : 0u; return m; }

// One environment's coupled solve, by the 32 active lanes of the calling wave.  lds0 = slot 0 of lane 0 of the workgroup's array,
// e = the environment's lane (its LDS column), ws = this wave's workspace.
template <int NSETS>
MCG_DEV void coop_solve(ModelPtr Pm, LdsPtr lds0, int e, LdsPtr ws, int ncon, CoopClocks& CK) {
  const int L = threadIdx.x & (PNP_LANES - 1);
  const PnpScratch ME(lds0 + e);                   // env e's column: a uniform slot index is a broadcast read
  COOP_COUNT(0, 1);
#ifdef MCG_STAGE_CLOCKS
  CK.last = __builtin_readcyclecounter();
#endif
  // ---- the environment's cube and pair numbers (uniform), as the lane-parallel solve derives them
  Cube Cb; real drs[2];
  for (int k = 0; k < 3; k++) Cb.pos[k] = ME.ld(XCH_CB + k);
  for (int k = 0; k < 4; k++) Cb.quat[k] = ME.ld(XCH_CB + 3 + k);
  for (int k = 0; k < 6; k++) { Cb.vel[k] = ME.ld(XCH_CB + 7 + k); Cb.warm[k] = ME.ld(XCH_CB + 13 + k); }
  drs[0] = ME.ld(XCH_DR); drs[1] = ME.ld(XCH_DR + 1);
  CubeSys<PnpScratch> CS(ME, Cb, drs);          // (for its pair numbers only)
  CS.pm_bits = (unsigned long long)Pm;
  CS.derive(Pm);
  // ---- does a contact ride on the gripper's bodies (their joint frames are posed then); twist columns of the twelve joints about the cube centre
  bool grip = false;
  unsigned sig = (unsigned)ncon;
  for (int c = 0; c < ncon; c++) {
    const int type = (int)ME.ld(LDS_CON + c * CON_STRIDE + CON_TYPE);
    grip = grip || pair_robot_body(type) >= 6;
    sig = coop_sig_step(sig, type);
  }
  sig |= 0x80000000u;
  unsigned guess[COOP_SETS]; bool have_guess;
  { const real p0 = ME.ld(XCH_ACT0), p1 = ME.ld(XCH_ACT1);
    have_guess = MCG_COOP_GUESS && NSETS <= 3 && (unsigned)__double2hiint(p0) == sig;
    guess[0] = (unsigned)__double2loint(p0); guess[1] = (unsigned)__double2hiint(p1); guess[2] = (unsigned)__double2loint(p1); guess[3] = 0u; }
  unsigned fin[COOP_SETS] = {0u, 0u, 0u, 0u}; bool conv = false;
  COOP_COUNT(7, have_guess ? 1 : 0);
  real tc[NB][6];
  coop_twists(ME, Cb.pos, grip, tc);
  real qd[NB], vc[6];
  _Pragma("unroll") for (int j = 0; j < NB; j++) qd[j] = ME.ld(XCH_QD + j);
  _Pragma("unroll") for (int k = 0; k < 6; k++) vc[k] = Cb.vel[k];

  COOP_TICK(ST_CO_SETUP);
  // ---- rows.  Row r = 32 s + L of set s: r < 10 the limit of joint r; else pyramid row (r - 10) % 6 of contact (r - 10) / 6.
  const int nrows = 10 + 6 * ncon;
  const int nsets = (nrows + PNP_LANES - 1) / PNP_LANES;        // uniform; 1 .. NSETS
  real J[NSETS][COOP_NV], Dr[NSETS], aref[NSETS];
  _Pragma("unroll") for (int s = 0; s < NSETS; s++) {
    _Pragma("unroll") for (int j = 0; j < COOP_NV; j++) J[s][j] = 0;
    Dr[s] = 0; aref[s] = 0;
    if (s < nsets) { int cls; coop_build_row(ME, CS, Cb.pos, tc, qd, vc, PNP_LANES * s + L, ncon, J[s], Dr[s], aref[s], cls); (void)cls; }
  }

  MCG_TICK_PIN(Dr, NSETS); MCG_TICK_PIN(aref, NSETS);
  COOP_TICK(ST_CO_ROWS);
  // ---- H0 and g0, lane i < 18 holds row i (lanes 18.. shadow row 17 and are never read)
  const int i = sel(L < COOP_NV, L, COOP_NV - 1);
  real H0[COOP_NV], g0;
  {
    unsigned mE = 0, mM = 0;                     // pattern of the lane's own row (columns <= i)
    static_for<NB>([&](auto I) { constexpr int k = I; mE = sel(i == k, coop_row_mask(PAT_E, k), mE); mM = sel(i == k, coop_row_mask(PAT_M, k), mM); });
    const bool rob = i < NB;
    const int ir = sel(rob, i, 0);
    real he[NB], hm[NB];                         // all loads first (the column's slots share one LDS bank: every one of them is slow)
    static_for<NB>([&](auto Jj) { constexpr int j = Jj;
      // entry (i, j): below the diagonal in row i (lane-varying pattern bit), above it in row j (static row, lane-varying column)
      const int slot = sel(ir >= j, ir * (ir + 1) / 2 + j, j * (j + 1) / 2 + ir);
      he[j] = ME.ld(LDS_HEQ + slot); hm[j] = ME.ld(LDS_M + slot); });
    MCG_FENCE();
    static_for<NB>([&](auto Jj) { constexpr int j = Jj;
      const bool low = ir >= j;
      const bool nzE = sel(low, ((mE >> j) & 1u) != 0u, ((coop_row_mask(PAT_E, j) >> ir) & 1u) != 0u);
      const bool nzM = sel(low, ((mM >> j) & 1u) != 0u, ((coop_row_mask(PAT_M, j) >> ir) & 1u) != 0u);
      H0[j] = sel(rob, sel(nzE, he[j], 0.0) + sel(nzM, hm[j], 0.0), 0.0); });
    static_for<6>([&](auto Kk) { constexpr int k = Kk; H0[NB + k] = sel(i == NB + k, CS.Md[k], 0.0); });
    const real gr = ME.ld(PUB_G0 + ir);
    real gcv = 0;
    static_for<6>([&](auto Kk) { constexpr int k = Kk; gcv = sel(i == NB + k, CS.fs[k], gcv); });
    g0 = sel(rob, gr, gcv);
  }

  MCG_TICK_PIN(H0, COOP_NV);
  COOP_TICK(ST_CO_H0);
  // ---- the iterate lives one number per lane (lane i holds a_i); a uniform copy of a vector is made where one is needed, by v_readlane
  real al = ME.ld(sel(i < NB, PUB_WARM + i, XCH_CB + 13 + (i - NB)));
  // Assembly layout: all 32 lanes work on the 16 x 18 upper block of the increment, lane (ia, hb) on row ia, columns 9 hb .. 9 hb + 8;
  // rows 16 and 17 come from the symmetry (columns 16, 17 of the hb = 1 lanes) and their 2 x 2 corner from the broadcast values every
  // hb = 1 lane holds anyway.  One active row costs a lane 7 LDS reads and 18 flops (the row layout: 21 and 40).
  const int ia = L & 15, hb = L >> 4, cb = 9 * hb;
  // A window row is [D, D aref, J_0 .. J_17].  A lane reads TWO numbers of it -- its own entry J_t[ia] and one of the eleven its half-row
  // needs (D, D aref, J_t[9 hb .. 9 hb + 8], lane l of the 16-lane DPP row holds the l-th) -- and the eleven reach every lane of the
  // half by v_mov_b64_dpp row_newbcast: the four waves of the workgroup share one LDS pipe, the VALU is each wave's own.
  const int lseg = L & 15, iseg = lseg + ((hb == 1 && lseg >= 2) ? 9 : 0);
  struct WinRow { real own, seg; };
  auto load_row = [&](int t, WinRow& w) {
    const int o = t * COOP_WSTRIDE;
    w.own = ws[o + 2 + ia]; w.seg = ws[o + iseg];
  };
  struct AsmAcc { real Ah[9], ag, c66, c76, c77, g6, g7; };
  auto add_row = [&](const WinRow& w, AsmAcc& A) {
    const real D = coop_bcast16<0>(w.seg), Da = coop_bcast16<1>(w.seg);
    real jb[9];
    static_for<9>([&](auto Cc) { constexpr int c = Cc; jb[c] = coop_bcast16<2 + c>(w.seg); });
    const real cD = D * w.own;
    _Pragma("unroll") for (int c = 0; c < 9; c++) A.Ah[c] = fma(cD, jb[c], A.Ah[c]);
    A.ag = fma(Da, w.own, A.ag);
    const real d6 = D * jb[7], d7 = D * jb[8];                      // (hb = 1: columns 16 and 17)
    A.c66 = fma(d6, jb[7], A.c66); A.c76 = fma(d7, jb[7], A.c76); A.c77 = fma(d7, jb[8], A.c77);
    A.g6 = fma(Da, jb[7], A.g6); A.g7 = fma(Da, jb[8], A.g7);
  };

  for (int it = 0; it < 50; it++) {
    COOP_COUNT(1, 1);
    if (it == 49) COOP_COUNT(6, 1);
    // (a) residuals and the active set at a; H0 a on the way (the line search's smooth gradient)
    real r0[NSETS], h0a = 0; unsigned act[NSETS];
    _Pragma("unroll") for (int s = 0; s < NSETS; s++) r0[s] = -aref[s];
    static_for<COOP_NV>([&](auto Jj) { constexpr int j = Jj;
      const real aj = coop_rdlane(al, j);
      _Pragma("unroll") for (int s = 0; s < NSETS; s++) r0[s] = fma(J[s][j], aj, r0[s]);
      h0a = fma(H0[j], aj, h0a); });
    const bool use_guess = it == 0 && have_guess;                    // uniform
    _Pragma("unroll") for (int s = 0; s < NSETS; s++) act[s] = (unsigned)__ballot(Dr[s] > 0 && sel(use_guess, ((guess[s] >> L) & 1u) != 0u, r0[s] < 0));
    MCG_TICK_PIN(r0, NSETS);
    COOP_TICK(ST_CO_RESID);
    COOP_COUNT(3, __popc(act[0]) + (NSETS > 1 ? __popc(act[1 % NSETS]) : 0) + (NSETS > 2 ? __popc(act[2 % NSETS]) : 0));
    // (b) H = H0 + sum_active D J J^T, g = g0 + sum_active D aref J: the active rows pass through a 16-row window in LDS, read two
    // rows at a time (all loads of a pair are issued before the first use: one LDS round trip per pair, not one per load)
    AsmAcc A;
    _Pragma("unroll") for (int c = 0; c < 9; c++) A.Ah[c] = 0;
    A.ag = A.c66 = A.c76 = A.c77 = A.g6 = A.g7 = 0;
    _Pragma("unroll") for (int s = 0; s < NSETS; s++) {
      if (act[s] != 0u) {
        const int nact = __popc(act[s]);
        const int pos = __popc(act[s] & ((1u << L) - 1u));
        const bool mine = ((act[s] >> L) & 1u) != 0u;
        for (int w0 = 0; w0 < nact; w0 += COOP_WIN) {
          const int nw = sel(nact - w0 < COOP_WIN, nact - w0, COOP_WIN);
          coop_lds_sync();                                        // the window's previous readers are done
          if (mine && pos >= w0 && pos < w0 + COOP_WIN) {
            const int o = (pos - w0) * COOP_WSTRIDE;
            ws[o] = Dr[s]; ws[o + 1] = Dr[s] * aref[s];
            _Pragma("unroll") for (int j = 0; j < COOP_NV; j++) ws[o + 2 + j] = J[s][j];
          }
          if ((nw & 1) && L < COOP_WSTRIDE) ws[nw * COOP_WSTRIDE + L] = 0.0;      // odd count: a zero row completes the last pair
          coop_lds_sync();
          WinRow wa, wb;                                          // two rows per turn, the next pair's loads in flight meanwhile
          load_row(0, wa); load_row(1, wb);
          for (int t = 0; t < nw; t += 2) {
            const int tn = sel(t + 2 < nw, t + 2, t);               // (the last turn re-reads its own pair)
            WinRow na, nb;
            load_row(tn, na); load_row(tn + 1, nb);
            MCG_FENCE();
            add_row(wa, A); add_row(wb, A);
            MCG_FENCE();
            wa = na; wb = nb;
          }
        }
      }
    }
    // the increment, scattered as a full 18 x 18 matrix (and an 18-vector behind it), gathered row by row: lane i holds row i of H
    coop_lds_sync();
    _Pragma("unroll") for (int c = 0; c < 9; c++) ws[ia * COOP_NV + cb + c] = A.Ah[c];
    if (hb == 1) { ws[16 * COOP_NV + ia] = A.Ah[7]; ws[17 * COOP_NV + ia] = A.Ah[8]; }
    else ws[COOP_NV * COOP_NV + ia] = A.ag;
    if (L == 16) {
      ws[16 * COOP_NV + 16] = A.c66; ws[16 * COOP_NV + 17] = A.c76; ws[17 * COOP_NV + 16] = A.c76; ws[17 * COOP_NV + 17] = A.c77;
      ws[COOP_NV * COOP_NV + 16] = A.g6; ws[COOP_NV * COOP_NV + 17] = A.g7;
    }
    coop_lds_sync();
    real H[COOP_NV], g;
    _Pragma("unroll") for (int j = 0; j < COOP_NV; j++) H[j] = ws[i * COOP_NV + j];
    g = ws[COOP_NV * COOP_NV + i];
    MCG_FENCE();
    _Pragma("unroll") for (int j = 0; j < COOP_NV; j++) H[j] += H0[j];
    g += g0;
#ifdef MCG_COOP_DEBUG
    const bool dbg = blockIdx.x == 0 && it == 0 && g_coop_dbg_done[e] == 0;
    if (dbg && L < COOP_NV && threadIdx.x % 64 < 32) { double* o = g_coop_dbg + e * 512;
      for (int j = 0; j < COOP_NV; j++) o[L * COOP_NV + j] = H[j];
      o[324 + L] = g; o[360 + L] = al; }
#endif
    MCG_TICK_PIN(H, COOP_NV);
    COOP_TICK(ST_CO_ASM);
    // (c) H = L D L^T.  Lane i ends with the UNSCALED column entries H[i][k] = L[i][k] D_k in H[k] and the scaled L[i][k] in Lr[k]
    // (k < i; zero elsewhere, so the substitutions below need no masks); the pivot column travels by v_readlane, 1 / D_k stays uniform
    real Lr[COOP_NV], dinv[COOP_NV];
    static_for<COOP_NV>([&](auto Kk) { constexpr int k = Kk;
      const real dk = coop_rdlane(H[k], k);
      dinv[k] = rcp_nr(dk);
      const real lk = H[k] * dinv[k];
      static_for<COOP_NV - 1 - k>([&](auto Jj) { constexpr int j = k + 1 + Jj;
        const real sj = coop_rdlane(H[k], j);                      // H[j][k] = L[j][k] D_k
        H[j] = fma(-lk, sj, H[j]); });
      Lr[k] = sel(i > k, lk, 0.0); });
    MCG_TICK_PIN(Lr, COOP_NV);
    COOP_TICK(ST_CO_FACTOR);
    // (d) x = H^-1 g.  Forward: y = L^-1 g in row layout (lane k's accumulator is final when its turn comes and is not touched
    // afterwards: Lr[k] is zero there).  Backward on the unscaled columns, D_j x_j = y_j - sum_{j' > j} (L[j'][j] D_j) x_j', which
    // lane j gets through one LDS transpose.  The candidate's row residuals are accumulated as its entries appear.
    real acc = g;
    static_for<COOP_NV>([&](auto Kk) { constexpr int k = Kk;
      const real yk = coop_rdlane(acc, k);
      acc = fma(-Lr[k], yk, acc); });
    coop_lds_sync();
    if (L < COOP_NV) { _Pragma("unroll") for (int k = 0; k < COOP_NV; k++) ws[L * COOP_NV + k] = sel(L > k, H[k], 0.0); }
    coop_lds_sync();
    real U[COOP_NV];                                 // U[j] = H[j][i] (unscaled) for j > i, zero otherwise
    _Pragma("unroll") for (int j = 0; j < COOP_NV; j++) U[j] = ws[j * COOP_NV + i];
    MCG_FENCE();
    real rx[NSETS];
    _Pragma("unroll") for (int s = 0; s < NSETS; s++) rx[s] = -aref[s];
    real xl = 0;
    static_for<COOP_NV>([&](auto Kk) { constexpr int j = COOP_NV - 1 - Kk;
      const real xj = coop_rdlane(acc, j) * dinv[j];
      acc = fma(-U[j], xj, acc);
      xl = sel(i == j, xj, xl);
      _Pragma("unroll") for (int s = 0; s < NSETS; s++) rx[s] = fma(J[s][j], xj, rx[s]); });
#ifdef MCG_COOP_DEBUG
    if (dbg && L < COOP_NV && threadIdx.x % 64 < 32) g_coop_dbg[e * 512 + 342 + L] = xl;
#endif
    MCG_TICK_PIN(rx, NSETS);
    COOP_TICK(ST_CO_SOLVE);
    // (e) does the candidate keep the assumed active set?
    bool same = true;
    _Pragma("unroll") for (int s = 0; s < NSETS; s++) same = same && ((unsigned)__ballot(Dr[s] > 0 && rx[s] < 0) == act[s]);
    COOP_TICK(ST_CO_CHECK);
    COOP_COUNT(5, (use_guess && same) ? 1 : 0);
#ifdef MCG_COOP_DEBUG
    const bool dbt = blockIdx.x == 0 && g_coop_dbg_done[e] == 0 && it < 8 && threadIdx.x % 64 == 0;
    if (dbt) { double* o = g_coop_dbg + e * 512 + 400 + it * 8; o[0] = __popc(act[0]) + (NSETS > 1 ? __popc(act[1 % NSETS]) : 0); o[1] = same; o[2] = 0; o[3] = -1; }
#endif
    if (same || it < MCG_COOP_FULL_STEPS) {           // uniform.  A consistent candidate is the minimiser; the first iterations step to x anyway
      al = xl;
      if (same) { _Pragma("unroll") for (int s = 0; s < NSETS; s++) fin[s] = act[s]; conv = true; break; }
      continue;
    }
    // (f) exact line search from a along p = x - a: phi'(alpha) = s0 + alpha quad + sum_rows D min(0, r0 + alpha dr) dr, piecewise linear and
    // increasing; s0 = (H0 a - g0) . p and quad = p^T H0 p are the smooth part's
    COOP_COUNT(2, 1);
    const real pl = xl - al;
    real h0p = 0;
    static_for<COOP_NV>([&](auto Jj) { constexpr int j = Jj; h0p = fma(H0[j], coop_rdlane(pl, j), h0p); });
    const bool own = L < COOP_NV;
    const real s0 = coop_sum32(sel(own, (h0a - g0) * pl, 0.0)), quad = coop_sum32(sel(own, h0p * pl, 0.0));
    real dr_[NSETS];
    _Pragma("unroll") for (int s = 0; s < NSETS; s++) dr_[s] = rx[s] - r0[s];
    auto dphi = [&](real alp, real& slope) {
      COOP_COUNT(4, 1);
      real f = 0, sl = 0;
      _Pragma("unroll") for (int s = 0; s < NSETS; s++) {
        const real rr = fma(alp, dr_[s], r0[s]);
        const bool on = Dr[s] > 0 && rr < 0;
        const real dd = Dr[s] * dr_[s];
        f += sel(on, dd * rr, 0.0); sl += sel(on, dd * dr_[s], 0.0);
      }
      slope = quad + coop_sum32(sl);
      return fma(alp, quad, s0) + coop_sum32(f);
    };
    real lo = 0, hi = 2, sl;
    const bool beyond = dphi(hi, sl) < 0;
    real alp = 1.0;
#ifdef MCG_COOP_DEBUG
    if (dbt && it == 1) { double* o = g_coop_dbg + e * 512 + 464 + 8 * 3; o[0] = 2; o[1] = beyond; o[2] = sl; }
#endif
    for (int b = 0; b < 8 && !beyond; b++) {                                   // (uniform)
      const real f = dphi(alp, sl);
#ifdef MCG_COOP_DEBUG
      if (dbt && it == 1) { double* o = g_coop_dbg + e * 512 + 464 + b * 3; o[0] = alp; o[1] = f; o[2] = sl; }
#endif
      const bool neg = f < 0;
      lo = sel(neg, alp, lo); hi = sel(neg, hi, alp);
      const real nwt = alp - f / sl;
      const real nx = sel(nwt >= lo && nwt <= hi, nwt, 0.5 * (lo + hi));      // closed: AT the root the step is zero and nwt == alp == lo or hi
      const bool moved = fabs(nx - alp) > 1e-10 * fmax(1.0, fabs(alp));      // on the root's own linear piece Newton stays put; the step
      alp = nx;                                                               // length needs no more: the LAST iteration is a full step
      if (!moved) break;
    }
    const real alpha = sel(beyond, 2.0, alp);
#ifdef MCG_COOP_DEBUG
    if (dbt) { double* o = g_coop_dbg + e * 512 + 400 + it * 8; o[2] = 1; o[3] = alpha; o[4] = s0; o[5] = quad; }
#endif
    al = fma(alpha, pl, al);
    MCG_TICK_PIN(&al, 1);
    COOP_TICK(ST_CO_LS);
  }
#ifdef MCG_COOP_DEBUG
  if (blockIdx.x == 0 && g_coop_dbg_done[e] == 0 && threadIdx.x % 64 < 32) {
    if (L < COOP_NV) g_coop_dbg[e * 512 + 380 + L] = al;
    __builtin_amdgcn_s_waitcnt(0);
    if (L == 0) g_coop_dbg_done[e] = 1;
  }
#endif
  // ---- hand the accelerations back: robot part where the warm start was, cube part in the cube's warm-start slots
  if (L < NB) ME.st(PUB_WARM + L, al);
  else if (L < COOP_NV) ME.st(XCH_CB + 13 + (L - NB), al);
  if (L == 0) { ME.st(XCH_ACT0, coop_pack((conv && NSETS <= 3) ? sig : 0u, fin[0])); ME.st(XCH_ACT1, coop_pack(fin[1], fin[2])); }      // (four sets do not fit the two slots: no set is carried)
  COOP_TICK(ST_CO_OUT);
}

// ---- TWO environments per wave (round 3, second half).  The 32-lane solve above leaves half of the wave's lanes unused and, in its
// matrix phases, 18 of 64.  Here the cube wave and the M / RNE waves keep their upper 32 lanes alive for the cooperative phase (they are
// masked off everywhere else) and every wave solves two flagged environments in lock-step: env A in lanes 0-31, env B in lanes 32-63,
// the same instruction stream, per-half state.  What had to change for that:
//   * nothing may be wave-uniform per environment: no v_readlane.  The factorisation runs in ONE 16-lane DPP row per environment on the
//     leading 16 x 16 block, every broadcast a v_mov_b64_dpp row_newbcast; rows 16 and 17 (the cube's last two angular dofs) are carried
//     TRANSPOSED -- lane j holds H[16][j] and H[17][j] -- and their 2 x 2 corner replicated in every lane, so their updates need only the
//     broadcasts the block already makes.  (Both DPP rows of a half hold the same numbers: nothing to mask.)
//   * vectors every row needs (the iterate, the candidate) go through 18 doubles of LDS instead of 18 v_readlane pairs;
//   * 32-lane sums: four DPP steps and one v_permlane16_swap (gfx950) instead of two v_readlane.
// And the assembly H = H0 + J~^T J~, g = g0 + J~^T aref~ (rows pre-scaled by sqrt(D): J~ = sqrt(D) J) is a matrix product: the active
// rows of a window go to the MATRIX CORES, four rows per v_mfma_f64_16x16x4_f64 -- A = B = the window's columns 0..15 for the leading
// block, columns 16..18 (dofs 16, 17 and aref~) for the border and the corner: three instructions per four rows and two LDS reads per
// lane, against ~80 VALU / DPP issue slots per row before (tools/microbench/mfma_f64.hip: operand and result layout, 80-90 clocks per
// instruction).
typedef double coop_vd4 __attribute__((ext_vector_type(4)));
typedef double coop_vd2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) coop_vd2* LdsPtr2;
constexpr int CP_WS = COOP_WS_DOUBLES / 2;                    // per environment
constexpr int CP_N = 16;                                       // the leading block
constexpr int CP_MSTRIDE = 18, CP_VOFF = CP_N * CP_MSTRIDE, CP_COFF = CP_VOFF + 3 * CP_N, CP_AOFF = CP_COFF + 8;
static_assert(CP_AOFF + COOP_NV <= CP_WS && (COOP_WIN - 1) * COOP_WSTRIDE + 32 <= CP_WS, "pair workspace");
static_assert(CP_VOFF % 2 == 0 && CP_COFF % 2 == 0 && CP_AOFF % 2 == 0 && COOP_WSTRIDE % 2 == 0 && CP_MSTRIDE % 2 == 0 && CP_WS % 2 == 0, "16-byte LDS accesses");

MCG_DEV real coop_sum32h(real v) {     // sum over the 32 lanes of each half of the wave, in every lane of the half
  v = coop_sum16(v);
  // the partner row's sum through the LDS crossbar (ds_bpermute, no memory touched)
  const int peer = (int)(((threadIdx.x & 63) ^ 16) << 2);
  const int lo = __builtin_amdgcn_ds_bpermute(peer, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(peer, __double2hiint(v));
  return v + __hiloint2double(hi, lo);
}
MCG_DEV coop_vd2 coop_ld2(LdsPtr p) { return *(LdsPtr2)p; }
MCG_DEV void coop_st2(LdsPtr p, real a, real b) { coop_vd2 v = {a, b}; *(LdsPtr2)p = v; }

// eA, eB: the two environments' lanes (LDS columns); haveB false: one environment left, the upper half idles on a copy of A's data
template <int NSETS>
MCG_DEV void coop_solve_pair(ModelPtr Pm, LdsPtr lds0, LdsPtr wsw, int eA, int eB, int nconA, int nconB, bool haveB, CoopClocks& CK) {
  constexpr int NV = COOP_NV, N = CP_N;
  const int T = threadIdx.x & 63, hl = T & 31, l16 = T & 15, drow = T >> 4;
  const bool up = T >= 32;
  const int e = sel(up, eB, eA), ncon = sel(up, nconB, nconA);
  const bool have = !up || haveB;
  const int nconmax = nconA > nconB ? nconA : nconB;             // uniform
  const PnpScratch ME(lds0 + e);
  const LdsPtr ws = wsw + sel(up, CP_WS, 0);                     // this half's workspace; env A's is wsw, env B's wsw + CP_WS
  COOP_COUNT(0, haveB ? 2 : 1);
#ifdef MCG_STAGE_CLOCKS
  CK.last = __builtin_readcyclecounter();
#endif
  // ---- the environment's cube and pair numbers
  Cube Cb; real drs[2];
  for (int k = 0; k < 3; k++) Cb.pos[k] = ME.ld(XCH_CB + k);
  for (int k = 0; k < 4; k++) Cb.quat[k] = ME.ld(XCH_CB + 3 + k);
  for (int k = 0; k < 6; k++) { Cb.vel[k] = ME.ld(XCH_CB + 7 + k); Cb.warm[k] = ME.ld(XCH_CB + 13 + k); }
  drs[0] = ME.ld(XCH_DR); drs[1] = ME.ld(XCH_DR + 1);
  CubeSys<PnpScratch> CS(ME, Cb, drs);
  CS.pm_bits = (unsigned long long)Pm;
  CS.derive(Pm);
  bool grip = false;
  unsigned sig = (unsigned)ncon;
  for (int c = 0; c < nconmax; c++) {
    const int type = (int)ME.ld(LDS_CON + sel(c < ncon, c, 0) * CON_STRIDE + CON_TYPE);
    const bool in = c < ncon;
    grip = grip || (in && pair_robot_body(type) >= 6);
    sig = sel(in, coop_sig_step(sig, type), sig);
  }
  sig |= 0x80000000u;
  unsigned guess[2]; bool have_guess;                            // (per half)
  { const real p0 = ME.ld(XCH_ACT0), p1 = ME.ld(XCH_ACT1);
    have_guess = MCG_COOP_GUESS && (unsigned)__double2hiint(p0) == sig;
    guess[0] = (unsigned)__double2loint(p0); guess[1] = (unsigned)__double2hiint(p1); }
  unsigned fin[2] = {0u, 0u}; bool conv = false;
  static_assert(NSETS <= 2, "the pair solve carries two sets of rows");
  COOP_COUNT(7, __popcll(__ballot(hl == 0 && have && have_guess)));
  real tc[NB][6];
  coop_twists(ME, Cb.pos, grip, tc);
  real qd[NB], vc[6];
  _Pragma("unroll") for (int j = 0; j < NB; j++) qd[j] = ME.ld(XCH_QD + j);
  _Pragma("unroll") for (int k = 0; k < 6; k++) vc[k] = Cb.vel[k];
  COOP_TICK(ST_CO_SETUP);
  // ---- rows, as in coop_solve, then scaled by sqrt(D): the cost is 1/2 sum min(0, J~ a - aref~)^2 and D never appears again
  const int nsets = (10 + 6 * nconmax + PNP_LANES - 1) / PNP_LANES;          // uniform
  real J[NSETS][NV], aref[NSETS]; bool lv[NSETS];
#ifdef MCG_STAGE_CLOCKS
  int rowcls[NSETS];                                                // 0 limit, 1 static geom - robot, 2 a contact of the cube
  _Pragma("unroll") for (int s = 0; s < NSETS; s++) rowcls[s] = 0;
#endif
  _Pragma("unroll") for (int s = 0; s < NSETS; s++) {
    _Pragma("unroll") for (int j = 0; j < NV; j++) J[s][j] = 0;
    aref[s] = 0; lv[s] = false;
    if (s < nsets) {
      real D, ar; int cls;
      coop_build_row(ME, CS, Cb.pos, tc, qd, vc, PNP_LANES * s + hl, ncon, J[s], D, ar, cls);
#ifdef MCG_STAGE_CLOCKS
      rowcls[s] = cls;
#endif
      const real sd = sel(have && D > 0, sqrt(D), 0.0);            // (an idle half has no rows)
      _Pragma("unroll") for (int j = 0; j < NV; j++) J[s][j] *= sd;
      aref[s] = sd * ar; lv[s] = sd > 0;
    }
  }
  MCG_TICK_PIN(aref, NSETS);
  COOP_TICK(ST_CO_ROWS);
  // ---- H0 and g0: lane j of each DPP row holds row j < 16 of the leading block; rows 16, 17 of H0 are the cube inertia's last two
  // diagonal entries (uniform)
  const int i = l16;
  real H0[N], g0;
  {
    unsigned mE = 0, mM = 0;
    static_for<NB>([&](auto I) { constexpr int k = I; mE = sel(i == k, coop_row_mask(PAT_E, k), mE); mM = sel(i == k, coop_row_mask(PAT_M, k), mM); });
    const bool rob = i < NB;
    const int ir = sel(rob, i, 0);
    real he[NB], hm[NB];
    static_for<NB>([&](auto Jj) { constexpr int j = Jj;
      const int slot = sel(ir >= j, ir * (ir + 1) / 2 + j, j * (j + 1) / 2 + ir);
      he[j] = ME.ld(LDS_HEQ + slot); hm[j] = ME.ld(LDS_M + slot); });
    MCG_FENCE();
    static_for<NB>([&](auto Jj) { constexpr int j = Jj;
      const bool low = ir >= j;
      const bool nzE = sel(low, ((mE >> j) & 1u) != 0u, ((coop_row_mask(PAT_E, j) >> ir) & 1u) != 0u);
      const bool nzM = sel(low, ((mM >> j) & 1u) != 0u, ((coop_row_mask(PAT_M, j) >> ir) & 1u) != 0u);
      H0[j] = sel(rob, sel(nzE, he[j], 0.0) + sel(nzM, hm[j], 0.0), 0.0); });
    static_for<N - NB>([&](auto Kk) { constexpr int k = Kk; H0[NB + k] = sel(i == NB + k, CS.Md[k], 0.0); });
    const real gr = ME.ld(PUB_G0 + ir);
    real gcv = 0;
    static_for<N - NB>([&](auto Kk) { constexpr int k = Kk; gcv = sel(i == NB + k, CS.fs[k], gcv); });
    g0 = sel(rob, gr, gcv);
  }
  const real md6 = CS.Md[4], md7 = CS.Md[5], g06 = CS.fs[4], g07 = CS.fs[5];
  MCG_TICK_PIN(H0, N);
  COOP_TICK(ST_CO_H0);
  // ---- the iterate: a_i in lane i of the DPP rows (i < 16), a_16 and a_17 in every lane
  real am = ME.ld(sel(i < NB, PUB_WARM + i, XCH_CB + 13 + (i - NB)));
  real a6 = ME.ld(XCH_CB + 13 + 4), a7 = ME.ld(XCH_CB + 13 + 5);
  bool done = !have;
  const LdsPtr wsB = wsw + CP_WS;
  // The increments J~^T [J~ aref~] stay in the matrix cores' accumulators ACROSS the iterations: an iteration passes only the rows whose
  // membership changed through the window -- entering rows with weight +1, leaving rows with -1 (column 19 of the window row; the A
  // operand carries it) -- instead of the whole active set again.  The second iteration of a solve (the usual count under a random
  // policy is two: the carried set, then the corrected one) moves a handful of rows, not the forty its set holds; the resting cube's
  // 24 rows are assembled once.  The first iteration is what it was, bit for bit; later ones differ from a fresh assembly by the
  // rounding of a sum taken in another order.
  coop_vd4 cA0 = {0, 0, 0, 0}, cA1 = cA0, cA2 = cA0, cB0 = cA0, cB1 = cA0, cB2 = cA0;
  bool was[NSETS];
  _Pragma("unroll") for (int s = 0; s < NSETS; s++) was[s] = false;

  for (int it = 0; it < 50; it++) {
    COOP_COUNT(1, 1);
    // (a) the iterate to every row: 18 doubles of LDS
    coop_lds_sync();
    if ((drow & 1) == 0) ws[CP_AOFF + l16] = am;
    if (hl == 0) coop_st2(ws + CP_AOFF + N, a6, a7);
    coop_lds_sync();
    real av[NV];
    _Pragma("unroll") for (int j = 0; j < NV; j += 2) { const coop_vd2 v = coop_ld2(ws + CP_AOFF + j); av[j] = v[0]; av[j + 1] = v[1]; }
    MCG_FENCE();
    real r0[NSETS], h0a = 0; unsigned actA[NSETS], actB[NSETS];
    _Pragma("unroll") for (int s = 0; s < NSETS; s++) r0[s] = -aref[s];
    _Pragma("unroll") for (int j = 0; j < NV; j++) { _Pragma("unroll") for (int s = 0; s < NSETS; s++) r0[s] = fma(J[s][j], av[j], r0[s]); }
    _Pragma("unroll") for (int j = 0; j < N; j++) h0a = fma(H0[j], av[j], h0a);
    const bool use_guess = it == 0 && have_guess;                      // (per half)
    unsigned dltA[NSETS], dltB[NSETS]; real wgt[NSETS];
    _Pragma("unroll") for (int s = 0; s < NSETS; s++) {
      const bool on = sel(use_guess, lv[s] && ((guess[s] >> hl) & 1u) != 0u, r0[s] < 0);
      const unsigned long long b = __ballot(on && !done);              // (a finished half assembles nothing)
      actA[s] = (unsigned)b; actB[s] = (unsigned)(b >> 32);
      const bool chg = !done && on != was[s];                          // (... and changes nothing)
      const unsigned long long d = __ballot(chg);
      dltA[s] = (unsigned)d; dltB[s] = (unsigned)(d >> 32);
      wgt[s] = sel(on, 1.0, -1.0);
      was[s] = sel(done, was[s], on);
    }
    MCG_TICK_PIN(r0, NSETS);
    COOP_TICK(ST_CO_RESID);
    COOP_COUNT(3, __popc(dltA[0]) + __popc(dltB[0]) + (NSETS > 1 ? __popc(dltA[1 % NSETS]) + __popc(dltB[1 % NSETS]) : 0) + (NSETS > 2 ? __popc(dltA[2 % NSETS]) + __popc(dltB[2 % NSETS]) : 0));
    // (b) the increments J~^T [J~ aref~] of both environments on the matrix cores, the rows that enter or leave the active set passing
    // through a 16-row window each
    {
      // positions in the environment's list of changed rows, the sets one after the other
      int nA = 0, nB = 0, pos[NSETS]; bool mine[NSETS];
      _Pragma("unroll") for (int s = 0; s < NSETS; s++) {
        const unsigned own = sel(up, dltB[s], dltA[s]);
        pos[s] = sel(up, nB, nA) + __popc(own & ((1u << hl) - 1u));
        mine[s] = ((own >> hl) & 1u) != 0u;
        nA += __popc(dltA[s]); nB += __popc(dltB[s]);
      }
      const int nmx = nA > nB ? nA : nB, n_own = sel(up, nB, nA);
      for (int w0 = 0; w0 < nmx; w0 += COOP_WIN) {                  // uniform
        coop_lds_sync();                                            // the window's previous readers are done
        _Pragma("unroll") for (int s = 0; s < NSETS; s++) {
          if (mine[s] && pos[s] >= w0 && pos[s] < w0 + COOP_WIN) {
            const LdsPtr o = ws + (pos[s] - w0) * COOP_WSTRIDE;
            _Pragma("unroll") for (int j = 0; j < NV; j += 2) coop_st2(o + j, J[s][j], J[s][j + 1]);
            coop_st2(o + NV, aref[s], wgt[s]);
          }
        }
        {   // zero rows complete the last group of four
          const int left = n_own - w0;
          const int nw = sel(left < 0, 0, sel(left > COOP_WIN, COOP_WIN, left));
          const int npad = (-nw) & 3;
          if (hl < COOP_WSTRIDE) { _Pragma("unroll") for (int t = 0; t < 3; t++) if (t < npad) ws[(nw + t) * COOP_WSTRIDE + hl] = 0.0; }
        }
        coop_lds_sync();
        const int lA = nA - w0, lB = nB - w0;
        const int gA = (sel(lA < 0, 0, sel(lA > COOP_WIN, COOP_WIN, lA)) + 3) >> 2, gB = (sel(lB < 0, 0, sel(lB > COOP_WIN, COOP_WIN, lB)) + 3) >> 2;
        // four window rows per instruction: lane (k = drow, c = l16) holds W[4 g + k][c].  All operands of an environment's window first (one
        // LDS round trip, not one per group), then its matrix instructions back to back
        auto window = [&](const LdsPtr wse, int ng, coop_vd4& c0, coop_vd4& c1, coop_vd4& c2) {
          real lo[4], hi[4], wg[4];
          _Pragma("unroll") for (int g = 0; g < 4; g++) { const int o = (4 * g + drow) * COOP_WSTRIDE; lo[g] = wse[o + l16]; hi[g] = wse[o + l16 + N]; wg[g] = wse[o + NV + 1]; }
          MCG_FENCE();
          _Pragma("unroll") for (int g = 0; g < 4; g++) {
            if (g < ng) {                                           // uniform
              const real h3 = sel(l16 < 3, hi[g], 0.0);
              const real la = lo[g] * wg[g], ha = h3 * wg[g];         // the row's weight (+1 entering, -1 leaving, 0 padding) on the A operand
              c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(la, lo[g], c0, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(la, h3, c1, 0, 0, 0);
              c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ha, h3, c2, 0, 0, 0);
            }
          }
        };
        window(wsw, gA, cA0, cA1, cA2);
        window(wsB, gB, cB0, cB1, cB2);
      }
    }
    // result element v of lane (drow, l16) is entry (drow + 4 v, l16): scattered as a 16 x 16 matrix, three 16-vectors (columns 16, 17 and
    // the gradient) and the 2 x 3 corner; then lane j of each DPP row gathers row j (= column j) and its entries of rows 16, 17
    coop_lds_sync();
    _Pragma("unroll") for (int v = 0; v < 4; v++) { wsw[(drow + 4 * v) * CP_MSTRIDE + l16] = cA0[v]; wsB[(drow + 4 * v) * CP_MSTRIDE + l16] = cB0[v]; }
    if (l16 < 3) { _Pragma("unroll") for (int v = 0; v < 4; v++) { wsw[CP_VOFF + l16 * N + drow + 4 * v] = cA1[v]; wsB[CP_VOFF + l16 * N + drow + 4 * v] = cB1[v]; } }
    if (drow < 2 && l16 < 3) { wsw[CP_COFF + drow * 3 + l16] = cA2[0]; wsB[CP_COFF + drow * 3 + l16] = cB2[0]; }
    coop_lds_sync();
    real Hc[N], e6, e7, g, c66, c76, c77, g6, g7;
    _Pragma("unroll") for (int k = 0; k < N; k += 2) { const coop_vd2 v = coop_ld2(ws + l16 * CP_MSTRIDE + k); Hc[k] = v[0]; Hc[k + 1] = v[1]; }
    e6 = ws[CP_VOFF + l16]; e7 = ws[CP_VOFF + N + l16]; g = ws[CP_VOFF + 2 * N + l16];
    { const coop_vd2 u0 = coop_ld2(ws + CP_COFF), u1 = coop_ld2(ws + CP_COFF + 2), u2 = coop_ld2(ws + CP_COFF + 4);
      c66 = u0[0]; c76 = u0[1]; g6 = u1[0]; c77 = u2[0]; g7 = u2[1]; }
    MCG_FENCE();
    _Pragma("unroll") for (int k = 0; k < N; k++) Hc[k] += H0[k];
    g += g0; c66 += md6; c77 += md7; g6 += g06; g7 += g07;
#ifdef MCG_COOP_DEBUG
    const bool dbg = blockIdx.x == 0 && it == 0 && have && g_coop_dbg_done[e] == 0 && (drow & 1) == 0;
    if (dbg) { double* o = g_coop_dbg + e * 512;
      for (int k = 0; k < N; k++) o[l16 * NV + k] = Hc[k];
      o[l16 * NV + 16] = e6; o[l16 * NV + 17] = e7; o[16 * NV + l16] = e6; o[17 * NV + l16] = e7;
      o[324 + l16] = g; o[360 + l16] = am;
      if (l16 == 0) { o[16 * NV + 16] = c66; o[16 * NV + 17] = c76; o[17 * NV + 16] = c76; o[17 * NV + 17] = c77; o[324 + 16] = g6; o[324 + 17] = g7; o[360 + 16] = a6; o[360 + 17] = a7; } }
#endif
    MCG_TICK_PIN(Hc, N);
    COOP_TICK(ST_CO_ASM);
    // (c) H = L D L^T in the 16 lanes of a DPP row.  Lane j keeps the unscaled H[j][k] = L[j][k] D_k in Hc[k] (k < j), the scaled
    // L[j][k] in Lr[k] (zero for k >= j); e6 / e7 end as H[16][j], H[17][j] (unscaled), the corner as D_16, L[17][16] D_16, D_17 + ...
    // (the scaled L[j][k] = Hc[k] / D_k is formed again where the forward substitution needs it: sixteen more multiplications, sixteen
    // fewer numbers across the register peak)
    real dinv[N], dinv_own = 0;
    static_for<N>([&](auto Kk) { constexpr int k = Kk;
      const real dk = coop_bcast16<k>(Hc[k]);
      dinv[k] = rcp_nr(dk);
      const real lk = Hc[k] * dinv[k];
      const real b6 = coop_bcast16<k>(e6), b7 = coop_bcast16<k>(e7);
      static_for<N - 1 - k>([&](auto Jj) { constexpr int j = k + 1 + Jj;
        const real sj = coop_bcast16<j>(Hc[k]);
        Hc[j] = fma(-lk, sj, Hc[j]); });
      const real lkm = sel(l16 > k, lk, 0.0);
      e6 = fma(-lkm, b6, e6); e7 = fma(-lkm, b7, e7);
      const real l6 = b6 * dinv[k], l7 = b7 * dinv[k];
      c66 = fma(-l6, b6, c66); c76 = fma(-l7, b6, c76); c77 = fma(-l7, b7, c77);
      dinv_own = sel(l16 == k, dinv[k], dinv_own); });
    const real dinv6 = rcp_nr(c66), l76 = c76 * dinv6;
    const real dinv7 = rcp_nr(fma(-l76, c76, c77));
    MCG_TICK_PIN(Hc, N);
    COOP_TICK(ST_CO_FACTOR);
    // (d) x = H^-1 g: forward in row layout, rows 16 / 17 by two 16-lane sums; backward on the unscaled columns through one LDS transpose
    real acc = g;
    static_for<N>([&](auto Kk) { constexpr int k = Kk;
      const real yk = coop_bcast16<k>(acc);
      acc = fma(-sel(l16 > k, Hc[k] * dinv[k], 0.0), yk, acc); });
    const real zt = acc * dinv_own;
    const real y6 = g6 - coop_sum16(e6 * zt);
    const real y7 = g7 - coop_sum16(e7 * zt) - l76 * y6;
    const real x7 = y7 * dinv7, x6 = fma(-l76, x7, y6 * dinv6);
    acc = fma(-e6, x6, fma(-e7, x7, acc));
    coop_lds_sync();
    if ((drow & 1) == 0) { _Pragma("unroll") for (int k = 0; k < N; k += 2) coop_st2(ws + l16 * CP_MSTRIDE + k, sel(l16 > k, Hc[k], 0.0), sel(l16 > k + 1, Hc[k + 1], 0.0)); }
    coop_lds_sync();
    real U[N];                                       // U[j] = H[j][i] (unscaled) for j > i, zero otherwise
    _Pragma("unroll") for (int j = 0; j < N; j++) U[j] = ws[j * CP_MSTRIDE + l16];
    MCG_FENCE();
    real xm = 0;
    static_for<N>([&](auto Kk) { constexpr int j = N - 1 - Kk;
      const real xj = coop_bcast16<j>(acc) * dinv[j];
      acc = fma(-U[j], xj, acc);
      xm = sel(l16 == j, xj, xm); });
#ifdef MCG_COOP_DEBUG
    if (dbg) { double* o = g_coop_dbg + e * 512; o[342 + l16] = xm; if (l16 == 0) { o[342 + 16] = x6; o[342 + 17] = x7; } }
#endif
    // the candidate to every row
    coop_lds_sync();
    if ((drow & 1) == 0) ws[CP_AOFF + l16] = xm;
    if (hl == 0) coop_st2(ws + CP_AOFF + N, x6, x7);
    coop_lds_sync();
    real rx[NSETS];
    _Pragma("unroll") for (int s = 0; s < NSETS; s++) rx[s] = -aref[s];
    _Pragma("unroll") for (int j = 0; j < NV; j += 2) {
      const coop_vd2 v = coop_ld2(ws + CP_AOFF + j);
      _Pragma("unroll") for (int s = 0; s < NSETS; s++) rx[s] = fma(J[s][j + 1], v[1], fma(J[s][j], v[0], rx[s]));
    }
    MCG_TICK_PIN(rx, NSETS);
    COOP_TICK(ST_CO_SOLVE);
    // (e) does each candidate keep its active set?
    bool sameA = true, sameB = true;
    _Pragma("unroll") for (int s = 0; s < NSETS; s++) {
      const unsigned long long b = __ballot(rx[s] < 0 && !done);
      sameA = sameA && (unsigned)b == actA[s]; sameB = sameB && (unsigned)(b >> 32) == actB[s];
    }
    const bool same = sel(up, sameB, sameA);
    COOP_TICK(ST_CO_CHECK);
    COOP_COUNT(5, __popcll(__ballot(hl == 0 && have && use_guess && same)));
    const bool full = same || it < MCG_COOP_FULL_STEPS;
    const bool need = !done && !full;                // this half wants a line search
#ifdef MCG_COOP_DEBUG
    const bool dbt = blockIdx.x == 0 && have && !done && g_coop_dbg_done[e] == 0 && it < 8 && hl == 0;
    if (dbt) { double* o = g_coop_dbg + e * 512 + 400 + it * 8; const unsigned a0 = sel(up, actB[0], actA[0]), a1 = sel(up, actB[1 % NSETS], actA[1 % NSETS]);
      o[0] = __popc(a0) + (NSETS > 1 ? __popc(a1) : 0); o[1] = same; o[2] = 0; o[3] = -1; }
#endif
    if (!done && full) { am = xm; a6 = x6; a7 = x7; }
    { const bool fresh = !done && same;
      _Pragma("unroll") for (int s = 0; s < NSETS; s++) fin[s] = sel(fresh, sel(up, actB[s], actA[s]), fin[s]);
      conv = conv || fresh; }
    done = done || same;
    if (__any(need)) {
      // (f) exact line search (see coop_solve), each half its own; a half that needs none walks along with frozen numbers
      COOP_COUNT(2, 1);
      const real pm = xm - am, p6 = x6 - a6, p7 = x7 - a7;
      real h0p = 0;
      static_for<N>([&](auto Jj) { constexpr int j = Jj; h0p = fma(H0[j], coop_bcast16<j>(pm), h0p); });
      const real s0 = coop_sum16((h0a - g0) * pm) + (md6 * a6 - g06) * p6 + (md7 * a7 - g07) * p7;
      const real quad = coop_sum16(h0p * pm) + md6 * p6 * p6 + md7 * p7 * p7;
      real dr_[NSETS];
      _Pragma("unroll") for (int s = 0; s < NSETS; s++) dr_[s] = rx[s] - r0[s];
      auto dphi = [&](real alp, real& slope) {
        COOP_COUNT(4, 1);
        real f = 0, sl = 0;
        _Pragma("unroll") for (int s = 0; s < NSETS; s++) {
          const real rr = fma(alp, dr_[s], r0[s]);
          const bool on = rr < 0;
          f += sel(on, dr_[s] * rr, 0.0); sl += sel(on, dr_[s] * dr_[s], 0.0);
        }
        slope = quad + coop_sum32h(sl);
        return fma(alp, quad, s0) + coop_sum32h(f);
      };
      real lo = 0, hi = 2, sl;
      const bool beyond = dphi(hi, sl) < 0;
      real alp = 1.0;
      bool fin = !need || beyond;
#ifdef MCG_COOP_DEBUG
      if (dbt && need && it == 1) { double* o = g_coop_dbg + e * 512 + 464 + 8 * 3; o[0] = 2; o[1] = beyond; o[2] = sl; }
#endif
      for (int b = 0; b < 8; b++) {
        if (!__any(!fin)) break;                       // uniform
        const real f = dphi(alp, sl);
#ifdef MCG_COOP_DEBUG
        if (dbt && need && it == 1 && !fin) { double* o = g_coop_dbg + e * 512 + 464 + b * 3; o[0] = alp; o[1] = f; o[2] = sl; }
#endif
        const bool neg = f < 0;
        const real lo2 = sel(neg, alp, lo), hi2 = sel(neg, hi, alp);
        const real nwt = alp - f / sl;
        const real nx = sel(nwt >= lo2 && nwt <= hi2, nwt, 0.5 * (lo2 + hi2));
        const bool moved = fabs(nx - alp) > 1e-10 * fmax(1.0, fabs(alp));
        lo = sel(fin, lo, lo2); hi = sel(fin, hi, hi2); alp = sel(fin, alp, nx);
        fin = fin || !moved;
      }
      const real alpha = sel(beyond, 2.0, alp);
#ifdef MCG_COOP_DEBUG
      if (dbt && need) { double* o = g_coop_dbg + e * 512 + 400 + it * 8; o[2] = 1; o[3] = alpha; o[4] = s0; o[5] = quad; }
#endif
      if (need) { am = fma(alpha, pm, am); a6 = fma(alpha, p6, a6); a7 = fma(alpha, p7, a7); }
      MCG_TICK_PIN(&am, 1);
      COOP_TICK(ST_CO_LS);
    }
    if (!__any(!done)) break;                          // uniform: both halves have their minimiser
  }
#ifdef MCG_COOP_DEBUG
  if (blockIdx.x == 0 && have && g_coop_dbg_done[e] == 0 && (drow & 1) == 0) {
    double* o = g_coop_dbg + e * 512; o[380 + l16] = am; if (l16 == 0) { o[380 + 16] = a6; o[380 + 17] = a7; }
    __builtin_amdgcn_s_waitcnt(0);
    if (l16 == 0) g_coop_dbg_done[e] = 1;
  }
#endif
#ifdef MCG_STAGE_CLOCKS
  {   // census: would the set of TWO sub-steps ago have been right (period-2 chatter)?
    const real q0 = ME.ld(LDS_POLY + 62), q1 = ME.ld(LDS_POLY + 63);
    const bool sig2 = (unsigned)__double2hiint(q0) == sig;
    const unsigned g20 = (unsigned)__double2loint(q0), g21 = (unsigned)__double2hiint(q1);
    const bool eq2 = have && conv && sig2 && g20 == fin[0] && g21 == fin[1 % NSETS];
    const bool eq1 = have && conv && have_guess && guess[0] == fin[0] && guess[1 % NSETS] == fin[1 % NSETS];
    const unsigned long long b2 = __ballot(eq2 && hl == 0), b1 = __ballot(eq1 && hl == 0), b12 = __ballot(eq2 && !eq1 && hl == 0);
    if (T == 0) { atomicAdd(&g_stage_clocks[ST_COUNT + CN_G_EQ1], (unsigned long long)__popcll(b1)); atomicAdd(&g_stage_clocks[ST_COUNT + CN_G_EQ2], (unsigned long long)__popcll(b2));
                  atomicAdd(&g_stage_clocks[ST_COUNT + CN_G_EQ2N1], (unsigned long long)__popcll(b12)); }
    if (have && hl == 0) { ME.st(LDS_POLY + 62, ME.ld(XCH_ACT0)); ME.st(LDS_POLY + 63, ME.ld(XCH_ACT1)); }      // (before the hand-back overwrites them)
  }
  {   // where a carried active set was wrong: rows by class, missing from / surplus in the guess
    _Pragma("unroll") for (int s = 0; s < NSETS; s++) {
      const bool g = ((guess[s] >> hl) & 1u) != 0u && lv[s], f = ((fin[s] >> hl) & 1u) != 0u;
      const bool cnt = have && have_guess && conv;
      const unsigned long long wrong = __ballot(cnt && g != f);
      const unsigned long long w0 = __ballot(cnt && g != f && rowcls[s] == 0), w1 = __ballot(cnt && g != f && rowcls[s] == 1), w2 = __ballot(cnt && g != f && rowcls[s] == 2);
      const unsigned long long miss = __ballot(cnt && !g && f), extra = __ballot(cnt && g && !f);
      if (T == 0 && wrong) {
        atomicAdd(&g_stage_clocks[ST_COUNT + CN_G_LIM], (unsigned long long)__popcll(w0)); atomicAdd(&g_stage_clocks[ST_COUNT + CN_G_STAT], (unsigned long long)__popcll(w1));
        atomicAdd(&g_stage_clocks[ST_COUNT + CN_G_CUBE], (unsigned long long)__popcll(w2));
        atomicAdd(&g_stage_clocks[ST_COUNT + CN_G_MISSING], (unsigned long long)__popcll(miss)); atomicAdd(&g_stage_clocks[ST_COUNT + CN_G_EXTRA], (unsigned long long)__popcll(extra));
      }
    }
    bool anyw = false;
    _Pragma("unroll") for (int s = 0; s < NSETS; s++) anyw = anyw || (have && have_guess && conv && (((guess[s] >> hl) & 1u) != 0u && lv[s]) != (((fin[s] >> hl) & 1u) != 0u));
    const unsigned long long aw = __ballot(anyw);
    if (T == 0) atomicAdd(&g_stage_clocks[ST_COUNT + CN_G_FAILED], (unsigned long long)(((unsigned)aw != 0u) + ((unsigned)(aw >> 32) != 0u)));
  }
#endif
  // ---- hand the accelerations back
  if (have && (drow & 1) == 0) {
    if (l16 < NB) ME.st(PUB_WARM + l16, am);
    else ME.st(XCH_CB + 13 + (l16 - NB), am);
    if (l16 == 0) { ME.st(XCH_CB + 13 + 4, a6); ME.st(XCH_CB + 13 + 5, a7);
                    ME.st(XCH_ACT0, coop_pack(conv ? sig : 0u, fin[0])); ME.st(XCH_ACT1, coop_pack(fin[1 % NSETS], 0u)); }
  }
  COOP_TICK(ST_CO_OUT);
}

// ---- the cooperative phase.  The cube wave and the M / RNE waves run it between barriers S4 and S5, all 64 lanes alive, with the same
// `mask` (bit l: lane l's environment is flagged).  The flagged environments are handed out from a counter in LDS (solves differ by a
// factor of ten in their Newton iteration counts: a fixed split leaves the others waiting for the unluckiest), two per turn of a wave;
// the robot wave clears the counter before S4 and sleeps at S5.
//
// Where the code lives decides the kernel's HBM traffic.  An out-of-line function saves and restores every callee-saved register it
// uses -- 218 dwords a lane for the 32-lane solve -- on EVERY call: with one call per wave per coupled sub-step that was 56 KB per
// wave-call, 133 KB of scratch traffic per env-step of the scripted grasp (98x the algorithmic bytes; rocprofv3 PMC, profiles/r03v).  So
// the common shape (two environments of up to nine contacts) is INLINED into the kernel's cube-wave and M / RNE-wave branches, which hold
// next to nothing across the phase (the cube wave its cube, the others nothing).  The robot wave -- whose live state would be spilled
// around an inlined copy, and whose contact-free pipeline an inlined copy slowed in round 2 -- called an out-of-line instance until the
// pair solve; since then it takes only a rank-fixed share (every COOP_ROBOT_EVERY-th flagged environment, and none below that many).  The rare shape (more than nine
// contacts) and the one-environment-per-wave routing (MCG_COOP_PAIR=0: the first implementation, kept as a cross-check of the second)
// are out of line.
static __device__ __noinline__ void coop_solve_single(unsigned long long model_bits, unsigned lds_base, int e_, int wave, int ncon_) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)model_bits), hi = __builtin_amdgcn_readfirstlane((unsigned)(model_bits >> 32));
  const ModelPtr P = (ModelPtr)(((unsigned long long)hi << 32) | lo);
  const LdsPtr lds0 = (LdsPtr)(uintptr_t)__builtin_amdgcn_readfirstlane(lds_base);
  const int e = __builtin_amdgcn_readfirstlane(e_), w = __builtin_amdgcn_readfirstlane(wave);
  const int ncon = __builtin_amdgcn_readfirstlane(ncon_);
  const LdsPtr ws = lds0 + COOP_WS_ROW * PNP_LANES + w * COOP_WS_DOUBLES;
  CoopClocks CK;
#ifdef MCG_STAGE_CLOCKS
  for (int k = 0; k < ST_CO_IDLE - ST_CO_SETUP; k++) CK.t[k] = 0;
  for (int k = 0; k < 8; k++) CK.n[k] = 0;
#endif
  if (10 + 6 * ncon <= 2 * PNP_LANES) coop_solve<2>(P, lds0, e, ws, ncon, CK);
  else coop_solve<COOP_SETS>(P, lds0, e, ws, ncon, CK);
  coop_flush_clocks(CK);
}

// mode 0: every wave takes one environment at a time (MCG_COOP_PAIR=0) | 1: a 64-lane wave, two at a time | 2: the robot wave beside
// 64-lane waves.  WHICH solver an environment gets must not depend on the race for the counter -- the two differ in the last bits, and a
// run must reproduce bit for bit -- so the robot wave's share is fixed by rank: of the flagged environments, in lane order, every
// seventh is the robot wave's (a 32-lane solve takes about as long as a pair's), the others are handed out in pairs.  (Who is paired
// with whom does race; a half's arithmetic never sees the other half's numbers.)
constexpr int COOP_ROBOT_EVERY = 13;     // (round 4: 7 -> 13.  With twice the flagged environments the robot wave had a share in every fifth sub-step, and each of its out-of-line calls is scratch traffic; it now joins only when the other three waves would need a third round)
template <bool PAIRS = true>      // PAIRS = false: the robot wave's instance (modes 0 and 2 only) -- without the inlined pair solve its out-of-line
                                  // frame saves a handful of registers per call instead of ~50 dwords a lane
MCG_DEV void coop_phase_body(ModelPtr P, LdsPtr lds0, unsigned m, int w, int mode) {
  const LdsPtr ws = lds0 + COOP_WS_ROW * PNP_LANES + w * COOP_WS_DOUBLES;
  CoopClocks CK;
#ifdef MCG_STAGE_CLOCKS
  for (int k = 0; k < ST_CO_IDLE - ST_CO_SETUP; k++) CK.t[k] = 0;
  for (int k = 0; k < 8; k++) CK.n[k] = 0;
#endif
  typedef __attribute__((address_space(3))) unsigned* LdsCtr;
  const LdsCtr ctr = (LdsCtr)(lds0 + COOP_CTR_SLOT * PNP_LANES);
  const int total = __popc(m);
  auto grab = [&]() {
    unsigned k = 0;
    if ((threadIdx.x & 63) == 0) k = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return (int)__builtin_amdgcn_readfirstlane(k);
  };
  auto kth = [&](int k) { unsigned mm = m; for (int q = 0; q < k; q++) mm &= mm - 1u; return __builtin_ctz(mm); };      // the k-th flagged lane
  auto ncon_of = [&](int e) { return __builtin_amdgcn_readfirstlane((int)lds0[XCH_NCON * PNP_LANES + e]); };
  if (mode == 2) {
    for (int k = COOP_ROBOT_EVERY - 1; k < total; k += COOP_ROBOT_EVERY) { const int e = kth(k); coop_solve_single((unsigned long long)P, (unsigned)(uintptr_t)lds0, e, w, ncon_of(e)); }
  } else if (mode == 0) {
    for (;;) {
      const int k = grab();
      if (k >= total) break;
      const int e = kth(k);
      coop_solve_single((unsigned long long)P, (unsigned)(uintptr_t)lds0, e, w, ncon_of(e));
    }
  } else if constexpr (PAIRS) {
    const int npair = total - total / COOP_ROBOT_EVERY;              // the ranks that are not the robot wave's
    auto rank_of = [&](int j) { return j + j / (COOP_ROBOT_EVERY - 1); };
    for (;;) {
      const int j1 = grab();
      if (j1 >= npair) break;
      const int e1 = kth(rank_of(j1)), n1 = ncon_of(e1);
      // a list of more than nine contacts (rows beyond two sets) goes through the 32-lane solve -- INLINED here (one call site): these waves
      // hold nothing across the phase, and an out-of-line call saves and restores ~220 dwords a lane (round 4: with every mesh colliding such
      // lists are common under the mocap controller, and the calls were 370 MB of HBM writes per launch, 48x the algorithmic bytes)
      int e2 = e1, n2 = n1, big = -1, nbig = 0; bool have2 = false, pair = true;
      if (10 + 6 * n1 > 2 * PNP_LANES) { big = e1; nbig = n1; pair = false; }
      else {
        const int j2 = grab();
        if (j2 < npair) {
          const int ec = kth(rank_of(j2)), nc = ncon_of(ec);
          if (10 + 6 * nc <= 2 * PNP_LANES) { e2 = ec; n2 = nc; have2 = true; } else { big = ec; nbig = nc; }
        }
      }
      if (pair) coop_solve_pair<2>(P, lds0, ws, e1, e2, n1, n2, have2, CK);
      if (big >= 0) coop_solve<COOP_SETS>(P, lds0, big, ws, nbig, CK);
    }
  }
  coop_flush_clocks(CK);
}

// the robot wave's instance: 32 lanes, one environment at a time through the out-of-line solve (its live state stays in its registers)
static __device__ __noinline__ void coop_phase(unsigned long long model_bits, unsigned lds_base, unsigned mask, int wave, int mode) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)model_bits), hi = __builtin_amdgcn_readfirstlane((unsigned)(model_bits >> 32));
  const ModelPtr P = (ModelPtr)(((unsigned long long)hi << 32) | lo);
  coop_phase_body<false>(P, (LdsPtr)(uintptr_t)__builtin_amdgcn_readfirstlane(lds_base), __builtin_amdgcn_readfirstlane(mask), __builtin_amdgcn_readfirstlane(wave),
                         __builtin_amdgcn_readfirstlane(mode));
}

}  // namespace mcg
