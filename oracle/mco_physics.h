/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU float64 restatement of the arithmetic the reference's hot path delegates to MuJoCo 2.3.2
 * (`mujoco.mj_step(model, data, nstep=frame_skip)` at /root/reference/mycobotgym/envs/mycobot.py:170,189
 * and `mj_forward` at :213,229,453) for the model class the MyCobot scene needs: a tree of
 * bodies with hinge / free joints, affine `general` actuators over joints or fixed tendons,
 * connect / joint / weld equalities, mocap bodies, joint limits, box contacts, soft constraints solved
 * by a primal Newton method, semi-implicit Euler with implicit joint damping.
 *
 * PARITY UNPINNED: MuJoCo 2.3.2 (requirements.txt:4) is a third-party dependency that is absent
 * from /root/reference and from this image, and the reference ships no tests or golden vectors
 * for this path.  Every rule below is restated from the published MuJoCo algorithm as recalled
 * (all marked [RECALL]); it is pinned only by the reference's own data
 * (tests/test_oracle_known_answers.py): the known answers of SURVEY.md Appendix E (forward kinematics,
 * mesh volumes, id table) and the settled `fetch_env` keyframe of mycobot280_mocap.xml:6-9, which is a
 * statics known answer -- the gripper's six deflections are reproduced (gear joints to 1e-6 relative)
 * and the arm hangs on the weld at the keyframe if and only if the weld's six rows are equally stiff.
 * Not reproduced: the cube's rest height in the keyframes (contacts 2x too stiff).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 */
#ifndef MCO_PHYSICS_H
#define MCO_PHYSICS_H

#ifdef __cplusplus
extern "C" {
#endif

#define MCO_MAXBODY 32
#define MCO_MAXJNT 16
#define MCO_MAXNQ 24
#define MCO_MAXNV 24
#define MCO_MAXGEOM 48
#define MCO_MAXSITE 8
#define MCO_MAXU 8
#define MCO_MAXEQ 8
#define MCO_MAXTEN 2
#define MCO_MAXTENJ 4
#define MCO_MAXEXCL 16
#define MCO_MAXCON 64      /* contacts an env can hold (a mesh geom's twin counts); the list is cut in ENTRIES, see mco_model.maxentry */
#define MCO_MAXEFC 416
#define MCO_NMESH 14       /* collision polytopes (mycobotgym_amd/model/polytope.py: MESH_NAMES) */
#define MCO_MAXMOCAP 1

enum { MCO_JNT_FREE = 0, MCO_JNT_HINGE = 3 };
enum { MCO_GEOM_PLANE = 0, MCO_GEOM_BOX = 6, MCO_GEOM_MESH = 7 };
enum { MCO_EQ_CONNECT = 0, MCO_EQ_WELD = 1, MCO_EQ_JOINT = 2 };
enum { MCO_EFC_EQUALITY = 0, MCO_EFC_LIMIT = 1, MCO_EFC_CONTACT = 2 };

typedef struct mco_model {
  int nbody, njnt, nq, nv, ngeom, nsite, nu, neq, ntendon, nexclude;
  double timestep, gravity[3], meaninertia;
  int enable_contact;            /* 0: collision stage skipped (Reach / free-space configs) */
  int collide_scope_geom;        /* >= 0: only pairs that involve this geom collide (the build's scoped set: the cube) ... */
  int collide_extra[48];         /* ... plus pairs of one geom flagged 1 (static: ground plane, table) and one flagged 2 (finger pad) */
  /* Convex-mesh collision (mycobot280_main.xml:105-247): MuJoCo collides a mesh geom's convex hull (libccd MPR, one
   * contact per pair).  Restated here for mesh <-> ground plane / static box / the scope geom (the cube) on the mesh's COLLISION
   * POLYTOPE (mycobotgym_amd/model/polytope.py: a vertex subset of the hull within 1 mm of it, the whole hull for the gripper's small
   * parts) by an EXACT separating-axis test -- the box's face axes, the polytope's face normals and the edge x edge directions that
   * are facet normals of the Minkowski difference -- one contact along the axis of least penetration (mco_collision.c).
   * geom_poly[g] = the geom's polytope (0 .. MCO_NMESH-1) or -1; poly = the table blob in geom coordinates (shared, not owned). */
  int geom_poly[48];
  const double* poly;
  int maxentry;                  /* the contact list is cut after this many ENTRIES (a primitive contact, or the identical contacts of a
                                    mesh's twin geoms: one entry); the HIP kernels hold 16 (MAXCON in csrc/mcg_cube.hpp).  MuJoCo has no cap. */
  /* Study switches (oracle/rule_study.py): alternatives to [RECALL] rules the reference's keyframes can discriminate.
   * All zero = the adopted rule set, which is what the HIP kernels implement and every parity test runs.
   *   rule[0] weld diagApprox   0 one common (translational) weight for the six rows | 1 translational rows 0-2, rotational rows 3-5
   *   rule[1] weld impedance    0 one impedance at the 6-norm of the residual | 1 one per row at |pos_row| | 2 fixed at solimp[0]
   *   rule[2] pyramid diagApprox 0 tran + mu_k^2 (tran | rot) | 1 tran
   *   rule[3] pyramid R         0 Rpy = 2 mu^2 R(first edge) on all edges | 1 each edge keeps its own R | 2 Rpy = 4 mu^2 R(first edge) */
  int rule[8];
  /* bodies */
  int body_parent[MCO_MAXBODY], body_rootid[MCO_MAXBODY], body_weldid[MCO_MAXBODY];
  int body_dofadr[MCO_MAXBODY], body_dofnum[MCO_MAXBODY];
  int body_mocapid[MCO_MAXBODY];  /* -1, or the slot in data->mocap_pos / mocap_quat (mocap model variant) */
  double body_pos[MCO_MAXBODY][3], body_quat[MCO_MAXBODY][4];
  double body_ipos[MCO_MAXBODY][3], body_iquat[MCO_MAXBODY][4];
  double body_mass[MCO_MAXBODY], body_inertia[MCO_MAXBODY][3];
  /* joints / dofs */
  int jnt_type[MCO_MAXJNT], jnt_body[MCO_MAXJNT], jnt_qposadr[MCO_MAXJNT], jnt_dofadr[MCO_MAXJNT];
  int jnt_limited[MCO_MAXJNT];
  double jnt_pos[MCO_MAXJNT][3], jnt_axis[MCO_MAXJNT][3], jnt_range[MCO_MAXJNT][2];
  double jnt_solref[MCO_MAXJNT][2], jnt_solimp[MCO_MAXJNT][5];
  int dof_body[MCO_MAXNV], dof_jnt[MCO_MAXNV], dof_parent[MCO_MAXNV];
  double dof_armature[MCO_MAXNV], dof_damping[MCO_MAXNV], qpos0[MCO_MAXNQ];
  /* geoms */
  int geom_type[MCO_MAXGEOM], geom_body[MCO_MAXGEOM], geom_condim[MCO_MAXGEOM];
  int geom_contype[MCO_MAXGEOM], geom_conaffinity[MCO_MAXGEOM];
  double geom_pos[MCO_MAXGEOM][3], geom_quat[MCO_MAXGEOM][4], geom_size[MCO_MAXGEOM][3];
  double geom_friction[MCO_MAXGEOM][3], geom_solref[MCO_MAXGEOM][2], geom_solimp[MCO_MAXGEOM][5];
  /* sites */
  int site_body[MCO_MAXSITE];
  double site_pos[MCO_MAXSITE][3], site_quat[MCO_MAXSITE][4];
  /* actuators: trntype 0 = joint, 1 = tendon */
  int act_trntype[MCO_MAXU], act_trnid[MCO_MAXU], act_ctrllimited[MCO_MAXU], act_forcelimited[MCO_MAXU];
  double act_gear[MCO_MAXU], act_gainprm[MCO_MAXU][3], act_biasprm[MCO_MAXU][3];
  double act_ctrlrange[MCO_MAXU][2], act_forcerange[MCO_MAXU][2];
  /* fixed tendons */
  int ten_num[MCO_MAXTEN], ten_jnt[MCO_MAXTEN][MCO_MAXTENJ];
  double ten_coef[MCO_MAXTEN][MCO_MAXTENJ];
  /* equalities */
  int eq_type[MCO_MAXEQ], eq_obj1[MCO_MAXEQ], eq_obj2[MCO_MAXEQ];
  double eq_data[MCO_MAXEQ][11], eq_solref[MCO_MAXEQ][2], eq_solimp[MCO_MAXEQ][5];
  int exclude[MCO_MAXEXCL][2];
  /* derived at qpos0 by mco_setconst */
  double body_invweight0[MCO_MAXBODY][2], dof_invweight0[MCO_MAXNV];
} mco_model;

typedef struct mco_contact {
  double dist, pos[3], frame[9], friction[5], solref[2], solimp[5], includemargin;
  int dim, geom1, geom2, efc_address;
} mco_contact;

typedef struct mco_data {
  /* state */
  double qpos[MCO_MAXNQ], qvel[MCO_MAXNV], ctrl[MCO_MAXU], qacc_warmstart[MCO_MAXNV], time;
  /* position stage */
  double xpos[MCO_MAXBODY][3], xquat[MCO_MAXBODY][4], xmat[MCO_MAXBODY][9];
  double xipos[MCO_MAXBODY][3], ximat[MCO_MAXBODY][9];
  double xanchor[MCO_MAXJNT][3], xaxis[MCO_MAXJNT][3];
  double geom_xpos[MCO_MAXGEOM][3], geom_xmat[MCO_MAXGEOM][9];
  double site_xpos[MCO_MAXSITE][3], site_xmat[MCO_MAXSITE][9];
  double subtree_com[MCO_MAXBODY][3], cinert[MCO_MAXBODY][10], crb[MCO_MAXBODY][10];
  double cdof[MCO_MAXNV][6];
  double ten_length[MCO_MAXTEN], ten_J[MCO_MAXTEN][MCO_MAXNV];
  double act_length[MCO_MAXU], act_moment[MCO_MAXU][MCO_MAXNV];
  double qM[MCO_MAXNV][MCO_MAXNV], qL[MCO_MAXNV][MCO_MAXNV]; /* dense M and its Cholesky factor */
  /* velocity stage */
  double cvel[MCO_MAXBODY][6], cdof_dot[MCO_MAXNV][6];
  double ten_velocity[MCO_MAXTEN], act_velocity[MCO_MAXU];
  double qfrc_passive[MCO_MAXNV], qfrc_bias[MCO_MAXNV];
  /* actuation / acceleration */
  double act_force[MCO_MAXU], qfrc_actuator[MCO_MAXNV], qfrc_smooth[MCO_MAXNV], qacc_smooth[MCO_MAXNV];
  /* constraints */
  int ncon, nefc, ne, nl;
  mco_contact contact[MCO_MAXCON];
  int efc_type[MCO_MAXEFC], efc_id[MCO_MAXEFC];
  double efc_J[MCO_MAXEFC][MCO_MAXNV], efc_pos[MCO_MAXEFC], efc_margin[MCO_MAXEFC];
  double efc_diagApprox[MCO_MAXEFC], efc_R[MCO_MAXEFC], efc_D[MCO_MAXEFC], efc_KBIP[MCO_MAXEFC][4];
  double efc_vel[MCO_MAXEFC], efc_aref[MCO_MAXEFC], efc_force[MCO_MAXEFC];
  double qfrc_constraint[MCO_MAXNV], qacc[MCO_MAXNV];
  int solver_iter, warning_badstate;
  int nentry, ndrop;             /* entries of the contact list (see mco_model.maxentry); contacts the cap cut off */
  /* mocap bodies (kept last: pyoracle addresses the leading state members by offset) */
  double mocap_pos[MCO_MAXMOCAP][3], mocap_quat[MCO_MAXMOCAP][4];
} mco_data;

/* generic field setters so that a ctypes caller need not mirror the struct layout */
int mco_model_sizeof(void);
int mco_data_sizeof(void);
int mco_model_set_i(mco_model* m, const char* field, const int* v, int n);
int mco_model_set_d(mco_model* m, const char* field, const double* v, int n);
int mco_model_get_d(const mco_model* m, const char* field, double* v, int n);
void mco_model_set_poly(mco_model* m, const double* blob);   /* the caller keeps the blob alive */
int mco_data_get_d(const mco_data* d, const char* field, double* v, int n);
int mco_data_get_i(const mco_data* d, const char* field, int* v, int n);
int mco_data_set_d(mco_data* d, const char* field, const double* v, int n);

void mco_setconst(mco_model* m);                         /* mj_setConst: invweight0, meaninertia */
void mco_reset_data(const mco_model* m, mco_data* d);    /* mj_resetData */
void mco_forward(const mco_model* m, mco_data* d);       /* mj_forward */
void mco_step(const mco_model* m, mco_data* d);          /* mj_step (Euler) */
void mco_jac(const mco_model* m, const mco_data* d, double* jacp, double* jacr,
             const double point[3], int body);           /* mj_jac, 3 x nv row-major each */
void mco_jac_site(const mco_model* m, const mco_data* d, double* jacp, double* jacr, int site);
double mco_energy(const mco_model* m, const mco_data* d, double* potential, double* kinetic);

/* mju_* helpers the env layer uses */
void mco_mat2quat(double quat[4], const double mat[9]);
void mco_negquat(double res[4], const double quat[4]);
void mco_mulquat(double res[4], const double a[4], const double b[4]);
void mco_quat2vel(double res[3], const double quat[4], double dt);
void mco_quat2mat(double mat[9], const double quat[4]);

#ifdef __cplusplus
}
#endif
#endif
