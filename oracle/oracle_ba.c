/*
 * oracle_ba.c — CPU restatement of the reference's bundle adjustment
 *               (Cg2oOptimizer + g2o slam3d types + OptimizationAlgorithmLevenberg + sparse LL').
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under svi_mapper_amd/ may include, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker and
 * as the timed "CPU restatement of the g2o/CHOLMOD path" (never labelled g2o).
 *
 * PARITY UNPINNED: the arithmetic lives in g2o + CHOLMOD ("trunk", unpinned: readme.txt:30-31),
 * neither of which is part of /root/reference or installed here, and the reference ships no tests
 * or golden vectors.  What is restated, and from where:
 *
 *   graph construction rules        src/optimization/Cg2oOptimizer.cpp:982-1073 (edge factories),
 *                                   :1229-1290 (_setAndgetPose), :1383-1466 (_setLandmarkMeasurementsWORLD),
 *                                   :1468-1512 (_applyOptimizationToLandmarks)
 *   solver configuration            src/optimization/Cg2oOptimizer.cpp:83-89 (BlockSolverX, LinearSolverCholmod,
 *                                   OptimizationAlgorithmLevenberg; nothing marginalised => full system, no Schur)
 *   iteration schedule              src/optimization/Cg2oOptimizer.cpp:954-980 (_optimizeUnLimited)
 *   gravity edge                    src/optimization/edge_se3_linear_acceleration.cpp:106-116
 *   minimal pose vector -> SE3      src/vision/CMiniVisionToolbox.cpp:354-377 (same map as g2o fromVectorMQT)
 *   g2o internals (vertex oplus, edge errors and analytic Jacobians, Cauchy kernel, LM control,
 *   plain vs robust chi2)           restated from upstream g2o as summarised in SURVEY.md §8a-6/7 and Appendix B.
 *
 * The linear solve is a scalar up-looking sparse Cholesky of the FULL (3L + 6P) system, landmarks
 * ordered before poses (the fill-reducing order a minimum-degree ordering produces on this
 * structure; it stands in for CHOLMOD's AMD).  Non positive definite => failed LM trial.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { T_XYZ = 0, T_DEPTH = 1, T_DISP = 2 };          /* projection edges (pose, landmark)   */
enum { A_SE3 = 0, A_ACCEL = 1, A_LMLM = 2 };          /* "aux" edges                          */

typedef struct {
    int64_t id;
    double  R[9], t[3];
    int     fixed;
    int     col; /* first scalar column in the system, -1 if fixed */
} opose;

typedef struct {
    int64_t id;
    double  p[3];
    int     fixed;
    int     col;
} olm;

typedef struct {
    int     type, robust;
    int     pose, lm;   /* indices into poses / lms (insertion order) */
    double  z[3];
    double  info[6];    /* upper triangle 00 01 02 11 12 22 */
} oproj;

typedef struct {
    int     type, robust;
    int     a, b;       /* SE3: pose i, pose j; ACCEL: pose, -1; LMLM: lm i, lm j */
    double  z[12];      /* SE3: Z (R,t); ACCEL: a[3]; LMLM: z[3] */
    double  off[12];    /* ACCEL: IMU->LEFT offset */
    double  info[21];   /* SE3: 6x6 upper; else 3x3 upper in [0..5] */
} oaux;

/* int64 -> int open addressing map */
typedef struct { int64_t* k; int* v; size_t cap, n; } omap;

typedef struct {
    /* options */
    double fx, fy, cx, cy, baseline, delta, tau, lo, hi;
    int    max_trials;
    double d_xyz, d_depth, d_disp, d_sane;
    int    accel_numeric; /* 1: g2o's central differences for the gravity edge (default) */
    double imu_off[12];   /* g2o::ParameterSE3Offset eOFFSET_IMUtoLEFT (Cg2oOptimizer.cpp:209-213): R row-major, t */

    opose* P; int np, capp;
    olm*   L; int nl, capl;
    oproj* E; int64_t ne, cape;
    oaux*  A; int na, capa;
    omap   mp, ml;

    /* system */
    int      ready;
    int64_t  n;            /* scalar unknowns */
    int64_t* Ap; int* Ai; double* Ax; /* upper CSC of H */
    int*     eblk;         /* per proj edge: offset of its (lm,pose) block inside pose columns, -1 */
    int*     ablk;         /* per aux SE3 edge: offset of the (i,j) block in the higher pose's columns */
    double*  b;            /* rhs */
    double*  x;            /* increment */
    /* factor */
    int      sym_ok;
    int*     parent; int64_t* Lp; int* Li; double* Lx; int64_t* Lnz;
    int*     flag; int* stack; double* work; double* Cx;
    /* backup */
    double*  bakP; double* bakL;
    /* LM */
    double   lambda, ni;
    double   last_plain, last_robust;
    int      last_trials;
    uint64_t it_total, trials_total, chol_fail;
    /* trace */
    double*  trace; int ntrace, captrace;
} orc_ba;

/* ------------------------------------------------------------------------------------------ */
static uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static void omap_init(omap* m) { m->cap = 1024; m->n = 0; m->k = malloc(m->cap * 8); m->v = malloc(m->cap * 4); for (size_t i = 0; i < m->cap; ++i) m->v[i] = -1; }
static void omap_free(omap* m) { free(m->k); free(m->v); }
static int  omap_get(const omap* m, int64_t key)
{
    size_t h = mix64((uint64_t)key) & (m->cap - 1);
    while (m->v[h] >= 0) { if (m->k[h] == key) return m->v[h]; h = (h + 1) & (m->cap - 1); }
    return -1;
}
static void omap_put(omap* m, int64_t key, int val)
{
    if ((m->n + 1) * 2 > m->cap) {
        omap o = *m; m->cap *= 2; m->n = 0; m->k = malloc(m->cap * 8); m->v = malloc(m->cap * 4);
        for (size_t i = 0; i < m->cap; ++i) m->v[i] = -1;
        for (size_t i = 0; i < o.cap; ++i) if (o.v[i] >= 0) omap_put(m, o.k[i], o.v[i]);
        free(o.k); free(o.v);
    }
    size_t h = mix64((uint64_t)key) & (m->cap - 1);
    while (m->v[h] >= 0) h = (h + 1) & (m->cap - 1);
    m->k[h] = key; m->v[h] = val; m->n++;
}

/* ------------------------------------------------------------------------------------------ */
orc_ba* orc_ba_create(double fx, double fy, double cx, double cy, double baseline)
{
    orc_ba* o = calloc(1, sizeof(orc_ba));
    o->fx = fx; o->fy = fy; o->cx = cx; o->cy = cy; o->baseline = baseline;
    o->delta = 1.0;                 /* RobustKernelCauchy default */
    o->tau = 1e-5; o->lo = 1.0 / 3.0; o->hi = 2.0 / 3.0; o->max_trials = 10; /* g2o LM defaults */
    o->d_xyz = 10.0; o->d_depth = 50.0; o->d_disp = 10000.0; o->d_sane = 1e12; /* Cg2oOptimizer.h:92-95 */
    o->accel_numeric = 1;
    o->imu_off[0] = o->imu_off[4] = o->imu_off[8] = 1.0; /* stereo-only cameras: identity (:100,116) */
    omap_init(&o->mp); omap_init(&o->ml);
    return o;
}

static void free_system(orc_ba* o)
{
    free(o->Ap); free(o->Ai); free(o->Ax); free(o->eblk); free(o->ablk); free(o->b); free(o->x);
    free(o->parent); free(o->Lp); free(o->Li); free(o->Lx); free(o->Lnz); free(o->flag); free(o->stack);
    free(o->work); free(o->Cx); free(o->bakP); free(o->bakL);
    o->Ap = 0; o->Ai = 0; o->Ax = 0; o->eblk = 0; o->ablk = 0; o->b = 0; o->x = 0; o->parent = 0; o->Lp = 0;
    o->Li = 0; o->Lx = 0; o->Lnz = 0; o->flag = 0; o->stack = 0; o->work = 0; o->Cx = 0; o->bakP = 0; o->bakL = 0;
    o->ready = 0; o->sym_ok = 0;
}

void orc_ba_destroy(orc_ba* o)
{
    if (!o) return;
    free_system(o);
    free(o->P); free(o->L); free(o->E); free(o->A); free(o->trace);
    omap_free(&o->mp); omap_free(&o->ml);
    free(o);
}

void orc_ba_set_lm(orc_ba* o, double tau, double lo, double hi, int max_trials, double delta)
{
    o->tau = tau; o->lo = lo; o->hi = hi; o->max_trials = max_trials; o->delta = delta;
}
void orc_ba_set_accel_numeric(orc_ba* o, int on) { o->accel_numeric = on; }
/* CStereoCameraIMU: m_matTransformationIMUtoCAMERA of the LEFT camera becomes the offset parameter every gravity edge
 * refers to (Cg2oOptimizer.cpp:213, :988) */
void orc_ba_set_imu_offset(orc_ba* o, const double off[12]) { memcpy(o->imu_off, off, 96); }

int orc_ba_add_pose(orc_ba* o, int64_t id, const double T[12], int fixed)
{
    if (omap_get(&o->mp, id) >= 0 || omap_get(&o->ml, id) >= 0) return 1;
    if (o->np == o->capp) { o->capp = o->capp ? 2 * o->capp : 256; o->P = realloc(o->P, sizeof(opose) * o->capp); }
    opose* p = &o->P[o->np];
    p->id = id; memcpy(p->R, T, 72); memcpy(p->t, T + 9, 24); p->fixed = fixed; p->col = -1;
    omap_put(&o->mp, id, o->np++);
    o->ready = 0;
    return 0;
}

int orc_ba_add_landmark(orc_ba* o, int64_t id, const double p[3], int fixed)
{
    if (omap_get(&o->mp, id) >= 0 || omap_get(&o->ml, id) >= 0) return 1;
    if (o->nl == o->capl) { o->capl = o->capl ? 2 * o->capl : 1024; o->L = realloc(o->L, sizeof(olm) * o->capl); }
    olm* l = &o->L[o->nl];
    l->id = id; memcpy(l->p, p, 24); l->fixed = fixed; l->col = -1;
    omap_put(&o->ml, id, o->nl++);
    o->ready = 0;
    return 0;
}

int orc_ba_add_edge_proj(orc_ba* o, int type, int64_t pose_id, int64_t lm_id, const double z[3],
                         const double info[6], int robust)
{
    const int ip = omap_get(&o->mp, pose_id), il = omap_get(&o->ml, lm_id);
    if (ip < 0 || il < 0 || type < 0 || type > 2) return 1;
    if (o->ne == o->cape) { o->cape = o->cape ? 2 * o->cape : 4096; o->E = realloc(o->E, sizeof(oproj) * o->cape); }
    oproj* e = &o->E[o->ne++];
    e->type = type; e->robust = robust; e->pose = ip; e->lm = il;
    memcpy(e->z, z, 24); memcpy(e->info, info, 48);
    o->ready = 0;
    return 0;
}

static oaux* new_aux(orc_ba* o)
{
    if (o->na == o->capa) { o->capa = o->capa ? 2 * o->capa : 256; o->A = realloc(o->A, sizeof(oaux) * o->capa); }
    oaux* a = &o->A[o->na++];
    memset(a, 0, sizeof(*a));
    o->ready = 0;
    return a;
}

int orc_ba_add_edge_se3(orc_ba* o, int64_t id_i, int64_t id_j, const double Z[12], const double info[21], int robust)
{
    const int i = omap_get(&o->mp, id_i), j = omap_get(&o->mp, id_j);
    if (i < 0 || j < 0 || i == j) return 1;
    oaux* a = new_aux(o);
    a->type = A_SE3; a->robust = robust; a->a = i; a->b = j;
    memcpy(a->z, Z, 96); memcpy(a->info, info, 21 * 8);
    return 0;
}

int orc_ba_add_edge_accel(orc_ba* o, int64_t pose_id, const double acc[3], const double off[12], const double info[6])
{
    const int i = omap_get(&o->mp, pose_id);
    if (i < 0) return 1;
    oaux* a = new_aux(o);
    a->type = A_ACCEL; a->a = i; a->b = -1;
    memcpy(a->z, acc, 24);
    if (off) memcpy(a->off, off, 96);
    else { a->off[0] = a->off[4] = a->off[8] = 1.0; }
    memcpy(a->info, info, 48);
    return 0;
}

int orc_ba_add_edge_lm_lm(orc_ba* o, int64_t id_i, int64_t id_j, const double z[3], const double info[6], int robust)
{
    const int i = omap_get(&o->ml, id_i), j = omap_get(&o->ml, id_j);
    if (i < 0 || j < 0 || i == j) return 1;
    oaux* a = new_aux(o);
    a->type = A_LMLM; a->robust = robust; a->a = i; a->b = j;
    memcpy(a->z, z, 24); memcpy(a->info, info, 48);
    return 0;
}

/* --------------------------------- small SE3 helpers --------------------------------------- */
static void mat3_mul(const double* A, const double* B, double* C)
{
    double r[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
        r[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(C, r, 72);
}
static void mat3T_mul(const double* A, const double* B, double* C) /* A' B */
{
    double r[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
        r[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
    memcpy(C, r, 72);
}
static void mat3_vec(const double* A, const double* v, double* r)
{
    double t0 = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
    double t1 = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
    double t2 = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    r[0] = t0; r[1] = t1; r[2] = t2;
}
static void mat3T_vec(const double* A, const double* v, double* r)
{
    double t0 = A[0] * v[0] + A[3] * v[1] + A[6] * v[2];
    double t1 = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
    double t2 = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
    r[0] = t0; r[1] = t1; r[2] = t2;
}
/* unit quaternion (w,x,y,z) -> rotation matrix, the textbook formula Eigen uses */
static void quat_to_R(double w, double x, double y, double z, double* R)
{
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
/* rotation matrix -> unit quaternion with w >= 0 (Shepperd's branches; g2o normalises and flips) */
static void R_to_quat(const double* m, double* q /* w x y z */)
{
    double t = m[0] + m[4] + m[8];
    double w, v[3];
    if (t > 0) {
        t = sqrt(t + 1.0); w = 0.5 * t; t = 0.5 / t;
        v[0] = (m[7] - m[5]) * t; v[1] = (m[2] - m[6]) * t; v[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
        v[i] = 0.5 * t; t = 0.5 / t;
        w    = (m[3 * k + j] - m[3 * j + k]) * t;
        v[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        v[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
    const double nrm = sqrt(w * w + v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double s = 1.0 / nrm;
    if (w < 0) s = -s;
    q[0] = w * s; q[1] = v[0] * s; q[2] = v[1] * s; q[3] = v[2] * s;
}
/* g2o VertexSE3::oplusImpl: X <- X * fromVectorMQT(d), d = (dt, dq_xyz)
 * (same map as CMiniVisionToolbox::getTransformationFromVector, CMiniVisionToolbox.cpp:354-377) */
void orc_pose_oplus(const double* R, const double* t, const double* d, double* Rn, double* tn)
{
    double dR[9];
    const double n2 = d[3] * d[3] + d[4] * d[4] + d[5] * d[5];
    const double w2 = 1.0 - n2;
    if (w2 < 0) { memset(dR, 0, 72); dR[0] = dR[4] = dR[8] = 1.0; }
    else quat_to_R(sqrt(w2), d[3], d[4], d[5], dR);
    double rt[3];
    mat3_vec(R, d, rt);
    tn[0] = t[0] + rt[0]; tn[1] = t[1] + rt[1]; tn[2] = t[2] + rt[2];
    mat3_mul(R, dR, Rn);
}
/* entry point used by tests: T (12) oplus d (6) -> Tout (12) */
void orc_se3_oplus(const double T[12], const double d[6], double Tout[12])
{
    orc_pose_oplus(T, T + 9, d, Tout, Tout + 9);
}

static void sym3(const double* u, double* O) /* upper 6 -> full 3x3 */
{
    O[0] = u[0]; O[1] = u[1]; O[2] = u[2];
    O[3] = u[1]; O[4] = u[3]; O[5] = u[4];
    O[6] = u[2]; O[7] = u[4]; O[8] = u[5];
}

/* ------------------------------ projection edges (Appendix B) ------------------------------ */
/* e (3), J9 = d e / d (dt, dq, dp) (3 x 9 row-major). Z = R'(p - t). */
static void proj_eval(const orc_ba* o, int type, const double* R, const double* t, const double* p,
                      const double* z, double* e, double* J9)
{
    double d[3] = { p[0] - t[0], p[1] - t[1], p[2] - t[2] }, Z[3];
    mat3T_vec(R, d, Z);
    double J[27]; /* [-I | 2[Z]x | R'] : g2o EdgeSE3PointXYZ::linearizeOplus */
    if (J9) {
        memset(J, 0, sizeof(J));
        J[0] = J[10] = J[20] = -1.0;
        J[0 * 9 + 4] = -2 * Z[2]; J[0 * 9 + 5] = 2 * Z[1];
        J[1 * 9 + 3] = 2 * Z[2];  J[1 * 9 + 5] = -2 * Z[0];
        J[2 * 9 + 3] = -2 * Z[1]; J[2 * 9 + 4] = 2 * Z[0];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) J[i * 9 + 6 + j] = R[3 * j + i];
    }
    if (type == T_XYZ) {
        e[0] = Z[0] - z[0]; e[1] = Z[1] - z[1]; e[2] = Z[2] - z[2];
        if (J9) memcpy(J9, J, sizeof(J));
        return;
    }
    /* p' = K Z  (w2i = Kcam * w2l) */
    const double px = o->fx * Z[0] + o->cx * Z[2], py = o->fy * Z[1] + o->cy * Z[2], pz = Z[2];
    e[0] = px / pz - z[0];
    e[1] = py / pz - z[1];
    e[2] = (type == T_DEPTH ? pz : 1.0 / pz) - z[2];
    if (J9) {
        double Jp[27];
        for (int c = 0; c < 9; ++c) {
            Jp[c]      = o->fx * J[c] + o->cx * J[18 + c];
            Jp[9 + c]  = o->fy * J[9 + c] + o->cy * J[18 + c];
            Jp[18 + c] = J[18 + c];
        }
        const double iz2 = 1.0 / (pz * pz);
        for (int c = 0; c < 9; ++c) {
            J9[c]      = iz2 * (Jp[c] * pz - px * Jp[18 + c]);
            J9[9 + c]  = iz2 * (Jp[9 + c] * pz - py * Jp[18 + c]);
            J9[18 + c] = (type == T_DEPTH) ? Jp[18 + c] : -iz2 * Jp[18 + c];
        }
    }
}

/* RobustKernelCauchy::robustify */
static void cauchy(double delta, double e2, double* rho0, double* rho1)
{
    const double dsqr = delta * delta, aux = e2 / dsqr + 1.0;
    *rho0 = dsqr * log(aux);
    *rho1 = 1.0 / aux;
}

static double quad3(const double* O, const double* e)
{
    double s = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) s += e[i] * O[3 * i + j] * e[j];
    return s;
}

/* ------------------------------ EdgeSE3 (odometry) ----------------------------------------- */
/* e = toVectorMQT(Z^-1 Xi^-1 Xj); Ji, Jj 6x6 row-major (exact derivative w.r.t. the MQT increments) */
void orc_se3_edge(const double Xi[12], const double Xj[12], const double Z[12], double e[6], double* Ji, double* Jj)
{
    const double *Ri = Xi, *ti = Xi + 9, *Rj = Xj, *tj = Xj + 9, *Rz = Z, *tz = Z + 9;
    /* B = Xi^-1 Xj */
    double Rb[9], tb[3], d[3] = { tj[0] - ti[0], tj[1] - ti[1], tj[2] - ti[2] };
    mat3T_mul(Ri, Rj, Rb); mat3T_vec(Ri, d, tb);
    /* A = Z^-1 = (Rz', -Rz' tz) */
    double Ra[9], ta[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Ra[3 * i + j] = Rz[3 * j + i];
    mat3_vec(Ra, tz, ta); ta[0] = -ta[0]; ta[1] = -ta[1]; ta[2] = -ta[2];
    /* E = A B */
    double Re[9], te[3];
    mat3_mul(Ra, Rb, Re); mat3_vec(Ra, tb, te); te[0] += ta[0]; te[1] += ta[1]; te[2] += ta[2];
    double qe[4];
    R_to_quat(Re, qe);
    e[0] = te[0]; e[1] = te[1]; e[2] = te[2]; e[3] = qe[1]; e[4] = qe[2]; e[5] = qe[3];
    if (!Ji) return;
    /* sign of the canonicalisation: compose unnormalised product qa*qb and compare with qe */
    double qa[4], qb[4];
    R_to_quat(Ra, qa); R_to_quat(Rb, qb);
    /* q = qa (x) qb */
    double wq = qa[0] * qb[0] - (qa[1] * qb[1] + qa[2] * qb[2] + qa[3] * qb[3]);
    double vq[3] = { qa[0] * qb[1] + qb[0] * qa[1] + (qa[2] * qb[3] - qa[3] * qb[2]),
                     qa[0] * qb[2] + qb[0] * qa[2] + (qa[3] * qb[1] - qa[1] * qb[3]),
                     qa[0] * qb[3] + qb[0] * qa[3] + (qa[1] * qb[2] - qa[2] * qb[1]) };
    double sgn = (wq * qe[0] + vq[0] * qe[1] + vq[1] * qe[2] + vq[2] * qe[3]) < 0 ? -1.0 : 1.0;
    memset(Ji, 0, 36 * 8); memset(Jj, 0, 36 * 8);
    /* wrt dj: E' = E * Dj : t' = te + Re dt ; q' = qe (x) (1, dq) */
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Jj[6 * r + c] = Re[3 * r + c];
    {
        const double we = qe[0], *ve = qe + 1;
        const double M[9] = { we, -ve[2], ve[1], ve[2], we, -ve[0], -ve[1], ve[0], we }; /* we I + [ve]x */
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Jj[6 * (3 + r) + 3 + c] = M[3 * r + c];
    }
    /* wrt di: E' = A Di^-1 B : t' = Ra (I - 2[dq]x)(tb - dt) + ta */
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Ji[6 * r + c] = -Ra[3 * r + c];
    {
        const double S[9] = { 0, -tb[2], tb[1], tb[2], 0, -tb[0], -tb[1], tb[0], 0 }; /* [tb]x */
        double RS[9];
        mat3_mul(Ra, S, RS);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Ji[6 * r + 3 + c] = 2 * RS[3 * r + c];
        /* rotation: vec(qa (x) (1,-dq) (x) qb) => -( -vb va' + (wb I - [vb]x)(wa I + [va]x) ) */
        const double wa = qa[0], *va = qa + 1, wb = qb[0], *vb = qb + 1;
        const double Ma[9] = { wa, -va[2], va[1], va[2], wa, -va[0], -va[1], va[0], wa };
        const double Mb[9] = { wb, vb[2], -vb[1], -vb[2], wb, vb[0], vb[1], -vb[0], wb }; /* wb I - [vb]x */
        double MM[9];
        mat3_mul(Mb, Ma, MM);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
            Ji[6 * (3 + r) + 3 + c] = sgn * (vb[r] * va[c] - MM[3 * r + c]);
    }
}

/* ------------------------------ gravity edge ----------------------------------------------- */
static void accel_err(const double* R, const double* off, const double* a, double* e)
{
    double v[3], w[3];
    mat3_vec(off, a, v); /* n2w().linear() = R * R_off */
    mat3_vec(R, v, w);
    e[0] = w[0]; e[1] = w[1]; e[2] = w[2] + 1.0; /* - (0,0,-1)  edge_se3_linear_acceleration.cpp:112 */
}
static void accel_jac(const orc_ba* o, const opose* p, const oaux* a, double* J /*3x6*/)
{
    if (o->accel_numeric) { /* g2o BaseUnaryEdge::linearizeOplus: central differences, delta 1e-9 */
        const double delta = 1e-9, scalar = 1.0 / (2 * delta);
        for (int d = 0; d < 6; ++d) {
            double dv[6] = { 0, 0, 0, 0, 0, 0 }, Rn[9], tn[3], e1[3], e2[3];
            dv[d] = delta;  orc_pose_oplus(p->R, p->t, dv, Rn, tn); accel_err(Rn, a->off, a->z, e1);
            dv[d] = -delta; orc_pose_oplus(p->R, p->t, dv, Rn, tn); accel_err(Rn, a->off, a->z, e2);
            for (int r = 0; r < 3; ++r) J[6 * r + d] = scalar * (e1[r] - e2[r]);
        }
    } else { /* analytic: d/d dq = -2 R [R_off a]x */
        double v[3];
        mat3_vec(a->off, a->z, v);
        const double S[9] = { 0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0 };
        double RS[9];
        mat3_mul(p->R, S, RS);
        memset(J, 0, 18 * 8);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) J[6 * r + 3 + c] = -2 * RS[3 * r + c];
    }
}

/* ------------------------------ reference construction rules ------------------------------- */
/* Cg2oOptimizer::_setAndgetPose + the gravity edge added by the caller (:480, :982-997) */
int orc_ba_add_keyframe(orc_ba* o, int64_t id, int64_t from_id, const double T[12], const double shift[3], const double accel[3])
{
    double X[12];
    memcpy(X, T, 96);
    if (shift) { X[9] += shift[0]; X[10] += shift[1]; X[11] += shift[2]; }
    const int from = omap_get(&o->mp, from_id);
    if (from < 0) return 1;
    if (orc_ba_add_pose(o, id, X, 0)) return 1;
    const opose* pf = &o->P[from];
    /* measurement = Xfrom^-1 * Xcur */
    double Z[12], d[3] = { X[9] - pf->t[0], X[10] - pf->t[1], X[11] - pf->t[2] };
    mat3T_mul(pf->R, X, Z); mat3T_vec(pf->R, d, Z + 9);
    const double s = 1.0 / (1.0 + (Z[9] * Z[9] + Z[10] * Z[10] + Z[11] * Z[11])); /* :1259 */
    double info[21]; memset(info, 0, sizeof(info));
    /* upper triangle row-major of 6x6: diagonal entries at 0,6,11,15,18,20 */
    info[0] = info[6] = info[11] = 100000.0 * s; info[15] = info[18] = info[20] = 100000.0; /* m_matInformationPose */
    if (orc_ba_add_edge_se3(o, from_id, id, Z, info, 0)) return 1;
    const double a0[3] = { 0, 0, 0 }, I3[6] = { 1, 0, 0, 1, 0, 1 };
    return orc_ba_add_edge_accel(o, id, accel ? accel : a0, o->imu_off, I3);
}

/* Cg2oOptimizer::_setLandmarkMeasurementsWORLD (:1383-1466) with the factories (:999-1073) */
int orc_ba_add_measurements(orc_ba* o, int64_t pose_id, int64_t n, const int64_t* lm_id, const float* uvL,
                            const float* uvR, const double* xyz, int64_t stored[3])
{
    const int ip = omap_get(&o->mp, pose_id);
    if (ip < 0) return 1;
    int64_t cnt[3] = { 0, 0, 0 };
    for (int64_t m = 0; m < n; ++m) {
        const int il = omap_get(&o->ml, lm_id[m]);
        if (il < 0) continue; /* landmark not in graph: silently skipped (:1393-1396) */
        const opose* P = &o->P[ip];
        const olm*   L = &o->L[il];
        const double* pm = xyz + 3 * m;
        double d[3] = { L->p[0] - P->t[0], L->p[1] - P->t[1], L->p[2] - P->t[2] }, pe[3];
        mat3T_vec(P->R, d, pe);
        const double l2abs = pm[0] * pm[0] + pm[1] * pm[1] + pm[2] * pm[2];
        const double l2rel = (pe[0] * pe[0] + pe[1] * pe[1] + pe[2] * pe[2]) / l2abs;
        if (!(0.75 < l2rel && 1.25 > l2rel)) continue;             /* :1409 */
        const double w = 1.0 / pm[2];                                /* :1412 */
        if (o->d_xyz > l2abs) {                                      /* :1415 */
            const double info[6] = { w * 1000, 0, 0, w * 1000, 0, w * 1000 };
            orc_ba_add_edge_proj(o, T_XYZ, pose_id, lm_id[m], pm, info, 1);
            cnt[0]++;
        } else if (o->d_depth > l2abs) {                             /* :1426 */
            const double z[3] = { (double)uvL[2 * m], (double)uvL[2 * m + 1], pm[2] };
            const double info[6] = { w, 0, 0, w, 0, w * 100 };
            orc_ba_add_edge_proj(o, T_DEPTH, pose_id, lm_id[m], z, info, 1);
            cnt[1]++;
        } else if (o->d_disp > l2abs) {                              /* :1437 */
            const double disp = (double)(uvL[2 * m] - uvR[2 * m]);   /* float subtraction, then promoted (:1440) */
            if (1.0 < disp) {                                        /* :1443 */
                const double z[3] = { (double)uvL[2 * m], (double)uvL[2 * m + 1], disp / (o->fx * o->baseline) }; /* :1058 */
                const double info[6] = { w, 0, 0, w, 0, w * 1000 };
                orc_ba_add_edge_proj(o, T_DISP, pose_id, lm_id[m], z, info, 1);
                cnt[2]++;
            }
        }
    }
    if (stored) { stored[0] = cnt[0]; stored[1] = cnt[1]; stored[2] = cnt[2]; }
    return 0;
}

/* ------------------------------ system structure ------------------------------------------- */
typedef struct { int64_t key; int idx; } kv;
static int kv_cmp(const void* a, const void* b)
{
    const kv* x = a; const kv* y = b;
    return x->key < y->key ? -1 : x->key > y->key ? 1 : x->idx - y->idx;
}

/* Build column numbering (landmarks by ascending id, then poses by ascending id), the upper CSC
 * pattern of H and the per-edge block offsets. */
int orc_ba_initialize(orc_ba* o)
{
    free_system(o);
    /* LMLM edges must have exactly one free end (the reference fixes the reference landmark, :445) */
    for (int a = 0; a < o->na; ++a) if (o->A[a].type == A_LMLM) {
        const int fi = o->L[o->A[a].a].fixed, fj = o->L[o->A[a].b].fixed;
        if (!fi && !fj) return 2;
    }
    kv* ol = malloc(sizeof(kv) * (o->nl + 1)); kv* op = malloc(sizeof(kv) * (o->np + 1));
    for (int i = 0; i < o->nl; ++i) { ol[i].key = o->L[i].id; ol[i].idx = i; }
    for (int i = 0; i < o->np; ++i) { op[i].key = o->P[i].id; op[i].idx = i; }
    qsort(ol, o->nl, sizeof(kv), kv_cmp); qsort(op, o->np, sizeof(kv), kv_cmp);
    int64_t n = 0;
    for (int i = 0; i < o->nl; ++i) { olm* l = &o->L[ol[i].idx]; if (l->fixed) l->col = -1; else { l->col = (int)n; n += 3; } }
    const int64_t n_lm = n;
    for (int i = 0; i < o->np; ++i) { opose* p = &o->P[op[i].idx]; if (p->fixed) p->col = -1; else { p->col = (int)n; n += 6; } }
    o->n = n;
    free(ol);

    /* off-diagonal blocks: key = colblock_startcol * n + rowblock_startcol  (row < col) */
    int64_t nb = 0, capb = o->ne + o->na + 16;
    kv* blk = malloc(sizeof(kv) * capb);
    for (int64_t e = 0; e < o->ne; ++e) {
        const int cl = o->L[o->E[e].lm].col, cp = o->P[o->E[e].pose].col;
        if (cl >= 0 && cp >= 0) { blk[nb].key = (int64_t)cp * n + cl; blk[nb].idx = 3; nb++; }
    }
    for (int a = 0; a < o->na; ++a) if (o->A[a].type == A_SE3) {
        int ci = o->P[o->A[a].a].col, cj = o->P[o->A[a].b].col;
        if (ci >= 0 && cj >= 0) {
            const int lo = ci < cj ? ci : cj, hi = ci < cj ? cj : ci;
            blk[nb].key = (int64_t)hi * n + lo; blk[nb].idx = 6; nb++;
        }
    }
    qsort(blk, nb, sizeof(kv), kv_cmp);
    int64_t nu = 0;
    for (int64_t i = 0; i < nb; ++i) if (i == 0 || blk[i].key != blk[nu - 1].key) blk[nu++] = blk[i];
    /* per pose column-block: rows above the diagonal */
    int64_t* above = calloc(n + 1, 8); /* indexed by start col of the col block */
    for (int64_t i = 0; i < nu; ++i) above[blk[i].key / n] += blk[i].idx;
    o->Ap = malloc(8 * (n + 1));
    int64_t nnz = 0;
    /* landmark columns: only the diagonal block */
    for (int64_t c = 0; c < n_lm; ++c) { o->Ap[c] = nnz; nnz += (c % 3) + 1; }
    for (int i = 0; i < o->np; ++i) {
        const opose* p = &o->P[op[i].idx];
        if (p->col < 0) continue;
        for (int k = 0; k < 6; ++k) { o->Ap[p->col + k] = nnz; nnz += above[p->col] + k + 1; }
    }
    o->Ap[n] = nnz;
    o->Ai = malloc(4 * nnz); o->Ax = malloc(8 * nnz); o->Cx = malloc(8 * nnz);
    /* block offsets inside the pose columns: running row offset per col block */
    int64_t* boff = malloc(8 * (nu + 1));
    {
        int64_t run = 0, cur = -1;
        for (int64_t i = 0; i < nu; ++i) {
            const int64_t cb = blk[i].key / n;
            if (cb != cur) { cur = cb; run = 0; }
            boff[i] = run; run += blk[i].idx;
        }
    }
    /* fill row indices */
    for (int64_t c = 0; c < n_lm; ++c) { const int64_t c0 = c - c % 3; for (int k = 0; k <= c % 3; ++k) o->Ai[o->Ap[c] + k] = (int)(c0 + k); }
    for (int64_t i = 0; i < nu; ++i) {
        const int64_t cb = blk[i].key / n, rb = blk[i].key % n;
        for (int k = 0; k < 6; ++k) for (int r = 0; r < blk[i].idx; ++r) o->Ai[o->Ap[cb + k] + boff[i] + r] = (int)(rb + r);
    }
    for (int i = 0; i < o->np; ++i) {
        const opose* p = &o->P[op[i].idx];
        if (p->col < 0) continue;
        for (int k = 0; k < 6; ++k) for (int r = 0; r <= k; ++r) o->Ai[o->Ap[p->col + k] + above[p->col] + r] = p->col + r;
    }
    free(op);
    /* per edge block offsets by binary search */
    o->eblk = malloc(4 * (o->ne + 1));
    for (int64_t e = 0; e < o->ne; ++e) {
        const int cl = o->L[o->E[e].lm].col, cp = o->P[o->E[e].pose].col;
        o->eblk[e] = -1;
        if (cl < 0 || cp < 0) continue;
        const int64_t key = (int64_t)cp * n + cl;
        int64_t lo = 0, hi = nu - 1;
        while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (blk[mid].key < key) lo = mid + 1; else hi = mid; }
        o->eblk[e] = (int)boff[lo];
    }
    o->ablk = malloc(4 * (o->na + 1));
    for (int a = 0; a < o->na; ++a) {
        o->ablk[a] = -1;
        if (o->A[a].type != A_SE3) continue;
        int ci = o->P[o->A[a].a].col, cj = o->P[o->A[a].b].col;
        if (ci < 0 || cj < 0) continue;
        const int lo_c = ci < cj ? ci : cj, hi_c = ci < cj ? cj : ci;
        const int64_t key = (int64_t)hi_c * n + lo_c;
        int64_t lo = 0, hi = nu - 1;
        while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (blk[mid].key < key) lo = mid + 1; else hi = mid; }
        o->ablk[a] = (int)boff[lo];
    }
    free(blk); free(boff); free(above);
    o->b = calloc(n + 1, 8); o->x = calloc(n + 1, 8);
    o->parent = malloc(4 * (n + 1)); o->Lp = malloc(8 * (n + 2)); o->Lnz = malloc(8 * (n + 1));
    o->flag = malloc(4 * (n + 1)); o->stack = malloc(4 * (n + 1)); o->work = calloc(n + 1, 8);
    o->bakP = malloc(sizeof(double) * 12 * (o->np + 1)); o->bakL = malloc(sizeof(double) * 3 * (o->nl + 1));
    o->ready = 1;
    return 0;
}

/* ------------------------------ errors and linearisation ----------------------------------- */
/* g2o computeActiveErrors + activeRobustChi2 + OptimizableGraph::chi2 */
static void compute_errors(orc_ba* o, double* plain, double* robust)
{
    double sp = 0, sr = 0;
    for (int64_t i = 0; i < o->ne; ++i) {
        const oproj* e = &o->E[i];
        double er[3], O[9];
        proj_eval(o, e->type, o->P[e->pose].R, o->P[e->pose].t, o->L[e->lm].p, e->z, er, 0);
        sym3(e->info, O);
        const double c = quad3(O, er);
        sp += c;
        if (e->robust) { double r0, r1; cauchy(o->delta, c, &r0, &r1); sr += r0; } else sr += c;
    }
    for (int i = 0; i < o->na; ++i) {
        const oaux* a = &o->A[i];
        double c = 0;
        if (a->type == A_SE3) {
            double Xi[12], Xj[12], e[6];
            memcpy(Xi, o->P[a->a].R, 72); memcpy(Xi + 9, o->P[a->a].t, 24);
            memcpy(Xj, o->P[a->b].R, 72); memcpy(Xj + 9, o->P[a->b].t, 24);
            orc_se3_edge(Xi, Xj, a->z, e, 0, 0);
            int k = 0;
            for (int r = 0; r < 6; ++r) for (int cc = r; cc < 6; ++cc, ++k) c += (r == cc ? 1.0 : 2.0) * e[r] * a->info[k] * e[cc];
        } else if (a->type == A_ACCEL) {
            double e[3], O[9];
            accel_err(o->P[a->a].R, a->off, a->z, e); sym3(a->info, O); c = quad3(O, e);
        } else {
            double e[3], O[9];
            const double *pi = o->L[a->a].p, *pj = o->L[a->b].p;
            for (int r = 0; r < 3; ++r) e[r] = pj[r] - pi[r] - a->z[r];
            sym3(a->info, O); c = quad3(O, e);
        }
        sp += c;
        if (a->robust) { double r0, r1; cauchy(o->delta, c, &r0, &r1); sr += r0; } else sr += c;
    }
    *plain = sp; *robust = sr;
    o->last_plain = sp; o->last_robust = sr;
}

/* add a dense block M (nr x nc, row-major) at rows r0.., cols c0.. of the upper CSC; for diagonal
 * blocks only the upper triangle is stored */
static inline void add_diag_block(orc_ba* o, int c0, int dim, const double* M, int64_t above)
{
    for (int c = 0; c < dim; ++c) {
        double* col = o->Ax + o->Ap[c0 + c] + above;
        for (int r = 0; r <= c; ++r) col[r] += M[r * dim + c];
    }
}

/* g2o BlockSolver::buildSystem: H = sum J' (rho1 Omega) J, b = - sum J' rho1 Omega e */
static void build_system(orc_ba* o)
{
    memset(o->Ax, 0, 8 * o->Ap[o->n]);
    memset(o->b, 0, 8 * o->n);
    for (int64_t i = 0; i < o->ne; ++i) {
        const oproj* e = &o->E[i];
        const opose* P = &o->P[e->pose];
        const olm*   L = &o->L[e->lm];
        if (P->col < 0 && L->col < 0) continue;
        double er[3], J[27], O[9];
        proj_eval(o, e->type, P->R, P->t, L->p, e->z, er, J);
        sym3(e->info, O);
        double w = 1.0;
        if (e->robust) { double r0; cauchy(o->delta, quad3(O, er), &r0, &w); }
        for (int k = 0; k < 9; ++k) O[k] *= w;
        double Oe[3]; mat3_vec(O, er, Oe);
        /* OJ = Omega_w * J  (3x9) */
        double OJ[27];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 9; ++c)
            OJ[9 * r + c] = O[3 * r] * J[c] + O[3 * r + 1] * J[9 + c] + O[3 * r + 2] * J[18 + c];
        if (P->col >= 0) {
            double H[36];
            for (int a = 0; a < 6; ++a) for (int c = 0; c < 6; ++c)
                H[6 * a + c] = J[a] * OJ[c] + J[9 + a] * OJ[9 + c] + J[18 + a] * OJ[18 + c];
            const int64_t above = o->Ap[P->col + 1] - o->Ap[P->col] - 1;
            add_diag_block(o, P->col, 6, H, above);
            for (int a = 0; a < 6; ++a) o->b[P->col + a] -= J[a] * Oe[0] + J[9 + a] * Oe[1] + J[18 + a] * Oe[2];
        }
        if (L->col >= 0) {
            double H[9];
            for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c)
                H[3 * a + c] = J[6 + a] * OJ[6 + c] + J[15 + a] * OJ[15 + c] + J[24 + a] * OJ[24 + c];
            add_diag_block(o, L->col, 3, H, 0);
            for (int a = 0; a < 3; ++a) o->b[L->col + a] -= J[6 + a] * Oe[0] + J[15 + a] * Oe[1] + J[24 + a] * Oe[2];
        }
        if (P->col >= 0 && L->col >= 0) { /* H_lp (3 rows) x (6 cols) = J_l' Omega J_p */
            const int off = o->eblk[i];
            for (int c = 0; c < 6; ++c) {
                double* col = o->Ax + o->Ap[P->col + c] + off;
                for (int r = 0; r < 3; ++r) col[r] += J[6 + r] * OJ[c] + J[15 + r] * OJ[9 + c] + J[24 + r] * OJ[18 + c];
            }
        }
    }
    for (int i = 0; i < o->na; ++i) {
        const oaux* a = &o->A[i];
        if (a->type == A_SE3) {
            const opose *Pi = &o->P[a->a], *Pj = &o->P[a->b];
            if (Pi->col < 0 && Pj->col < 0) continue;
            double Xi[12], Xj[12], e[6], Ji[36], Jj[36], O[36];
            memcpy(Xi, Pi->R, 72); memcpy(Xi + 9, Pi->t, 24); memcpy(Xj, Pj->R, 72); memcpy(Xj + 9, Pj->t, 24);
            orc_se3_edge(Xi, Xj, a->z, e, Ji, Jj);
            int k = 0;
            for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c, ++k) O[6 * r + c] = O[6 * c + r] = a->info[k];
            double w = 1.0;
            if (a->robust) {
                double c2 = 0, r0;
                for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) c2 += e[r] * O[6 * r + c] * e[c];
                cauchy(o->delta, c2, &r0, &w);
            }
            double Oe[6], OJi[36], OJj[36];
            for (int r = 0; r < 6; ++r) {
                double s = 0;
                for (int c = 0; c < 6; ++c) s += w * O[6 * r + c] * e[c];
                Oe[r] = s;
                for (int c = 0; c < 6; ++c) {
                    double si = 0, sj = 0;
                    for (int q = 0; q < 6; ++q) { si += w * O[6 * r + q] * Ji[6 * q + c]; sj += w * O[6 * r + q] * Jj[6 * q + c]; }
                    OJi[6 * r + c] = si; OJj[6 * r + c] = sj;
                }
            }
            double H[36];
            if (Pi->col >= 0) {
                for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { double s = 0; for (int q = 0; q < 6; ++q) s += Ji[6 * q + r] * OJi[6 * q + c]; H[6 * r + c] = s; }
                add_diag_block(o, Pi->col, 6, H, o->Ap[Pi->col + 1] - o->Ap[Pi->col] - 1);
                for (int r = 0; r < 6; ++r) { double s = 0; for (int q = 0; q < 6; ++q) s += Ji[6 * q + r] * Oe[q]; o->b[Pi->col + r] -= s; }
            }
            if (Pj->col >= 0) {
                for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { double s = 0; for (int q = 0; q < 6; ++q) s += Jj[6 * q + r] * OJj[6 * q + c]; H[6 * r + c] = s; }
                add_diag_block(o, Pj->col, 6, H, o->Ap[Pj->col + 1] - o->Ap[Pj->col] - 1);
                for (int r = 0; r < 6; ++r) { double s = 0; for (int q = 0; q < 6; ++q) s += Jj[6 * q + r] * Oe[q]; o->b[Pj->col + r] -= s; }
            }
            if (Pi->col >= 0 && Pj->col >= 0) {
                /* block (row vertex lo, col vertex hi) = J_lo' Omega J_hi */
                const int i_is_lo = Pi->col < Pj->col;
                const double* Jlo = i_is_lo ? Ji : Jj; const double* OJhi = i_is_lo ? OJj : OJi;
                const int chi = i_is_lo ? Pj->col : Pi->col;
                for (int c = 0; c < 6; ++c) {
                    double* col = o->Ax + o->Ap[chi + c] + o->ablk[i];
                    for (int r = 0; r < 6; ++r) { double s = 0; for (int q = 0; q < 6; ++q) s += Jlo[6 * q + r] * OJhi[6 * q + c]; col[r] += s; }
                }
            }
        } else if (a->type == A_ACCEL) {
            const opose* P = &o->P[a->a];
            if (P->col < 0) continue;
            double e[3], J[18], O[9], Oe[3], H[36];
            accel_err(P->R, a->off, a->z, e); accel_jac(o, P, a, J); sym3(a->info, O); mat3_vec(O, e, Oe);
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
                double s = 0;
                for (int q = 0; q < 3; ++q) for (int p = 0; p < 3; ++p) s += J[6 * q + r] * O[3 * q + p] * J[6 * p + c];
                H[6 * r + c] = s;
            }
            add_diag_block(o, P->col, 6, H, o->Ap[P->col + 1] - o->Ap[P->col] - 1);
            for (int r = 0; r < 6; ++r) o->b[P->col + r] -= J[r] * Oe[0] + J[6 + r] * Oe[1] + J[12 + r] * Oe[2];
        } else { /* EdgePointXYZ: e = pj - pi - z, Ji = -I, Jj = +I; one end fixed */
            const olm *Li = &o->L[a->a], *Lj = &o->L[a->b];
            double e[3], O[9], Oe[3];
            for (int r = 0; r < 3; ++r) e[r] = Lj->p[r] - Li->p[r] - a->z[r];
            sym3(a->info, O);
            double w = 1.0;
            if (a->robust) { double r0; cauchy(o->delta, quad3(O, e), &r0, &w); }
            for (int k = 0; k < 9; ++k) O[k] *= w;
            mat3_vec(O, e, Oe);
            if (Li->col >= 0) { add_diag_block(o, Li->col, 3, O, 0); for (int r = 0; r < 3; ++r) o->b[Li->col + r] += Oe[r]; }
            if (Lj->col >= 0) { add_diag_block(o, Lj->col, 3, O, 0); for (int r = 0; r < 3; ++r) o->b[Lj->col + r] -= Oe[r]; }
        }
    }
}

/* ------------------------------ sparse Cholesky (up-looking) ------------------------------- */
static void symbolic(orc_ba* o)
{
    const int64_t n = o->n;
    int* anc = o->stack; /* reuse as ancestor */
    for (int64_t k = 0; k < n; ++k) {
        o->parent[k] = -1; anc[k] = -1;
        for (int64_t p = o->Ap[k]; p < o->Ap[k + 1]; ++p) {
            int i = o->Ai[p];
            while (i != -1 && i < k) {
                const int nx = anc[i];
                anc[i] = (int)k;
                if (nx == -1) o->parent[i] = (int)k;
                i = nx;
            }
        }
    }
    /* column counts by walking row subtrees */
    for (int64_t k = 0; k < n; ++k) { o->Lnz[k] = 1; o->flag[k] = -1; }
    for (int64_t k = 0; k < n; ++k) {
        o->flag[k] = (int)k;
        for (int64_t p = o->Ap[k]; p < o->Ap[k + 1]; ++p) {
            int i = o->Ai[p];
            while (i < k && o->flag[i] != k) { o->Lnz[i]++; o->flag[i] = (int)k; i = o->parent[i]; }
        }
    }
    o->Lp[0] = 0;
    for (int64_t k = 0; k < n; ++k) o->Lp[k + 1] = o->Lp[k] + o->Lnz[k];
    free(o->Li); free(o->Lx);
    o->Li = malloc(4 * (o->Lp[n] + 1)); o->Lx = malloc(8 * (o->Lp[n] + 1));
    o->sym_ok = 1;
}

/* factor C (upper CSC values Cx on the pattern of A); returns 0 ok, 1 not positive definite */
static int numeric(orc_ba* o)
{
    const int64_t n = o->n;
    double* x = o->work;
    for (int64_t k = 0; k < n; ++k) { o->Lnz[k] = 0; o->flag[k] = -1; }
    for (int64_t k = 0; k < n; ++k) {
        /* pattern of row k of L: reach of A(0:k-1,k) in the etree, topological order */
        int top = (int)n;
        o->flag[k] = (int)k;
        double d = 0;
        for (int64_t p = o->Ap[k]; p < o->Ap[k + 1]; ++p) {
            int i = o->Ai[p];
            if (i == k) { d = o->Cx[p]; continue; }
            x[i] = o->Cx[p];
            int len = 0;
            while (o->flag[i] != k) { o->stack[len++] = i; o->flag[i] = (int)k; i = o->parent[i]; }
            while (len > 0) o->stack[--top] = o->stack[--len];
        }
        for (; top < n; ++top) {
            const int j = o->stack[top];
            const int64_t pj = o->Lp[j];
            const double lkj = x[j] / o->Lx[pj];
            x[j] = 0;
            const int64_t pe = pj + o->Lnz[j];
            for (int64_t p = pj + 1; p < pe; ++p) x[o->Li[p]] -= o->Lx[p] * lkj;
            d -= lkj * lkj;
            o->Li[pe] = (int)k; o->Lx[pe] = lkj; o->Lnz[j]++;
        }
        if (!(d > 0) || !isfinite(d)) { /* leave work vector clean */
            memset(x, 0, 8 * n);
            return 1;
        }
        o->Li[o->Lp[k]] = (int)k; o->Lx[o->Lp[k]] = sqrt(d); o->Lnz[k] = 1;
    }
    return 0;
}

static void chol_solve(const orc_ba* o, double* x)
{
    const int64_t n = o->n;
    for (int64_t j = 0; j < n; ++j) {
        x[j] /= o->Lx[o->Lp[j]];
        for (int64_t p = o->Lp[j] + 1; p < o->Lp[j + 1]; ++p) x[o->Li[p]] -= o->Lx[p] * x[j];
    }
    for (int64_t j = n - 1; j >= 0; --j) {
        for (int64_t p = o->Lp[j] + 1; p < o->Lp[j + 1]; ++p) x[j] -= o->Lx[p] * x[o->Li[p]];
        x[j] /= o->Lx[o->Lp[j]];
    }
}

/* solve (H + lambda I) x = b; 0 ok */
static int solve_damped(orc_ba* o, double lambda)
{
    if (!o->sym_ok) symbolic(o);
    memcpy(o->Cx, o->Ax, 8 * o->Ap[o->n]);
    for (int64_t c = 0; c < o->n; ++c) o->Cx[o->Ap[c + 1] - 1] += lambda; /* diagonal = last entry of the column */
    if (numeric(o)) { o->chol_fail++; return 1; }
    memcpy(o->x, o->b, 8 * o->n);
    chol_solve(o, o->x);
    return 0;
}

/* ------------------------------ LM (g2o OptimizationAlgorithmLevenberg) -------------------- */
static void push_state(orc_ba* o)
{
    for (int i = 0; i < o->np; ++i) { memcpy(o->bakP + 12 * i, o->P[i].R, 72); memcpy(o->bakP + 12 * i + 9, o->P[i].t, 24); }
    for (int i = 0; i < o->nl; ++i) memcpy(o->bakL + 3 * i, o->L[i].p, 24);
}
static void pop_state(orc_ba* o)
{
    for (int i = 0; i < o->np; ++i) { memcpy(o->P[i].R, o->bakP + 12 * i, 72); memcpy(o->P[i].t, o->bakP + 12 * i + 9, 24); }
    for (int i = 0; i < o->nl; ++i) memcpy(o->L[i].p, o->bakL + 3 * i, 24);
}
static void apply_update(orc_ba* o)
{
    for (int i = 0; i < o->np; ++i) if (o->P[i].col >= 0) {
        double Rn[9], tn[3];
        orc_pose_oplus(o->P[i].R, o->P[i].t, o->x + o->P[i].col, Rn, tn);
        memcpy(o->P[i].R, Rn, 72); memcpy(o->P[i].t, tn, 24);
    }
    for (int i = 0; i < o->nl; ++i) if (o->L[i].col >= 0) for (int k = 0; k < 3; ++k) o->L[i].p[k] += o->x[o->L[i].col + k];
}

static void trace_push(orc_ba* o, double chi0, double chi1, double lambda, int trials, int ok)
{
    if (o->ntrace + 5 > o->captrace) { o->captrace = o->captrace ? 2 * o->captrace : 640; o->trace = realloc(o->trace, 8 * o->captrace); }
    double* t = o->trace + o->ntrace;
    t[0] = chi0; t[1] = chi1; t[2] = lambda; t[3] = trials; t[4] = ok;
    o->ntrace += 5;
}

/* one g2o SparseOptimizer::optimize(iterations) call. returns iterations performed, <0 on error */
int orc_ba_optimize(orc_ba* o, int iterations)
{
    if (!o->ready) return -1;
    o->sym_ok = 0; /* _algorithm->init(): structure and factor analysis are rebuilt per call */
    int done = 0;
    for (int it = 0; it < iterations; ++it) {
        double plain, chi;
        compute_errors(o, &plain, &chi);
        double temp = chi;
        build_system(o);
        if (it == 0) { /* computeLambdaInit: tau * max |H_jj| over free vertices */
            double mx = 0;
            for (int64_t c = 0; c < o->n; ++c) { const double v = fabs(o->Ax[o->Ap[c + 1] - 1]); if (v > mx) mx = v; }
            o->lambda = o->tau * mx; o->ni = 2;
        }
        const double chi_before = chi;
        double rho = 0;
        int q = 0, stop_inf = 0;
        do {
            push_state(o);
            const int fail = solve_damped(o, o->lambda);
            apply_update(o); /* with a failed solve g2o applies the stale x; it is popped again below */
            compute_errors(o, &plain, &temp);
            if (fail) temp = 1.79769313486231570815e+308; /* std::numeric_limits<double>::max() */
            rho = chi - temp;
            double scale = 0;
            for (int64_t j = 0; j < o->n; ++j) scale += o->x[j] * (o->lambda * o->x[j] + o->b[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(temp)) {
                double alpha = 1.0 - pow(2 * rho - 1, 3);
                alpha = alpha < o->hi ? alpha : o->hi;
                const double sf = o->lo > alpha ? o->lo : alpha;
                o->lambda *= sf; o->ni = 2; chi = temp;
            } else {
                o->lambda *= o->ni; o->ni *= 2;
                pop_state(o);
                if (!isfinite(o->lambda)) { stop_inf = 1; break; } /* g2o breaks before qmax++ */
            }
            q++;
        } while (rho < 0 && q < o->max_trials);
        o->last_trials = q; o->it_total++; o->trials_total += q; done++;
        trace_push(o, chi_before, chi, o->lambda, q, !(q == o->max_trials || rho == 0 || stop_inf));
        if (q == o->max_trials || rho == 0 || stop_inf) break; /* SolverResult::Terminate */
    }
    return done;
}

/* Cg2oOptimizer::_optimizeUnLimited (Cg2oOptimizer.cpp:954-980) */
int orc_ba_optimize_until(orc_ba* o, double ratio, int first, int block, uint64_t* nominal, uint64_t* executed)
{
    if (!o->ready) return 1;
    uint64_t nom = 0, exe = 0;
    int r = orc_ba_optimize(o, first); if (r < 0) return 1;
    nom += first; exe += r;
    double prev = 1.1 * o->last_plain;
    while (ratio > o->last_plain / prev) {
        prev = o->last_plain;
        r = orc_ba_optimize(o, block); if (r < 0) return 1;
        nom += block; exe += r;
    }
    if (nominal) *nominal = nom;
    if (executed) *executed = exe;
    return 0;
}

/* ------------------------------ accessors --------------------------------------------------- */
void orc_ba_chi2(orc_ba* o, double* plain, double* robust)
{
    double p, r;
    compute_errors(o, &p, &r);
    if (plain) *plain = p;
    if (robust) *robust = r;
}
double orc_ba_last_plain_chi2(const orc_ba* o) { return o->last_plain; }
double orc_ba_lambda(const orc_ba* o) { return o->lambda; }
int64_t orc_ba_num_poses(const orc_ba* o) { return o->np; }
int64_t orc_ba_num_landmarks(const orc_ba* o) { return o->nl; }
int64_t orc_ba_num_edges(const orc_ba* o) { return o->ne; }
int64_t orc_ba_num_aux(const orc_ba* o) { return o->na; }
int64_t orc_ba_system_size(const orc_ba* o) { return o->n; }
uint64_t orc_ba_iterations(const orc_ba* o) { return o->it_total; }
uint64_t orc_ba_trials(const orc_ba* o) { return o->trials_total; }

int orc_ba_get_pose(const orc_ba* o, int64_t id, double T[12])
{
    const int i = omap_get(&o->mp, id);
    if (i < 0) return 1;
    memcpy(T, o->P[i].R, 72); memcpy(T + 9, o->P[i].t, 24);
    return 0;
}
int orc_ba_get_landmark(const orc_ba* o, int64_t id, double p[3])
{
    const int i = omap_get(&o->ml, id);
    if (i < 0) return 1;
    memcpy(p, o->L[i].p, 24);
    return 0;
}
/* insertion order */
void orc_ba_get_poses(const orc_ba* o, int64_t* ids, double* T)
{
    for (int i = 0; i < o->np; ++i) { if (ids) ids[i] = o->P[i].id; memcpy(T + 12 * i, o->P[i].R, 72); memcpy(T + 12 * i + 9, o->P[i].t, 24); }
}
void orc_ba_get_landmarks(const orc_ba* o, int64_t* ids, double* p)
{
    for (int i = 0; i < o->nl; ++i) { if (ids) ids[i] = o->L[i].id; memcpy(p + 3 * i, o->L[i].p, 24); }
}
/* projection edges in insertion order: type, pose id, lm id, z, info */
void orc_ba_get_edges(const orc_ba* o, int32_t* type, int64_t* pose_id, int64_t* lm_id, double* z, double* info)
{
    for (int64_t i = 0; i < o->ne; ++i) {
        const oproj* e = &o->E[i];
        type[i] = e->type; pose_id[i] = o->P[e->pose].id; lm_id[i] = o->L[e->lm].id;
        memcpy(z + 3 * i, e->z, 24); memcpy(info + 6 * i, e->info, 48);
    }
}
/* aux edges in insertion order: type (0 se3, 1 accel, 2 lmlm), ids, z (12), info (21) */
void orc_ba_get_aux(const orc_ba* o, int32_t* type, int64_t* id_a, int64_t* id_b, double* z, double* info)
{
    for (int i = 0; i < o->na; ++i) {
        const oaux* a = &o->A[i];
        type[i] = a->type;
        if (a->type == A_LMLM) { id_a[i] = o->L[a->a].id; id_b[i] = o->L[a->b].id; }
        else { id_a[i] = o->P[a->a].id; id_b[i] = a->b >= 0 ? o->P[a->b].id : -1; }
        memcpy(z + 12 * i, a->z, 96); memcpy(info + 21 * i, a->info, 21 * 8);
    }
}
/* per projection edge: error and Jacobians at the current estimate */
void orc_ba_edge_jacobians(const orc_ba* o, double* err, double* Jp, double* Jl)
{
    for (int64_t i = 0; i < o->ne; ++i) {
        const oproj* e = &o->E[i];
        double J[27];
        proj_eval(o, e->type, o->P[e->pose].R, o->P[e->pose].t, o->L[e->lm].p, e->z, err + 3 * i, J);
        for (int r = 0; r < 3; ++r) { memcpy(Jp + 18 * i + 6 * r, J + 9 * r, 48); memcpy(Jl + 9 * i + 3 * r, J + 9 * r + 6, 24); }
    }
}
/* pose-only edges in insertion order among their kind, at the current estimate: EdgeSE3 error (6) and the two 6x6
 * Jacobians; gravity edge error (3) and its 3x6 Jacobian (numeric or analytic as orc_ba_set_accel_numeric says) */
void orc_ba_aux_jacobians(const orc_ba* o, double* se3_err, double* se3_Ji, double* se3_Jj, double* acc_err, double* acc_J)
{
    int ks = 0, ka = 0;
    for (int i = 0; i < o->na; ++i) {
        const oaux* a = &o->A[i];
        if (a->type == A_SE3) {
            double Xi[12], Xj[12];
            memcpy(Xi, o->P[a->a].R, 72); memcpy(Xi + 9, o->P[a->a].t, 24);
            memcpy(Xj, o->P[a->b].R, 72); memcpy(Xj + 9, o->P[a->b].t, 24);
            orc_se3_edge(Xi, Xj, a->z, se3_err + 6 * ks, se3_Ji + 36 * ks, se3_Jj + 36 * ks);
            ++ks;
        } else if (a->type == A_ACCEL) {
            accel_err(o->P[a->a].R, a->off, a->z, acc_err + 3 * ka);
            accel_jac(o, &o->P[a->a], a, acc_J + 18 * ka);
            ++ka;
        }
    }
}
/* dense H (n x n, full symmetric, row-major) and b at the current estimate; column of each vertex
 * (by insertion index) in pose_col / lm_col (-1 fixed). returns n, or -1 if cap too small */
int64_t orc_ba_dense_system(orc_ba* o, double* H, double* b, int64_t cap, int32_t* pose_col, int32_t* lm_col)
{
    if (!o->ready) return -2;
    if (cap < o->n) return -1;
    double p, r;
    compute_errors(o, &p, &r);
    build_system(o);
    memset(H, 0, 8 * o->n * o->n);
    for (int64_t c = 0; c < o->n; ++c) for (int64_t q = o->Ap[c]; q < o->Ap[c + 1]; ++q) {
        const int64_t rr = o->Ai[q];
        H[rr * o->n + c] = o->Ax[q]; H[c * o->n + rr] = o->Ax[q];
    }
    memcpy(b, o->b, 8 * o->n);
    for (int i = 0; i < o->np; ++i) pose_col[i] = o->P[i].col;
    for (int i = 0; i < o->nl; ++i) lm_col[i] = o->L[i].col;
    return o->n;
}
/* LM trace: 5 doubles per iteration (robust chi before, after, lambda after, trials, continued) */
int orc_ba_trace(const orc_ba* o, double* out, int cap)
{
    const int n = o->ntrace < cap ? o->ntrace : cap;
    if (out) memcpy(out, o->trace, 8 * n);
    return o->ntrace;
}
void orc_ba_trace_clear(orc_ba* o) { o->ntrace = 0; }

/* _applyOptimizationToLandmarks pruning rule (Cg2oOptimizer.cpp:1486-1504) */
int64_t orc_ba_prune_diverged(orc_ba* o)
{
    int64_t removed = 0;
    int* dead = calloc(o->nl + 1, 4);
    for (int i = 0; i < o->nl; ++i) {
        const double* p = o->L[i].p;
        if (!(o->d_sane > p[0] * p[0] + p[1] * p[1] + p[2] * p[2])) { dead[i] = 1; removed++; }
    }
    if (removed) {
        int* remap = malloc(4 * (o->nl + 1));
        int k = 0;
        omap_free(&o->ml); omap_init(&o->ml);
        for (int i = 0; i < o->nl; ++i) { if (dead[i]) remap[i] = -1; else { remap[i] = k; o->L[k] = o->L[i]; omap_put(&o->ml, o->L[k].id, k); k++; } }
        o->nl = k;
        int64_t ke = 0;
        for (int64_t e = 0; e < o->ne; ++e) if (remap[o->E[e].lm] >= 0) { o->E[ke] = o->E[e]; o->E[ke].lm = remap[o->E[ke].lm]; ke++; }
        o->ne = ke;
        int ka = 0;
        for (int a = 0; a < o->na; ++a) {
            if (o->A[a].type == A_LMLM) {
                if (remap[o->A[a].a] < 0 || remap[o->A[a].b] < 0) continue;
                o->A[ka] = o->A[a]; o->A[ka].a = remap[o->A[ka].a]; o->A[ka].b = remap[o->A[ka].b]; ka++;
            } else o->A[ka++] = o->A[a];
        }
        o->na = ka;
        free(remap);
        o->ready = 0;
    }
    free(dead);
    return removed;
}
