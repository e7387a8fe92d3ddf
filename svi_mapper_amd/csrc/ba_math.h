// ba_math.h — per-edge and per-vertex arithmetic of the BA path (host + device).
//
// g2o slam3d semantics as restated in SURVEY.md Appendix B:
//   VertexSE3 estimate X = (R, t) LEFT->WORLD, increment d = (dt, dq_xyz): X <- X * (dt, quat(sqrt(1-|dq|^2), dq))
//       (same map as CMiniVisionToolbox::getTransformationFromVector, src/vision/CMiniVisionToolbox.cpp:354-377)
//   EdgeSE3PointXYZ / ...Depth / ...Disparity with identity offsets (factories: Cg2oOptimizer.cpp:999-1073)
//   EdgeSE3 (odometry, Cg2oOptimizer.cpp:1248-1266), RobustKernelCauchy.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace svi {

#define SVI_HD __host__ __device__ __forceinline__

// RobustKernelCauchy::robustify: rho0 = d^2 ln(1 + e2/d^2), rho1 = 1/(1 + e2/d^2)
SVI_HD void cauchy(double delta, double e2, double& rho0, double& rho1)
{
    const double dsqr = delta * delta, aux = e2 / dsqr + 1.0;
    rho0 = dsqr * log(aux);
    rho1 = 1.0 / aux;
}

// point in camera Z = R'(p - t)
SVI_HD void to_camera(const double* R, const double* t, const double* p, double* Z)
{
    const double d0 = p[0] - t[0], d1 = p[1] - t[1], d2 = p[2] - t[2];
    Z[0] = R[0] * d0 + R[3] * d1 + R[6] * d2;
    Z[1] = R[1] * d0 + R[4] * d1 + R[7] * d2;
    Z[2] = R[2] * d0 + R[5] * d1 + R[8] * d2;
}

// error only
SVI_HD void proj_error(int type, const double* R, const double* t, const double* p, const double* z, double fx,
                       double fy, double cx, double cy, double* e)
{
    double Z[3];
    to_camera(R, t, p, Z);
    if (type == 0) { e[0] = Z[0] - z[0]; e[1] = Z[1] - z[1]; e[2] = Z[2] - z[2]; return; }
    const double px = fx * Z[0] + cx * Z[2], py = fy * Z[1] + cy * Z[2], pz = Z[2];
    e[0] = px / pz - z[0];
    e[1] = py / pz - z[1];
    e[2] = (type == 1 ? pz : 1.0 / pz) - z[2];
}

// error and J (3 x 9 row-major) = d e / d (dt, dq, dp)
SVI_HD void proj_eval(int type, const double* R, const double* t, const double* p, const double* z, double fx,
                      double fy, double cx, double cy, double* e, double* J)
{
    double Z[3];
    to_camera(R, t, p, Z);
    // J0 = [ -I | 2[Z]x | R' ]
    double J0[27];
    J0[0] = -1.0; J0[1] = 0.0;  J0[2] = 0.0;  J0[3] = 0.0;          J0[4] = -2.0 * Z[2];  J0[5] = 2.0 * Z[1];
    J0[9] = 0.0;  J0[10] = -1.0; J0[11] = 0.0; J0[12] = 2.0 * Z[2];  J0[13] = 0.0;         J0[14] = -2.0 * Z[0];
    J0[18] = 0.0; J0[19] = 0.0; J0[20] = -1.0; J0[21] = -2.0 * Z[1]; J0[22] = 2.0 * Z[0];  J0[23] = 0.0;
    J0[6] = R[0];  J0[7] = R[3];  J0[8] = R[6];
    J0[15] = R[1]; J0[16] = R[4]; J0[17] = R[7];
    J0[24] = R[2]; J0[25] = R[5]; J0[26] = R[8];
    if (type == 0) {
        e[0] = Z[0] - z[0]; e[1] = Z[1] - z[1]; e[2] = Z[2] - z[2];
#pragma unroll
        for (int k = 0; k < 27; ++k) J[k] = J0[k];
        return;
    }
    const double px = fx * Z[0] + cx * Z[2], py = fy * Z[1] + cy * Z[2], pz = Z[2];
    e[0] = px / pz - z[0];
    e[1] = py / pz - z[1];
    e[2] = (type == 1 ? pz : 1.0 / pz) - z[2];
    const double iz2 = 1.0 / (pz * pz);
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const double a0 = fx * J0[c] + cx * J0[18 + c];
        const double a1 = fy * J0[9 + c] + cy * J0[18 + c];
        const double a2 = J0[18 + c];
        J[c]      = iz2 * (a0 * pz - px * a2);
        J[9 + c]  = iz2 * (a1 * pz - py * a2);
        J[18 + c] = (type == 1) ? a2 : -iz2 * a2;
    }
}

// Structured form of the three projection edges (EdgeSE3PointXYZ / ...Depth / ...Disparity):
//     J_pose = A [ -I | 2[Z]x ],   J_lm = A R'      with  Z = R'(p - t)  and  A = d e / d Z  (3x3)
// A is the identity for XYZ and has 5 non-zeros for the two projective kinds (rows 0,1: perspective
// division of K Z, row 2: depth p_z or inverse depth 1/p_z).  Everything downstream (H blocks, Schur
// products, back-substitution) is expressed through  C = A' (rho1 Omega) A  (symmetric 3x3),
// u = A' (rho1 Omega) e,  N = C R'  and Z:
//     H_pl = [ -N ; -2[Z]x N ]          H_ll = R N            b_l = -R u
//     H_pp = [ C , -C K ; K C , -K C K ] with K = 2[Z]x       b_p = [ u ; K u ]
// A is returned as (a00, a02, a11, a12, a22): a01 = a10 = a20 = a21 = 0.
SVI_HD void proj_core(int type, const double* R, const double* t, const double* p, const double* z, double fx,
                      double fy, double cx, double cy, double* e, double* Z, double* A5)
{
    to_camera(R, t, p, Z);
    if (type == 0) {
        e[0] = Z[0] - z[0]; e[1] = Z[1] - z[1]; e[2] = Z[2] - z[2];
        A5[0] = 1.0; A5[1] = 0.0; A5[2] = 1.0; A5[3] = 0.0; A5[4] = 1.0;
        return;
    }
    const double pz = Z[2], iz = 1.0 / pz, iz2 = iz * iz;
    const double px = fx * Z[0] + cx * pz, py = fy * Z[1] + cy * pz;
    e[0] = px * iz - z[0];
    e[1] = py * iz - z[1];
    e[2] = (type == 1 ? pz : iz) - z[2];
    A5[0] = fx * iz;  A5[1] = -fx * Z[0] * iz2;
    A5[2] = fy * iz;  A5[3] = -fy * Z[1] * iz2;
    A5[4] = (type == 1) ? 1.0 : -iz2;
}

// C = A' O A (upper: c00 c01 c02 c11 c12 c22) and u = A' O e for O symmetric (upper o00 o01 o02 o11 o12 o22)
SVI_HD void proj_cu(const double* A5, const double* O, const double* e, double* C, double* u)
{
    const double a00 = A5[0], a02 = A5[1], a11 = A5[2], a12 = A5[3], a22 = A5[4];
    // OA = O A : columns of A are (a00,0,0), (0,a11,0), (a02,a12,a22)
    const double oa00 = O[0] * a00, oa10 = O[1] * a00, oa20 = O[2] * a00;
    const double oa01 = O[1] * a11, oa11 = O[3] * a11, oa21 = O[4] * a11;
    const double oa02 = O[0] * a02 + O[1] * a12 + O[2] * a22;
    const double oa12 = O[1] * a02 + O[3] * a12 + O[4] * a22;
    const double oa22 = O[2] * a02 + O[4] * a12 + O[5] * a22;
    C[0] = a00 * oa00;                            // c00
    C[1] = a00 * oa01;                            // c01
    C[2] = a00 * oa02;                            // c02
    C[3] = a11 * oa11;                            // c11
    C[4] = a11 * oa12;                            // c12
    C[5] = a02 * oa02 + a12 * oa12 + a22 * oa22;  // c22
    (void)oa10; (void)oa20; (void)oa21;
    const double oe0 = O[0] * e[0] + O[1] * e[1] + O[2] * e[2];
    const double oe1 = O[1] * e[0] + O[3] * e[1] + O[4] * e[2];
    const double oe2 = O[2] * e[0] + O[4] * e[1] + O[5] * e[2];
    u[0] = a00 * oe0;
    u[1] = a11 * oe1;
    u[2] = a02 * oe0 + a12 * oe1 + a22 * oe2;
}

SVI_HD void quat_to_R(double w, double x, double y, double z, double* R)
{
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// rotation matrix -> unit quaternion (w,x,y,z) with w >= 0
SVI_HD void R_to_quat(const double* m, double* q)
{
    double t = m[0] + m[4] + m[8];
    double w, v[3];
    if (t > 0) {
        t = sqrt(t + 1.0); w = 0.5 * t; t = 0.5 / t;
        v[0] = (m[7] - m[5]) * t; v[1] = (m[2] - m[6]) * t; v[2] = (m[3] - m[1]) * t;
    } else {
        // (i, j, k) = the largest diagonal entry and its cyclic successors.  Written out per case: indexing m with i, j, k puts
        // the matrix - a register array of the caller - into scratch memory on the GPU (every access a memory round trip: the
        // pose-only edge kernels spent most of their 11 us there)
        const double m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3], m4 = m[4], m5 = m[5], m6 = m[6], m7 = m[7], m8 = m[8]; // (values, not addresses)
        int i = 0;
        if (m4 > m0) i = 1;
        if (m8 > (i == 1 ? m4 : m0)) i = 2;
        const int j = (i + 1) % 3;
        const double mii = i == 0 ? m0 : i == 1 ? m4 : m8;
        const double mjj = i == 0 ? m4 : i == 1 ? m8 : m0;
        const double mkk = i == 0 ? m8 : i == 1 ? m0 : m4;
        t = sqrt(mii - mjj - mkk + 1.0);
        double vi = 0.5 * t; t = 0.5 / t;
        const double d0 = m7 - m5, d1 = m2 - m6, d2 = m3 - m1; // m[3k+j] - m[3j+k] for i = 0, 1, 2
        const double s0 = m3 + m1, s1 = m7 + m5, s2 = m2 + m6; // m[3j+i] + m[3i+j]
        const double u0 = m6 + m2, u1 = m1 + m3, u2 = m5 + m7; // m[3k+i] + m[3i+k]
        w = (i == 0 ? d0 : i == 1 ? d1 : d2) * t;
        double vj = (i == 0 ? s0 : i == 1 ? s1 : s2) * t;
        double vk = (i == 0 ? u0 : i == 1 ? u1 : u2) * t;
        v[0] = (i == 0) ? vi : ((j == 0) ? vj : vk);
        v[1] = (i == 1) ? vi : ((j == 1) ? vj : vk);
        v[2] = (i == 2) ? vi : ((j == 2) ? vj : vk);
    }
    const double nrm = sqrt(w * w + v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double s = 1.0 / nrm;
    if (w < 0) s = -s;
    q[0] = w * s; q[1] = v[0] * s; q[2] = v[1] * s; q[3] = v[2] * s;
}

SVI_HD void mat3_mul(const double* A, const double* B, double* C)
{
    double r[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
#pragma unroll
    for (int k = 0; k < 9; ++k) C[k] = r[k];
}
SVI_HD void mat3T_mul(const double* A, const double* B, double* C)
{
    double r[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
#pragma unroll
    for (int k = 0; k < 9; ++k) C[k] = r[k];
}

// g2o VertexSE3::oplusImpl on T = (R row-major 9, t 3)
SVI_HD void pose_oplus(const double* T, const double* d, double* Tn)
{
    double dR[9];
    const double w2 = 1.0 - (d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    if (w2 < 0) { for (int k = 0; k < 9; ++k) dR[k] = 0.0; dR[0] = dR[4] = dR[8] = 1.0; }
    else quat_to_R(sqrt(w2), d[3], d[4], d[5], dR);
    const double t0 = T[9] + (T[0] * d[0] + T[1] * d[1] + T[2] * d[2]);
    const double t1 = T[10] + (T[3] * d[0] + T[4] * d[1] + T[5] * d[2]);
    const double t2 = T[11] + (T[6] * d[0] + T[7] * d[1] + T[8] * d[2]);
    mat3_mul(T, dR, Tn);
    Tn[9] = t0; Tn[10] = t1; Tn[11] = t2;
}

// EdgeSE3: e = toVectorMQT(Z^-1 Xi^-1 Xj) (6); Ji, Jj 6x6 row-major exact derivatives w.r.t. the
// increments of Xi, Xj (Ji == nullptr: error only)
SVI_HD void se3_edge_eval(const double* Xi, const double* Xj, const double* Z, double* e, double* Ji, double* Jj)
{
    const double *Ri = Xi, *ti = Xi + 9, *Rj = Xj, *tj = Xj + 9, *Rz = Z, *tz = Z + 9;
    double Rb[9], tb[3];
    mat3T_mul(Ri, Rj, Rb);
    {
        const double d0 = tj[0] - ti[0], d1 = tj[1] - ti[1], d2 = tj[2] - ti[2];
        tb[0] = Ri[0] * d0 + Ri[3] * d1 + Ri[6] * d2;
        tb[1] = Ri[1] * d0 + Ri[4] * d1 + Ri[7] * d2;
        tb[2] = Ri[2] * d0 + Ri[5] * d1 + Ri[8] * d2;
    }
    double Ra[9], ta[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Ra[3 * i + j] = Rz[3 * j + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) ta[i] = -(Ra[3 * i] * tz[0] + Ra[3 * i + 1] * tz[1] + Ra[3 * i + 2] * tz[2]);
    double Re[9], te[3];
    mat3_mul(Ra, Rb, Re);
#pragma unroll
    for (int i = 0; i < 3; ++i) te[i] = Ra[3 * i] * tb[0] + Ra[3 * i + 1] * tb[1] + Ra[3 * i + 2] * tb[2] + ta[i];
    double qe[4];
    R_to_quat(Re, qe);
    e[0] = te[0]; e[1] = te[1]; e[2] = te[2]; e[3] = qe[1]; e[4] = qe[2]; e[5] = qe[3];
    if (!Ji) return;
    double qa[4], qb[4];
    R_to_quat(Ra, qa);
    R_to_quat(Rb, qb);
    const double wq = qa[0] * qb[0] - (qa[1] * qb[1] + qa[2] * qb[2] + qa[3] * qb[3]);
    const double vq0 = qa[0] * qb[1] + qb[0] * qa[1] + (qa[2] * qb[3] - qa[3] * qb[2]);
    const double vq1 = qa[0] * qb[2] + qb[0] * qa[2] + (qa[3] * qb[1] - qa[1] * qb[3]);
    const double vq2 = qa[0] * qb[3] + qb[0] * qa[3] + (qa[1] * qb[2] - qa[2] * qb[1]);
    const double sgn = (wq * qe[0] + vq0 * qe[1] + vq1 * qe[2] + vq2 * qe[3]) < 0 ? -1.0 : 1.0;
#pragma unroll
    for (int k = 0; k < 36; ++k) { Ji[k] = 0.0; Jj[k] = 0.0; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) { Jj[6 * r + c] = Re[3 * r + c]; Ji[6 * r + c] = -Ra[3 * r + c]; }
    {
        const double we = qe[0], *ve = qe + 1;
        const double M[9] = {we, -ve[2], ve[1], ve[2], we, -ve[0], -ve[1], ve[0], we};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Jj[6 * (3 + r) + 3 + c] = M[3 * r + c];
    }
    {
        const double S[9] = {0, -tb[2], tb[1], tb[2], 0, -tb[0], -tb[1], tb[0], 0};
        double RS[9];
        mat3_mul(Ra, S, RS);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Ji[6 * r + 3 + c] = 2.0 * RS[3 * r + c];
        const double wa = qa[0], *va = qa + 1, wb = qb[0], *vb = qb + 1;
        const double Ma[9] = {wa, -va[2], va[1], va[2], wa, -va[0], -va[1], va[0], wa};
        const double Mb[9] = {wb, vb[2], -vb[1], -vb[2], wb, vb[0], vb[1], -vb[0], wb};
        double MM[9];
        mat3_mul(Mb, Ma, MM);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Ji[6 * (3 + r) + 3 + c] = sgn * (vb[r] * va[c] - MM[3 * r + c]);
    }
}

} // namespace svi
