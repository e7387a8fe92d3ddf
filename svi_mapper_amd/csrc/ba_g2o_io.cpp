// ba_g2o_io.cpp — .g2o text interchange for the BA handle (svi_ba_load_g2o / svi_ba_save_g2o).
//
// The reference writes its graphs with g2o::SparseOptimizer::save at
// src/optimization/Cg2oOptimizer.cpp:495-497 and :514; the tags below are the ones g2o's slam3d
// types register (upstream; g2o source is not part of the reference tree) plus the reference's own
// EDGE_SE3_LINEAR_ACCELERATION, whose read/write is src/optimization/edge_se3_linear_acceleration.cpp:35-103
// (parameter id, 3 measurement values, upper triangle of the 3x3 information).
//   VERTEX_SE3:QUAT id x y z qx qy qz qw            VERTEX_TRACKXYZ id x y z            FIX id...
//   PARAMS_SE3OFFSET id x y z qx qy qz qw           PARAMS_CAMERACALIB id x y z qx qy qz qw fx fy cx cy
//   EDGE_SE3:QUAT i j x y z qx qy qz qw  + 21       EDGE_SE3_TRACKXYZ p l pid x y z + 6
//   EDGE_PROJECT_DEPTH p l pid u v d + 6            EDGE_PROJECT_DISPARITY p l pid u v d + 6
//   EDGE_POINTXYZ i j x y z + 6                     EDGE_SE3_LINEAR_ACCELERATION p pid ax ay az + 6
// g2o does not serialise robust kernels: on load, landmark edges get the Cauchy kernel the
// reference always attaches (Cg2oOptimizer.cpp:1018,1042,1070,456), pose edges none.
// Parameter ids follow Cg2oOptimizer.h:21-27 (0 world, 1 camera left, 2 camera right, 3 IMU->LEFT).
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "ba_host.h"
#include "ba_math.h"

using namespace svi;

namespace {

void quat_pose(const double* v7, double* T) // x y z qx qy qz qw -> R(9), t(3)
{
    double q[4] = {v7[6], v7[3], v7[4], v7[5]};
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n > 0) for (double& x : q) x /= n;
    quat_to_R(q[0], q[1], q[2], q[3], T);
    T[9] = v7[0]; T[10] = v7[1]; T[11] = v7[2];
}

void pose_quat(const double* T, double* v7)
{
    double q[4];
    R_to_quat(T, q);
    v7[0] = T[9]; v7[1] = T[10]; v7[2] = T[11];
    v7[3] = q[1]; v7[4] = q[2]; v7[5] = q[3]; v7[6] = q[0];
}

} // namespace

extern "C" {

int svi_ba_load_g2o(svi_ba* ba, const char* path)
{
    if (!ba || !path) return fail(SVI_ERR_INVALID, "null argument");
    if (int rc_sync = ensure_host(ba)) return rc_sync;
    std::ifstream in(path);
    if (!in) return fail(SVI_ERR_IO, "cannot open %s", path);
    std::map<int, std::array<double, 12>> offsets;
    std::string line;
    int lineno = 0;
    // edges may precede nothing they reference in files g2o writes (parameters, vertices, edges), but be
    // tolerant: collect edge lines and add them after all vertices
    std::vector<std::string> edge_lines;
    std::vector<int64_t> fixed_ids;
    while (std::getline(in, line)) {
        ++lineno;
        std::istringstream is(line);
        std::string tag;
        if (!(is >> tag) || tag[0] == '#') continue;
        if (tag == "VERTEX_SE3:QUAT") {
            int64_t id; double v[7];
            is >> id;
            for (double& x : v) is >> x;
            if (!is) return fail(SVI_ERR_IO, "%s:%d: malformed VERTEX_SE3:QUAT", path, lineno);
            double T[12];
            quat_pose(v, T);
            if (int rc = svi_ba_add_pose(ba, id, T, 0)) return rc;
        } else if (tag == "VERTEX_TRACKXYZ") {
            int64_t id; double p[3];
            is >> id >> p[0] >> p[1] >> p[2];
            if (!is) return fail(SVI_ERR_IO, "%s:%d: malformed VERTEX_TRACKXYZ", path, lineno);
            if (int rc = svi_ba_add_landmark(ba, id, p, 0)) return rc;
        } else if (tag == "FIX") {
            int64_t id;
            while (is >> id) fixed_ids.push_back(id);
        } else if (tag == "PARAMS_SE3OFFSET") {
            int id; double v[7];
            is >> id;
            for (double& x : v) is >> x;
            if (!is) return fail(SVI_ERR_IO, "%s:%d: malformed PARAMS_SE3OFFSET", path, lineno);
            std::array<double, 12> T;
            quat_pose(v, T.data());
            offsets[id] = T;
            if (id == 3) memcpy(ba->imu_off, T.data(), 96); // eOFFSET_IMUtoLEFT (Cg2oOptimizer.h:26)
        } else if (tag == "PARAMS_CAMERACALIB") {
            int id; double v[11];
            is >> id;
            for (double& x : v) is >> x;
            if (!is) return fail(SVI_ERR_IO, "%s:%d: malformed PARAMS_CAMERACALIB", path, lineno);
            if (id == 1) { ba->opt.fx = v[7]; ba->opt.fy = v[8]; ba->opt.cx = v[9]; ba->opt.cy = v[10]; } // eCAMERA_LEFT
        } else if (tag.rfind("EDGE_", 0) == 0) {
            edge_lines.push_back(line);
        } else {
            return fail(SVI_ERR_IO, "%s:%d: unknown tag %s", path, lineno, tag.c_str());
        }
    }
    for (int64_t id : fixed_ids) {
        auto p = ba->pose_ix.find(id);
        if (p != ba->pose_ix.end()) { ba->poses[p->second].fixed = 1; continue; }
        auto l = ba->lm_ix.find(id);
        if (l != ba->lm_ix.end()) { ba->lms[l->second].fixed = 1; continue; }
        return fail(SVI_ERR_IO, "%s: FIX of unknown vertex %lld", path, (long long)id);
    }
    for (const std::string& el : edge_lines) {
        std::istringstream is(el);
        std::string tag;
        is >> tag;
        int rc = SVI_OK;
        if (tag == "EDGE_SE3:QUAT") {
            int64_t i, j; double v[7], info[21], Z[12];
            is >> i >> j;
            for (double& x : v) is >> x;
            for (double& x : info) is >> x;
            if (!is) return fail(SVI_ERR_IO, "%s: malformed EDGE_SE3:QUAT", path);
            quat_pose(v, Z);
            rc = svi_ba_add_edge_se3(ba, i, j, Z, info, 0);
        } else if (tag == "EDGE_SE3_TRACKXYZ" || tag == "EDGE_PROJECT_DEPTH" || tag == "EDGE_PROJECT_DISPARITY") {
            int64_t p, l; int pid; double z[3], info[6];
            is >> p >> l >> pid >> z[0] >> z[1] >> z[2];
            for (double& x : info) is >> x;
            if (!is) return fail(SVI_ERR_IO, "%s: malformed %s", path, tag.c_str());
            if (tag == "EDGE_SE3_TRACKXYZ") {
                auto off = offsets.find(pid);
                if (off != offsets.end())
                    for (int k = 0; k < 12; ++k)
                        if (std::fabs(off->second[k] - (k % 4 == 0 && k < 9 ? 1.0 : 0.0)) > 1e-12)
                            return fail(SVI_ERR_UNSUPPORTED, "%s: EDGE_SE3_TRACKXYZ with a non-identity offset parameter", path);
                rc = svi_ba_add_edge_xyz(ba, p, l, z, info, 1);
            } else if (tag == "EDGE_PROJECT_DEPTH") rc = svi_ba_add_edge_depth(ba, p, l, z, info, 1);
            else rc = svi_ba_add_edge_disparity(ba, p, l, z, info, 1);
        } else if (tag == "EDGE_POINTXYZ") {
            int64_t i, j; double z[3], info[6];
            is >> i >> j >> z[0] >> z[1] >> z[2];
            for (double& x : info) is >> x;
            if (!is) return fail(SVI_ERR_IO, "%s: malformed EDGE_POINTXYZ", path);
            rc = svi_ba_add_edge_lm_lm(ba, i, j, z, info, 1);
        } else if (tag == "EDGE_SE3_LINEAR_ACCELERATION") {
            int64_t p; int pid; double a[3], info[6];
            is >> p >> pid >> a[0] >> a[1] >> a[2];
            for (double& x : info) is >> x;
            if (!is) return fail(SVI_ERR_IO, "%s: malformed EDGE_SE3_LINEAR_ACCELERATION", path);
            auto off = offsets.find(pid);
            rc = svi_ba_add_edge_accel(ba, p, a, off != offsets.end() ? off->second.data() : nullptr, info);
        } else {
            return fail(SVI_ERR_IO, "%s: unknown edge tag %s", path, tag.c_str());
        }
        if (rc != SVI_OK) return rc;
    }
    return edges_flush(ba);
}

int svi_ba_save_g2o(svi_ba* ba, const char* path)
{
    if (!ba || !path) return fail(SVI_ERR_INVALID, "null argument");
    if (int rc_sync = ensure_host(ba)) return rc_sync;
    FILE* f = std::fopen(path, "w");
    if (!f) return fail(SVI_ERR_IO, "cannot open %s for writing", path);
    const svi_ba_options& o = ba->opt;
    // parameters as the reference registers them (Cg2oOptimizer.cpp:99-118)
    std::fprintf(f, "PARAMS_SE3OFFSET 0 0 0 0 0 0 0 1\n");
    std::fprintf(f, "PARAMS_CAMERACALIB 1 0 0 0 0 0 0 1 %.17g %.17g %.17g %.17g\n", o.fx, o.fy, o.cx, o.cy);
    std::fprintf(f, "PARAMS_CAMERACALIB 2 0 0 0 0 0 0 1 %.17g %.17g %.17g %.17g\n", o.fx, o.fy, o.cx, o.cy);
    // offset parameters: id 0 is the identity ("world"), id 3 the handle's IMU->LEFT offset (the only ones the reference has);
    // a gravity edge added with an offset of its own (svi_ba_add_edge_accel) gets a further parameter, ids 4, 5, ...
    std::vector<std::array<double, 12>> offs;
    std::vector<int> off_id, acc_pid(ba->acc.size(), 3);
    auto same = [](const double* a, const double* b) { for (int k = 0; k < 12; ++k) if (a[k] != b[k]) return false; return true; };
    static const double kIdent[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    for (size_t k = 0; k < ba->acc.size(); ++k) {
        const double* o12 = ba->acc[k].off;
        if (same(o12, ba->imu_off)) continue;
        if (same(o12, kIdent)) { acc_pid[k] = 0; continue; }
        size_t q = 0;
        while (q < offs.size() && !same(o12, offs[q].data())) ++q;
        if (q == offs.size()) { std::array<double, 12> t; memcpy(t.data(), o12, 96); offs.push_back(t); off_id.push_back(4 + (int)q); }
        acc_pid[k] = off_id[q];
    }
    double off7[7];
    pose_quat(ba->imu_off, off7);
    std::fprintf(f, "PARAMS_SE3OFFSET 3 %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", off7[0], off7[1], off7[2], off7[3], off7[4], off7[5], off7[6]);
    for (size_t q = 0; q < offs.size(); ++q) {
        pose_quat(offs[q].data(), off7);
        std::fprintf(f, "PARAMS_SE3OFFSET %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", off_id[q], off7[0], off7[1], off7[2], off7[3], off7[4], off7[5], off7[6]);
    }
    std::vector<int64_t> fixed;
    for (const HLm& l : ba->lms) {
        std::fprintf(f, "VERTEX_TRACKXYZ %lld %.17g %.17g %.17g\n", (long long)l.id, l.p[0], l.p[1], l.p[2]);
        if (l.fixed) fixed.push_back(l.id);
    }
    for (const HPose& p : ba->poses) {
        double v[7];
        pose_quat(p.T, v);
        std::fprintf(f, "VERTEX_SE3:QUAT %lld %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", (long long)p.id, v[0], v[1], v[2], v[3], v[4], v[5], v[6]);
        if (p.fixed) fixed.push_back(p.id);
    }
    for (int64_t id : fixed) std::fprintf(f, "FIX %lld\n", (long long)id);
    for (size_t k = 0; k < ba->acc.size(); ++k) {
        const HAcc& e = ba->acc[k];
        std::fprintf(f, "EDGE_SE3_LINEAR_ACCELERATION %lld %d %.17g %.17g %.17g", (long long)ba->poses[e.pose].id, acc_pid[k], e.a[0], e.a[1], e.a[2]);
        for (double x : e.info) std::fprintf(f, " %.17g", x);
        std::fprintf(f, "\n");
    }
    for (const HSe3& e : ba->se3) {
        double v[7];
        pose_quat(e.Z, v);
        std::fprintf(f, "EDGE_SE3:QUAT %lld %lld", (long long)ba->poses[e.i].id, (long long)ba->poses[e.j].id);
        for (double x : v) std::fprintf(f, " %.17g", x);
        for (double x : e.info) std::fprintf(f, " %.17g", x);
        std::fprintf(f, "\n");
    }
    static const char* kTag[3] = {"EDGE_SE3_TRACKXYZ", "EDGE_PROJECT_DEPTH", "EDGE_PROJECT_DISPARITY"};
    for (size_t i = 0; i < ba->proj.size(); ++i) {
        const EdgeStore& e = ba->proj;
        const double* z = e.z(i);
        std::fprintf(f, "%s %lld %lld %d %.17g %.17g %.17g", kTag[e.type(i)], (long long)ba->poses[e.pose[i]].id, (long long)ba->lms[e.lm[i]].id,
                     e.type(i) == 0 ? 0 : 1, z[0], z[1], z[2]);
        for (int q = 0; q < 6; ++q) std::fprintf(f, " %.17g", e.info(i)[q]);
        std::fprintf(f, "\n");
    }
    for (const HLL& e : ba->lmlm) {
        std::fprintf(f, "EDGE_POINTXYZ %lld %lld %.17g %.17g %.17g", (long long)ba->lms[e.i].id, (long long)ba->lms[e.j].id, e.z[0], e.z[1], e.z[2]);
        for (double x : e.info) std::fprintf(f, " %.17g", x);
        std::fprintf(f, "\n");
    }
    const bool ok = std::fclose(f) == 0;
    return ok ? SVI_OK : fail(SVI_ERR_IO, "write to %s failed", path);
}

} // extern "C"
