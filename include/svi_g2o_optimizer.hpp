// svi_g2o_optimizer.hpp — C++ facade over the BA half of the C ABI, shaped like the calls
// Cg2oOptimizer makes on its g2o::SparseOptimizer (src/optimization/Cg2oOptimizer.cpp).
//
//   reference call (file:line)                                   facade
//   m_cOptimizerSparse.addVertex(VertexSE3)          :41-54,1236  addPose(id, T, fixed)
//   m_cOptimizerSparse.addVertex(VertexPointXYZ)     :1146-1151   addLandmark(id, p)
//   _setAndgetPose(...) + _getEdgeLinearAcceleration :1229-1290   addKeyFrame(id, from, T, shift, accel)
//   _setLandmarkMeasurementsWORLD(...)               :1383-1466   addMeasurements(pose, n, ids, uvL, uvR, xyz)
//   _optimizeUnLimited(m_cOptimizerSparse)           :954-980     optimizeUnLimited()
//   _applyOptimizationToLandmarks / ...ToKeyFrames   :1468-1540   landmark(id) / pose(id) / pruneDiverged()
//   m_cOptimizerSparse.save(".g2o")                  :497,514     save(path)
//
// Plain C++17, no Eigen/g2o types in the signatures (poses as 12 doubles: R row-major, t), so it
// compiles in this repository's image; a Cg2oOptimizer built against Eigen maps
// Eigen::Isometry3d <-> double[12] with two memcpy-style helpers.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>

#include "svi_hot.h"

namespace svi {

class BundleAdjusterGPU {
public:
    using Pose = std::array<double, 12>;
    using Point = std::array<double, 3>;

    explicit BundleAdjusterGPU(double fx, double fy, double cx, double cy, double baseline_m, int device = 0)
    {
        svi_ba_options o;
        svi_ba_options_default(&o);
        o.fx = fx; o.fy = fy; o.cx = cx; o.cy = cy; o.baseline_m = baseline_m; o.device = device;
        check(svi_ba_create(&o, &h_));
    }
    ~BundleAdjusterGPU() { svi_ba_destroy(h_); }
    BundleAdjusterGPU(const BundleAdjusterGPU&) = delete;
    BundleAdjusterGPU& operator=(const BundleAdjusterGPU&) = delete;

    void addPose(int64_t id, const Pose& T, bool fixed = false) { check(svi_ba_add_pose(h_, id, T.data(), fixed)); }
    void addLandmark(int64_t id, const Point& p, bool fixed = false) { check(svi_ba_add_landmark(h_, id, p.data(), fixed)); }
    void addKeyFrame(int64_t id, int64_t from, const Pose& T, const Point* shift = nullptr, const Point* accel = nullptr)
    {
        check(svi_ba_add_keyframe(h_, id, from, T.data(), shift ? shift->data() : nullptr, accel ? accel->data() : nullptr));
    }
    // returns {xyz, uv-depth, uv-disparity} edges stored, like the counters of Cg2oOptimizer.cpp:498-501
    std::array<int64_t, 3> addMeasurements(int64_t pose, int64_t n, const int64_t* lm_ids, const float* uvL, const float* uvR,
                                           const double* xyzLEFT)
    {
        std::array<int64_t, 3> stored{};
        check(svi_ba_add_measurements(h_, pose, n, lm_ids, uvL, uvR, xyzLEFT, stored.data()));
        return stored;
    }
    // Cg2oOptimizer::_optimizeUnLimited: returns the reference's nominal iteration counter (1 + 10 k)
    uint64_t optimizeUnLimited(uint64_t* executed = nullptr)
    {
        uint64_t nominal = 0;
        check(svi_ba_initialize(h_));
        check(svi_ba_optimize_until(h_, 0.99, 1, 10, &nominal, executed));
        return nominal;
    }
    Pose pose(int64_t id) const { Pose T; check(svi_ba_get_pose(h_, id, T.data())); return T; }
    Point landmark(int64_t id) const { Point p; check(svi_ba_get_landmark(h_, id, p.data())); return p; }
    int64_t pruneDiverged() { int64_t n = 0; check(svi_ba_prune_diverged(h_, &n)); return n; }
    double chi2() const { double c = 0; check(svi_ba_chi2(h_, &c, nullptr)); return c; }
    void save(const std::string& path) const { check(svi_ba_save_g2o(h_, path.c_str())); }
    void load(const std::string& path) { check(svi_ba_load_g2o(h_, path.c_str())); }
    svi_ba* handle() { return h_; }

private:
    static void check(int rc)
    {
        if (rc != SVI_OK) throw std::runtime_error(std::string(svi_status_string(rc)) + ": " + svi_last_error());
    }
    svi_ba* h_ = nullptr;
};

} // namespace svi
