// svi_fundamental_matcher.hpp — C++ facade over the tracking half of the C ABI, shaped like the entry points of
// CFundamentalMatcher (src/core/CFundamentalMatcher.h:112-170):
//
//   reference call                                                   facade
//   getPoseStereoPosit( frame, ..., T_estimate, T_last, ..., ms )    planFrame(...) + getPoseStereoPosit(...)
//   trackEpipolar( frame, imgL, imgR, T_w2l, T_l2w, ms, ... )        planFrame(...) + trackEpipolar()
//   trackManual( frame, imgL, imgR, T_w2l, T_l2w, ms, ... )          planFrame(...) + trackManual()
//   addNewLandmarks( imgL, imgR, T_w2l, T_l2w, frame, ... )          addNewLandmarks( key points, sizes, descriptors )
//
// The reference keeps its landmarks as CLandmark objects and walks them one by one; the facade takes them as host
// structure-of-arrays (one row per active landmark), keeps the device copies and hands every result back as host arrays
// in the same row order - what a CFundamentalMatcher built on it would copy into / out of its CLandmark fields.
// Descriptor extraction is the built-in BRIEF (setImages: the rectified frames, the caller's 256 test pairs) unless an
// extractor callback is installed on the handle (svi_tracker_set_extractor); stage 2 needs a detector callback
// (svi_tracker_set_detector - GFTT stays with the caller).
//
// Plain C++17 + the HIP runtime API for the copies (compile with -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include, link
// -lsvi_hot -lamdhip64); no OpenCV / Eigen types in the signatures, so it compiles in this repository's image.
#pragma once
#include <hip/hip_runtime_api.h>

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "svi_hot.h"

namespace svi {

struct FrameLandmarks {                     // host SoA, n rows
    std::vector<double>  xyz_world;         // n x 3   CLandmark::vecPointXYZOptimized
    std::vector<float>   kp_size;           // n       dKeyPointSize
    std::vector<float>   last_disparity;    // n       getLastDisparity()
    std::vector<double>  uv_reference;      // n x 2   vecUVReferenceLEFT
    std::vector<int32_t> dp_index;          // n       the detection point that owns the landmark
    std::vector<uint8_t> last_desc_left;    // n x 32  getLastDescriptorLEFT()
    std::vector<uint8_t> last_desc_right;   // n x 32
    std::vector<uint8_t> ref_desc_left;     // n x 32  matDescriptorReferenceLEFT
    int size() const { return (int)kp_size.size(); }
};

struct TrackOutcome {                       // host SoA, n rows
    std::vector<int32_t> status;            // SVI_TRK_MATCH_*
    std::vector<int8_t>  stage;             // 1 / 2 / 3, 0 none
    std::vector<float>   uv_left, uv_right; // n x 2
    std::vector<double>  xyz_left;          // n x 3
    std::vector<uint8_t> desc_left, desc_right; // n x 32
};

class FundamentalMatcherGPU {
public:
    FundamentalMatcherGPU(const svi_track_camera& cam, int device = 0) : cam_(cam)
    {
        check(svi_matcher_create(device, nullptr, &m_));
        check(svi_tracker_create(m_, &cam_, &t_));
        stream_ = static_cast<hipStream_t>(svi_matcher_stream(m_));
    }
    ~FundamentalMatcherGPU()
    {
        svi_tracker_destroy(t_);
        if (b_) svi_brief_destroy(b_);
        svi_matcher_destroy(m_);
        for (void* p : owned_) (void)hipFree(p);
    }
    FundamentalMatcherGPU(const FundamentalMatcherGPU&) = delete;
    FundamentalMatcherGPU& operator=(const FundamentalMatcherGPU&) = delete;

    // the rectified frames of this time step for the built-in BRIEF extractor; pattern: 256 x (y1, x1, y2, x2) int8
    void setImages(const uint8_t* left, const uint8_t* right, int width, int height, const int8_t* pattern)
    {
        if (!b_) { check(svi_brief_create(m_, pattern, &b_)); check(svi_tracker_set_brief(t_, b_)); }
        const size_t bytes = (size_t)width * height;
        uint8_t* d = upload(img_, left, bytes, 2 * bytes);
        hip(hipMemcpyAsync(d + bytes, right, bytes, hipMemcpyHostToDevice, stream_));
        check(svi_brief_set_image_dev(b_, 0, d, width, height, width));
        check(svi_brief_set_image_dev(b_, 1, d + bytes, width, height, width));
    }

    // projection, FoV gate, search rectangles and epipolar segments of every landmark for the pose T_world_to_left;
    // dp_T: n_dp x 12, CDetectionPoint::matTransformationLEFTtoWORLD
    void planFrame(const double T_world_to_left[12], const std::vector<double>& dp_T, double motion_scaling, const FrameLandmarks& lm)
    {
        n_ = lm.size();
        svi_track_landmarks d{};
        d.n = n_;
        d.xyz_world = upload(buf_[0], lm.xyz_world.data(), sizeof(double) * 3 * n_);
        d.kp_size = upload(buf_[1], lm.kp_size.data(), sizeof(float) * n_);
        d.last_disparity = upload(buf_[2], lm.last_disparity.data(), sizeof(float) * n_);
        d.uv_reference = upload(buf_[3], lm.uv_reference.data(), sizeof(double) * 2 * n_);
        d.dp_index = upload(buf_[4], lm.dp_index.data(), sizeof(int32_t) * n_);
        d.last_desc_left = upload(buf_[5], lm.last_desc_left.data(), (size_t)32 * n_);
        d.last_desc_right = upload(buf_[6], lm.last_desc_right.data(), (size_t)32 * n_);
        d.ref_desc_left = lm.ref_desc_left.empty() ? nullptr : upload(buf_[7], lm.ref_desc_left.data(), (size_t)32 * n_);
        check(svi_tracker_plan(t_, T_world_to_left, dp_T.data(), (int)(dp_T.size() / 12), motion_scaling, &d));
    }

    TrackOutcome trackManual(const std::vector<uint8_t>* active = nullptr) { return run(svi_track_manual, active); }
    TrackOutcome trackEpipolar(const std::vector<uint8_t>* active = nullptr) { return run(svi_track_epipolar, active); }
    TrackOutcome trackStage1(const std::vector<uint8_t>* active = nullptr) { return run(svi_track_stage1, active); }
    TrackOutcome trackStage2(const std::vector<uint8_t>* active = nullptr) { return run(svi_track_stage2, active); }

    // stage 1 -> stage 2 and the frame pose from what they found; throws like the reference (CExceptionPoseOptimization)
    // when the solver fails, *pose_out (optional) receives the solver's report either way
    std::array<double, 12> getPoseStereoPosit(const double T_last[12], const double t_imu[3], const double T_estimate[12], TrackOutcome* found = nullptr,
                                              const std::vector<uint8_t>* active = nullptr, svi_posit_result* pose_out = nullptr)
    {
        svi_posit_params prm;
        svi_posit_params_default(&prm);
        for (int i = 0; i < 12; ++i) { prm.P_left[i] = cam_.P_left[i]; prm.P_right[i] = cam_.P_right[i]; }
        svi_posit_result pose{};
        svi_track_result r = result_buffers(n_);
        const uint8_t* act = active ? upload(buf_[8], active->data(), (size_t)n_) : nullptr;
        check(svi_track_pose_stereo_posit(t_, act, &prm, T_last, t_imu, T_estimate, &r, &pose));
        if (found) *found = download(r, n_);
        if (pose_out) *pose_out = pose;
        if (pose.status != SVI_POSIT_OK) throw std::runtime_error("pose optimization failed (status " + std::to_string(pose.status) + ")");
        std::array<double, 12> T;
        for (int i = 0; i < 12; ++i) T[i] = pose.T_world_to_left[i];
        return T;
    }

    // stereo partner + triangulation of fresh LEFT key points (uv n x 2, sizes n, descriptors n x 32)
    TrackOutcome addNewLandmarks(const std::vector<float>& uv_left, const std::vector<float>& kp_size, const std::vector<uint8_t>& desc_left)
    {
        const int n = (int)kp_size.size();
        svi_track_result r = result_buffers(n);
        const float* uv = upload(buf_[9], uv_left.data(), sizeof(float) * 2 * n);
        const float* ks = upload(buf_[10], kp_size.data(), sizeof(float) * n);
        const uint8_t* ds = upload(buf_[11], desc_left.data(), (size_t)32 * n);
        check(svi_track_add_new_landmarks(t_, uv, ks, ds, n, &r));
        return download(r, n);
    }

    svi_tracker* handle() { return t_; }
    svi_matcher* matcher() { return m_; }

private:
    struct Buf { void* p = nullptr; size_t cap = 0; };
    static void check(int rc)
    {
        if (rc != SVI_OK) throw std::runtime_error(std::string(svi_status_string(rc)) + ": " + svi_last_error());
    }
    static void hip(hipError_t e)
    {
        if (e != hipSuccess) throw std::runtime_error(std::string("HIP: ") + hipGetErrorString(e));
    }
    template <class T> T* reserve(Buf& b, size_t bytes)
    {
        if (bytes > b.cap) {
            hip(hipStreamSynchronize(stream_));
            if (b.p) { for (auto& o : owned_) if (o == b.p) o = nullptr; (void)hipFree(b.p); }
            hip(hipMalloc(&b.p, bytes ? bytes : 16));
            owned_.push_back(b.p);
            b.cap = bytes ? bytes : 16;
        }
        return static_cast<T*>(b.p);
    }
    template <class T> T* upload(Buf& b, const T* host, size_t bytes, size_t room = 0)
    {
        T* d = reserve<T>(b, room > bytes ? room : bytes);
        if (bytes) hip(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, stream_));
        return d;
    }
    svi_track_result result_buffers(int n)
    {
        // one block: status (4) stage (1) uv_left (8) uv_right (8) xyz (24) desc (32 + 32), 16-byte aligned pieces
        const size_t N = ((size_t)(n > 0 ? n : 1) + 15) / 16 * 16;
        uint8_t* base = reserve<uint8_t>(res_, N * (4 + 1 + 8 + 8 + 24 + 32 + 32));
        svi_track_result r;
        r.desc_left = base;
        r.desc_right = base + 32 * N;
        r.xyz_left = reinterpret_cast<double*>(base + 64 * N);
        r.uv_left = reinterpret_cast<float*>(base + 88 * N);
        r.uv_right = reinterpret_cast<float*>(base + 96 * N);
        r.status = reinterpret_cast<int32_t*>(base + 104 * N);
        r.stage = reinterpret_cast<int8_t*>(base + 108 * N);
        return r;
    }
    TrackOutcome download(const svi_track_result& r, int n)
    {
        TrackOutcome o;
        o.status.resize(n); o.stage.resize(n); o.uv_left.resize((size_t)2 * n); o.uv_right.resize((size_t)2 * n);
        o.xyz_left.resize((size_t)3 * n); o.desc_left.resize((size_t)32 * n); o.desc_right.resize((size_t)32 * n);
        if (n > 0) {
            hip(hipMemcpyAsync(o.status.data(), r.status, sizeof(int32_t) * n, hipMemcpyDeviceToHost, stream_));
            hip(hipMemcpyAsync(o.stage.data(), r.stage, (size_t)n, hipMemcpyDeviceToHost, stream_));
            hip(hipMemcpyAsync(o.uv_left.data(), r.uv_left, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, stream_));
            hip(hipMemcpyAsync(o.uv_right.data(), r.uv_right, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, stream_));
            hip(hipMemcpyAsync(o.xyz_left.data(), r.xyz_left, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, stream_));
            hip(hipMemcpyAsync(o.desc_left.data(), r.desc_left, (size_t)32 * n, hipMemcpyDeviceToHost, stream_));
            hip(hipMemcpyAsync(o.desc_right.data(), r.desc_right, (size_t)32 * n, hipMemcpyDeviceToHost, stream_));
        }
        hip(hipStreamSynchronize(stream_));
        return o;
    }
    TrackOutcome run(int (*fn)(svi_tracker*, const uint8_t*, const svi_track_result*), const std::vector<uint8_t>* active)
    {
        svi_track_result r = result_buffers(n_);
        const uint8_t* act = active ? upload(buf_[8], active->data(), (size_t)n_) : nullptr;
        check(fn(t_, act, &r));
        return download(r, n_);
    }

    svi_track_camera cam_;
    svi_matcher* m_ = nullptr;
    svi_tracker* t_ = nullptr;
    svi_brief* b_ = nullptr;
    hipStream_t stream_ = nullptr;
    int n_ = 0;
    Buf buf_[12], img_, res_;
    std::vector<void*> owned_;
};

} // namespace svi
