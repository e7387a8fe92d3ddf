// svi_cv_matcher.hpp — cv::DescriptorMatcher plug-in over the C ABI (include/svi_hot.h).
//
// Drop-in for the matcher the reference creates at src/core/CTriangulator.cpp:12
//     m_pMatcher( std::make_shared< cv::BFMatcher >( cv::NORM_HAMMING ) )
// and shares with CFundamentalMatcher (src/core/CFundamentalMatcher.cpp:20): replace that expression by
//     m_pMatcher( std::make_shared< svi::HammingMatcherGPU >( ) )
// The class follows the plug-in pattern the reference already uses for its own matcher,
// cv::CBTreeMatcher (src/vision/CBTreeMatcher.h:13-155): same overridden virtuals, k = 1 only.
//
// The reference calls the 2-Mat overload m_pMatcher->match( query, pool, matches ) at all 11 sites; OpenCV implements it as
// clone( true ) -> add( pool ) -> match( query ): clone() therefore SHARES the GPU handle (stream, device buffers) with the
// matcher it was cloned from - no GPU resource is created or destroyed per call - and keeps the device it was built for.
//
// Header-only; OpenCV is not in the build image of this repository: tests/test_adapter_headers.py compiles it against
// minimal stand-in declarations of the OpenCV names it uses (tests/stubs/opencv2: a syntax / override check, it pins
// nothing); see INTEGRATION.md.
#pragma once
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>

#include <memory>
#include <stdexcept>
#include <vector>

#include "svi_hot.h"

namespace svi {

class HammingMatcherGPU : public cv::DescriptorMatcher {
public:
    explicit HammingMatcherGPU(int device = 0) : device_(device)
    {
        svi_matcher* m = nullptr;
        if (svi_matcher_create(device, nullptr, &m) != SVI_OK) throw std::runtime_error(svi_last_error());
        m_ = std::shared_ptr<svi_matcher>(m, [](svi_matcher* p) { svi_matcher_destroy(p); });
    }
    ~HammingMatcherGPU() override = default;

    bool isMaskSupported() const override { return false; }
    // shares the handle; the train collection is copied unless emptyTrainData (cv::BFMatcher::clone does the same)
    cv::Ptr<cv::DescriptorMatcher> clone(bool emptyTrainData = false) const override
    {
        cv::Ptr<HammingMatcherGPU> c(new HammingMatcherGPU(*this));
        if (emptyTrainData) c->clear();
        return c;
    }

protected:
    // called by cv::DescriptorMatcher::match(query, train, matches) with k = 1 after add(train)
    void knnMatchImpl(cv::InputArray queryDescriptors, std::vector<std::vector<cv::DMatch>>& matches, int k,
                      cv::InputArrayOfArrays /*masks*/ = cv::noArray(), bool /*compactResult*/ = false) override
    {
        if (k != 1) throw cv::Exception(0, "only k = 1 is implemented", "knnMatchImpl", __FILE__, __LINE__);
        const cv::Mat q = queryDescriptors.getMat();
        if (q.type() != CV_8U || q.cols != 32) throw cv::Exception(0, "BRIEF-256 (N x 32 CV_8U) expected", "knnMatchImpl", __FILE__, __LINE__);
        matches.assign(static_cast<size_t>(q.rows), std::vector<cv::DMatch>());
        const cv::Mat qc = q.isContinuous() ? q : q.clone();
        std::vector<int32_t> idx(static_cast<size_t>(q.rows)), dist(static_cast<size_t>(q.rows));
        for (size_t img = 0; img < trainDescCollection.size(); ++img) {
            const cv::Mat& t = trainDescCollection[img];
            if (t.empty()) continue;
            const cv::Mat tc = t.isContinuous() ? t : t.clone();
            // 16-byte alignment of cv::Mat data is guaranteed by OpenCV's allocator
            if (svi_match_hamming256(m_.get(), qc.ptr<uint8_t>(), q.rows, tc.ptr<uint8_t>(), t.rows, nullptr, 257, idx.data(), dist.data()) != SVI_OK)
                throw cv::Exception(0, svi_last_error(), "knnMatchImpl", __FILE__, __LINE__);
            for (int i = 0; i < q.rows; ++i) {
                if (idx[i] < 0) continue;
                std::vector<cv::DMatch>& m = matches[static_cast<size_t>(i)];
                const cv::DMatch cand(i, idx[i], static_cast<int>(img), static_cast<float>(dist[i]));
                if (m.empty()) m.push_back(cand);
                else if (cand.distance < m[0].distance) m[0] = cand; // strict '<': first image wins ties, like BFMatcher
            }
        }
    }
    void radiusMatchImpl(cv::InputArray, std::vector<std::vector<cv::DMatch>>&, float, cv::InputArrayOfArrays = cv::noArray(),
                         bool = false) override
    {
        throw cv::Exception(0, "radiusMatchImpl not implemented", "radiusMatchImpl", __FILE__, __LINE__); // as CBTreeMatcher.h:141-146
    }

private:
    HammingMatcherGPU(const HammingMatcherGPU&) = default;   // for clone(): same handle, same device, copied train collection
    std::shared_ptr<svi_matcher> m_;
    int device_ = 0;
};

} // namespace svi
