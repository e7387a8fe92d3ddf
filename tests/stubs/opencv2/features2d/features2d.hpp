// Stand-in for cv::DescriptorMatcher (OpenCV 3.x features2d.hpp: the virtuals a plug-in overrides and the train
// collection it inherits) - see tests/stubs/opencv2/core/core.hpp.
#pragma once
#include "opencv2/core/core.hpp"
namespace cv {
class DescriptorMatcher {
public:
    virtual ~DescriptorMatcher() {}
    virtual void add(InputArrayOfArrays) {}
    virtual void clear() { trainDescCollection.clear(); }
    virtual bool empty() const { return trainDescCollection.empty(); }
    virtual bool isMaskSupported() const = 0;
    virtual void train() {}
    virtual Ptr<DescriptorMatcher> clone(bool emptyTrainData = false) const = 0;
    // the 2-Mat overload the reference calls: clone( true ) -> add -> match, as in OpenCV
    void match(InputArray query, InputArray trainDescriptors, std::vector<DMatch>& matches, InputArray = noArray()) const
    {
        Ptr<DescriptorMatcher> tmp = clone(true);
        tmp->add(trainDescriptors);
        std::vector<std::vector<DMatch>> knn;
        tmp->knnMatchImpl(query, knn, 1);
        matches.clear();
        for (auto& v : knn) for (auto& m : v) matches.push_back(m);
    }
protected:
    virtual void knnMatchImpl(InputArray queryDescriptors, std::vector<std::vector<DMatch>>& matches, int k, InputArrayOfArrays masks = noArray(),
                              bool compactResult = false) = 0;
    virtual void radiusMatchImpl(InputArray queryDescriptors, std::vector<std::vector<DMatch>>& matches, float maxDistance,
                                 InputArrayOfArrays masks = noArray(), bool compactResult = false) = 0;
    std::vector<Mat> trainDescCollection;
};
} // namespace cv
