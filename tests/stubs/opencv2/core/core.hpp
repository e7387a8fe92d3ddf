// Minimal stand-in declarations of the OpenCV 3.x names include/svi_cv_matcher.hpp uses - NOT OpenCV, no behaviour: they
// exist so that the adapter header goes through a compiler in an image without OpenCV (tests/test_adapter_headers.py:
// syntax, override signatures, const-correctness).  Nothing here pins results.
#pragma once
#include <memory>
#include <string>
#include <vector>
#define CV_8U 0
namespace cv {
template <class T> using Ptr = std::shared_ptr<T>;
template <class T, class... A> Ptr<T> makePtr(A&&... a) { return std::make_shared<T>(std::forward<A>(a)...); }
class Mat {
public:
    int rows = 0, cols = 0;
    int type() const { return CV_8U; }
    bool isContinuous() const { return true; }
    bool empty() const { return rows == 0; }
    Mat clone() const { return *this; }
    template <class T> const T* ptr(int = 0) const { return nullptr; }
};
class _InputArray {
public:
    _InputArray() {}
    _InputArray(const Mat&) {}
    Mat getMat(int = -1) const { return Mat(); }
};
typedef const _InputArray& InputArray;
typedef InputArray InputArrayOfArrays;
inline InputArray noArray() { static _InputArray a; return a; }
class Exception : public std::exception {
public:
    Exception(int, const std::string& e, const std::string&, const std::string&, int) : msg(e) {}
    const char* what() const noexcept override { return msg.c_str(); }
    std::string msg;
};
struct DMatch {
    DMatch() {}
    DMatch(int q, int t, int i, float d) : queryIdx(q), trainIdx(t), imgIdx(i), distance(d) {}
    int queryIdx = -1, trainIdx = -1, imgIdx = -1;
    float distance = 0.f;
};
} // namespace cv
