// stand-in for OpenCV core: the members of cv::Mat the binding touches (type check only)
#pragma once
#include <cstdint>
#define CV_8UC3 16
#define CV_32FC1 5
namespace cv {
class Mat {
 public:
  Mat() = default;
  Mat(int rows_, int cols_, int type, void* data_) : data((unsigned char*)data_), rows(rows_), cols(cols_), type_(type) {}
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  bool isContinuous() const { return true; }
  int type() const { return type_; }
  unsigned char* data = nullptr;
  int rows = 0, cols = 0;

 private:
  int type_ = 0;
};
}  // namespace cv
