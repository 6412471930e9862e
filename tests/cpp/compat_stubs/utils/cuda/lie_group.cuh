// stand-in declarations of the reference's utils/cuda/lie_group.cuh:8-45 (type check only)
#pragma once
#include <Eigen/Dense>
template <typename T>
class SE3 {
 public:
  SE3() {}
  SE3(const Eigen::Quaternion<T>& rot, const Eigen::Matrix<T, 3, 1>& trans) : R_(rot), t_(trans) {}
  static SE3<T> Identity() { return SE3<T>(Eigen::Quaternion<T>::Identity(), Eigen::Matrix<T, 3, 1>::Zero()); }
  Eigen::Quaternion<T> GetR() const { return R_; }
  Eigen::Matrix<T, 3, 1> GetT() const { return t_; }

 private:
  Eigen::Quaternion<T> R_;
  Eigen::Matrix<T, 3, 1> t_;
};
