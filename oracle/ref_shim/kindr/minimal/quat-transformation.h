// oracle/_ref build shim (TEST INFRASTRUCTURE): a rigid transformation with the three members of
// kindr::minimal::QuatTransformation that the mapper sources call.  kindr keeps (quaternion, position);
// this stand-in keeps the 4x4 matrix it was built from, so getTransformationMatrix() returns the input
// exactly, and inverse() is the rigid inverse [R^T | -R^T t].
#ifndef ESVO_REF_SHIM_KINDR
#define ESVO_REF_SHIM_KINDR
#include <glog/logging.h>  // kindr/minimal pulls glog in; DepthProblem.cpp & co. use LOG() through it
#include <Eigen/Eigen>
namespace kindr {
namespace minimal {
class QuatTransformation {
 public:
  QuatTransformation() { T_.setIdentity(); }
  explicit QuatTransformation(const Eigen::Matrix<double, 4, 4>& T) : T_(T) {}
  void setIdentity() { T_.setIdentity(); }
  Eigen::Matrix<double, 4, 4> getTransformationMatrix() const { return T_; }
  QuatTransformation inverse() const {
    Eigen::Matrix<double, 4, 4> I;
    I.setIdentity();
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) I(i, j) = T_(j, i);
    for (int i = 0; i < 3; ++i) I(i, 3) = -((I(i, 0) * T_(0, 3) + I(i, 1) * T_(1, 3)) + I(i, 2) * T_(2, 3));
    return QuatTransformation(I);
  }
 private:
  Eigen::Matrix<double, 4, 4> T_;
};
}  // namespace minimal
}  // namespace kindr
#endif
