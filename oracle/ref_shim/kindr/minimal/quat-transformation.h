// oracle/_ref build shim (TEST INFRASTRUCTURE): a rigid transformation with the three members of
// kindr::minimal::QuatTransformation that the mapper sources call.  kindr keeps (quaternion, position);
// this stand-in keeps the 4x4 matrix it was built from, so getTransformationMatrix() returns the input
// exactly, and inverse() is the rigid inverse [R^T | -R^T t].
#ifndef ESVO_REF_SHIM_KINDR
#define ESVO_REF_SHIM_KINDR
#include <glog/logging.h>  // kindr/minimal pulls glog in; DepthProblem.cpp & co. use LOG() through it
#include <Eigen/Eigen>
#include <cmath>
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
  // esvo_MVStereo.cpp only (the marker offset of getPoseAt, publishKFPose): the matrix type and the (position, quaternion) view
  typedef Eigen::Matrix<double, 4, 4> TransformationMatrix;
  struct Quat {
    double q[4];
    double w() const { return q[0]; }
    double x() const { return q[1]; }
    double y() const { return q[2]; }
    double z() const { return q[3]; }
  };
  Eigen::Matrix<double, 3, 1> getPosition() const { return Eigen::Matrix<double, 3, 1>(T_(0, 3), T_(1, 3), T_(2, 3)); }
  Quat getRotation() const {  // Shepperd's branch on the trace; only ever published, never fed back
    Quat r;
    const double tr = T_(0, 0) + T_(1, 1) + T_(2, 2);
    if (tr > 0) {
      const double s = 2.0 * std::sqrt(tr + 1.0);
      r.q[0] = 0.25 * s; r.q[1] = (T_(2, 1) - T_(1, 2)) / s; r.q[2] = (T_(0, 2) - T_(2, 0)) / s; r.q[3] = (T_(1, 0) - T_(0, 1)) / s;
    } else {
      int i = T_(1, 1) > T_(0, 0) ? 1 : 0;
      if (T_(2, 2) > T_(i, i)) i = 2;
      const int j = (i + 1) % 3, k = (i + 2) % 3;
      const double s = 2.0 * std::sqrt(1.0 + T_(i, i) - T_(j, j) - T_(k, k));
      r.q[1 + i] = 0.25 * s; r.q[0] = (T_(k, j) - T_(j, k)) / s; r.q[1 + j] = (T_(j, i) + T_(i, j)) / s; r.q[1 + k] = (T_(k, i) + T_(i, k)) / s;
    }
    return r;
  }
 private:
  Eigen::Matrix<double, 4, 4> T_;
};
}  // namespace minimal
}  // namespace kindr
#endif
