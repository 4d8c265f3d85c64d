// oracle/_ref build shim for the mapper NODE (esvo_Mapping.cpp) (TEST INFRASTRUCTURE): enough of the ROS API for the node to be
// constructed and for dataTransferring / MappingAtTime / InitializationAtTime to run; nothing is published, no thread loops.
#ifndef ESVO_REF_SHIM_NODE_KINDR_TF
#define ESVO_REF_SHIM_NODE_KINDR_TF
#include <tf/tf.h>
#include <kindr/minimal/quat-transformation.h>
namespace tf {
inline void transformTFToKindr(const StampedTransform& st, kindr::minimal::QuatTransformation* out) {
  Eigen::Matrix<double, 4, 4> M;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) M(i, j) = st.T[i * 4 + j];
  *out = kindr::minimal::QuatTransformation(M);
}
}
#endif
