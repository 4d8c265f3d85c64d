// oracle/_ref build shim for the mapper NODE (esvo_Mapping.cpp) (TEST INFRASTRUCTURE): enough of the ROS API for the node to be
// constructed and for dataTransferring / MappingAtTime / InitializationAtTime to run; nothing is published, no thread loops.
// the struct catkin generates from cfg/DVS_MappingStereo.cfg (field names and types from that file)
#ifndef ESVO_REF_SHIM_NODE_CFG
#define ESVO_REF_SHIM_NODE_CFG
namespace esvo_core {
struct DVS_MappingStereoConfig {
  double EM_TIME_THRESHOLD = 0.0001, EM_EPIPOLAR_THRESHOLD = 0.5, EM_TS_NCC_THRESHOLD = 0.1;
  int EM_NUM_EVENT_MATCHING = 30000, EM_PATCH_INTENSITY_THRESHOLD = 125;
  double EM_PATCH_VALID_RATIO = 0.1;
  int BM_MAX_NUM_EVENTS_PER_MATCHING = 400, BM_min_disparity = 0, BM_max_disparity = 40, BM_step = 2;
  double BM_ZNCC_Threshold = 0.1;
  double invDepth_min_range = 0.16, invDepth_max_range = 2.0, residual_vis_threshold = 12, stdVar_vis_threshold = 0.12;
  int age_max_range = 5, age_vis_threshold = 1, fusion_radius = 0, maxNumFusionFrames = 0, maxNumFusionPoints = 5000;
  int PROCESS_EVENT_NUM = 100, TS_HISTORY_LENGTH = 100, mapping_rate_hz = 20;
  bool Denoising = false, Regularization = false, ResetButton = false;
};
}
#endif
