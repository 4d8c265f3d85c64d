// replay_golden.cpp — drives libesvo_hip.so through the C++ host layer (include/esvo_hip.hpp) with the
// call sequence of esvo_Mapping::MappingAtTime (esvo_core/src/esvo_Mapping.cpp:261-431):
//   ebm_.createMatchProblem / match_all_HyperThread -> dpSolver_.solve / pointCulling ->
//   dqvDepthPoints_.push_back + dFusor_.update loop -> clean -> regularisation
// on a fixture dumped by tests/test_cpp_host.py, and writes every stage's output for comparison
// with the golden vectors; at the end the tracker's evaluation side (RegProblemLM) runs once on the local map.   usage: replay_golden <fixture.bin> <out.bin>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

#include "esvo_hip.hpp"

using namespace esvo_hip;

template <typename T>
static void rd(std::ifstream& f, T* p, size_t n) { f.read(reinterpret_cast<char*>(p), sizeof(T) * n); }
template <typename T>
static void wr(std::ofstream& f, const T* p, size_t n) { f.write(reinterpret_cast<const char*>(p), sizeof(T) * n); }

struct CalibBuf {
  esvo_calib_t c;
  std::vector<float> lut, mx, my;
  std::vector<uint8_t> mask;
  void read(std::ifstream& f, int W, int H) {
    const size_t n = (size_t)W * H;
    c.width = W; c.height = H;
    rd(f, c.P, 12);
    lut.resize(2 * n); mask.resize(n); mx.resize(n); my.resize(n);
    rd(f, lut.data(), 2 * n); rd(f, mask.data(), n); rd(f, mx.data(), n); rd(f, my.data(), n);
    c.rect_lut = lut.data(); c.rect_mask = mask.data(); c.map_x = mx.data(); c.map_y = my.data();
  }
};

int main(int argc, char** argv) {
  if (argc != 3) { std::fprintf(stderr, "usage: %s fixture.bin out.bin\n", argv[0]); return 2; }
  try {
    std::ifstream in(argv[1], std::ios::binary);
    std::ofstream out(argv[2], std::ios::binary);
    int32_t hdr[3];
    rd(in, hdr, 3);
    const int W = hdr[0], H = hdr[1], n_ticks = hdr[2];
    esvo_params_t prm;
    rd(in, &prm, 1);
    CalibBuf cl, cr;
    cl.read(in, W, H);
    cr.read(in, W, H);
    auto ctx = std::make_shared<Context>(prm, cl.c, cr.c, 0);
    EventBM ebm(ctx);
    DepthProblemSolver dpSolver(ctx);
    DepthFusion dFusor(ctx);
    const double cost_vis_threshold = prm.residual_vis_threshold * prm.residual_vis_threshold * (prm.patch_size_x * prm.patch_size_y);
    std::vector<uint8_t> last_tsL;
    double last_T[16] = {0};
    for (int k = 0; k < n_ticks; ++k) {
      StampedTimeSurfaceObs TS_obs;
      rd(in, &TS_obs.t_ns, 1);
      rd(in, TS_obs.T_world_cam, 16);
      uint64_t m = 0, n_ev = 0;
      rd(in, &m, 1);
      StampTransformationMap st_map;
      st_map.stamps_ns.resize(m); st_map.T_world_virtual.resize(16 * m);
      rd(in, st_map.stamps_ns.data(), m); rd(in, st_map.T_world_virtual.data(), 16 * m);
      rd(in, &n_ev, 1);
      std::vector<Event> vEvents(n_ev);
      rd(in, vEvents.data(), n_ev);
      std::vector<uint8_t> tsL((size_t)W * H), tsR((size_t)W * H);
      rd(in, tsL.data(), tsL.size()); rd(in, tsR.data(), tsR.size());
      TS_obs.TS_left = tsL.data(); TS_obs.TS_right = tsR.data();

      std::vector<EventMatchPair> vEMP;
      ebm.createMatchProblem(&TS_obs, &st_map, &vEvents);           // esvo_Mapping.cpp:308
      ebm.match_all_HyperThread(vEMP);                              // :309
      std::vector<DepthPoint> vdp;
      dpSolver.solve(&vEMP, &TS_obs, vdp);                          // :330
      DepthProblemSolver::pointCulling(vdp, prm.stdvar_vis_threshold, cost_vis_threshold, prm.invdepth_min, prm.invdepth_max);  // :334
      dFusor.pushFrame(vdp, st_map);                                // :346 / :360 + window policy
      const uint64_t numFusionCount = dFusor.update();              // :372-395
      std::vector<DepthPoint> map;
      dFusor.getDepthMap(map);

      uint64_t n;
      n = vEMP.size(); wr(out, &n, 1); wr(out, vEMP.data(), vEMP.size());
      n = vdp.size();  wr(out, &n, 1); wr(out, vdp.data(), vdp.size());
      wr(out, &numFusionCount, 1);
      n = map.size();  wr(out, &n, 1); wr(out, map.data(), map.size());
      last_tsL = tsL;
      std::memcpy(last_T, TS_obs.T_world_cam, sizeof(last_T));
    }
    {  // the tracker's evaluation side on the local map: RegProblemLM::setProblem / operator() / df (RegProblemLM.cpp)
      std::vector<float> xyz;
      dFusor.getPointCloud(xyz);
      RegProblemConfig cfg;  // cfg/tracking/*.yaml: kernelSize 5, Huber 50, batches of 300
      RegProblemLM reg(ctx, cfg);
      reg.setProblem(xyz.data(), xyz.size() / 3, last_T, last_tsL.data());
      reg.setStochasticSampling(0, cfg.BATCH_SIZE);
      const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, zero3[3] = {0, 0, 0};
      std::vector<double> fvec, fjac;
      reg(I4, fvec);
      reg.df(I3, zero3, fjac);
      uint64_t n = xyz.size(); wr(out, &n, 1); wr(out, xyz.data(), xyz.size());
      n = fvec.size(); wr(out, &n, 1); wr(out, fvec.data(), fvec.size());
      n = fjac.size(); wr(out, &n, 1); wr(out, fjac.data(), fjac.size());
    }
    // error behaviour: unsupported configuration is rejected, not approximated
    try {
      ebm.resetParameters(65, 25, 1, 40, 1, 0.1, false);  // (25 x 25 -- the reference's code default -- is supported since round 4)
      std::fprintf(stderr, "expected ESVO_ERR_UNSUPPORTED for a 65x25 patch\n");
      return 1;
    } catch (const Error& e) {
      if (e.code != ESVO_ERR_UNSUPPORTED) { std::fprintf(stderr, "unexpected error code %d\n", e.code); return 1; }
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "replay_golden: %s\n", e.what());
    return 1;
  }
  return 0;
}
