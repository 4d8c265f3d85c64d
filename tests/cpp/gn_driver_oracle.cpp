// esvo_hip::gauss_newton_register (include/esvo_hip.hpp) driven by the CPU oracle's normal equations: the tracker's host-side
// optimiser exercised without a GPU.  usage: gn_driver_oracle in.bin out.bin
//   in : i32 W, H | f64 P[12] | u8 ts[H*W] | u64 n | f32 xyz[n*3] | f64 T_world_ref[16] | f64 R0[9] | f64 t0[3] | i32 iters
//        [| i32 batch]   batch > 0: the batch advances with the iteration as in esvo_hip::RegProblemLM::solve (BATCH_SIZE)
//   out: f64 R[9] | f64 t[3] | f64 rms | i32 iterations
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "esvo_hip.hpp"
#include "../../oracle/esvo_oracle.h"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t W, H, iters, batch = 0;
  esvo_calib_t cal = {};
  size_t ok = fread(&W, 4, 1, f) + fread(&H, 4, 1, f) + fread(cal.P, 8, 12, f);
  cal.width = W; cal.height = H;
  std::vector<uint8_t> ts((size_t)W * H);
  ok += fread(ts.data(), 1, ts.size(), f);
  uint64_t n;
  ok += fread(&n, 8, 1, f);
  std::vector<float> xyz(3 * n);
  ok += fread(xyz.data(), 4, xyz.size(), f);
  double Tref[16], R0[9], t0[3];
  ok += fread(Tref, 8, 16, f) + fread(R0, 8, 9, f) + fread(t0, 8, 3, f) + fread(&iters, 4, 1, f);
  if (fread(&batch, 4, 1, f) != 1) batch = 0;
  fclose(f);
  // the oracle's camera needs the calibration arrays only for block matching: the tracker reads P and the (absent) mask
  std::vector<float> lut((size_t)W * H * 2, 0.f), mx((size_t)W * H, 0.f);
  cal.rect_lut = lut.data(); cal.map_x = mx.data(); cal.map_y = mx.data(); cal.rect_mask = nullptr;
  orc_tracker_handle trk = orc_tracker_create(&cal);
  orc_tracker_set_current(trk, ts.data(), 5);
  orc_tracker_set_reference(trk, xyz.data(), n, Tref);
  // the driver's speculative trials (np poses per call) evaluated one after the other: the oracle has no launch to share
  const bool batches = batch > 0 && (uint64_t)batch < n;
  const size_t n_batches = batches ? (n / (size_t)batch > 1 ? n / (size_t)batch : 1) : 1;
  auto ne = [&](int it, int np, const double* R, const double* t, double* Hm, double* b, double* cost, size_t* m) {
    const size_t off = batches ? ((size_t)it % n_batches) * (size_t)batch : 0, cnt = batches ? (size_t)batch : n;
    for (int q = 0; q < np; ++q) {
      double v[28];
      *m = orc_tracker_normal_equations(trk, R + 9 * q, t + 3 * q, off, cnt, 1, 50.0, v);
      int k = 0;
      for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) { Hm[36 * q + i * 6 + j] = Hm[36 * q + j * 6 + i] = v[k]; ++k; }
      for (int i = 0; i < 6; ++i) b[6 * q + i] = v[21 + i];
      cost[q] = v[27];
    }
    return true;
  };
  const esvo_hip::Registration g = esvo_hip::gauss_newton_register(ne, R0, t0, iters, 1e-3, !batches);
  orc_tracker_destroy(trk);
  f = fopen(argv[2], "wb");
  fwrite(g.R, 8, 9, f); fwrite(g.t, 8, 3, f); fwrite(&g.rms, 8, 1, f);
  int32_t it = g.iterations;
  fwrite(&it, 4, 1, f);
  fclose(f);
  return g.ok ? 0 : 1;
}
