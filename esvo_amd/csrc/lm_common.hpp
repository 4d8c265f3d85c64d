// lm_common.hpp — what the two refinement kernels (kernels_lm.hip: 15 x 7 register layouts; kernels_lm_any.hip: any patch
// size) share.
#pragma once
#include "common.hpp"

namespace esvo {

// internal::lmpar2 for n == 1 (Appendix B.1)
__device__ inline double lm_lmpar2(double r, double diag, double qtf, double delta, double& par) {
  const double dwarf = 2.2250738585072014e-308;
  double x = qtf / r;
  double wa2 = diag * x;
  double dxnorm = fabs(wa2);
  double fp = dxnorm - delta;
  if (fp <= 0.1 * delta) { par = 0; return x; }
  double wa1 = diag * wa2 / dxnorm;
  wa1 = wa1 / r;
  double temp = fabs(wa1);
  double parl = fp / delta / temp / temp;
  wa1 = r * qtf / diag;
  const double gn = fabs(wa1);
  double paru = gn / delta;
  if (paru == 0.) paru = dwarf / ((delta < 0.1) ? delta : 0.1);
  par = (par < parl) ? parl : par;   // std::max(par, parl)
  par = (paru < par) ? paru : par;   // std::min(par, paru)
  if (par == 0.) par = gn / dxnorm;
  int it = 0;
  while (true) {
    ++it;
    if (par == 0.) { const double c = 0.001 * paru; par = (dwarf < c) ? c : dwarf; }
    const double ds = sqrt(par) * diag;
    const double sdiag2 = r * r + ds * ds;
    const double sdiag = sqrt(sdiag2);
    x = r * qtf / sdiag2;
    wa2 = diag * x;
    dxnorm = fabs(wa2);
    temp = fp;
    fp = dxnorm - delta;
    if (fabs(fp) <= 0.1 * delta || (parl == 0. && fp <= temp && temp < 0.) || it == 10) break;
    wa1 = diag * (wa2 / dxnorm);
    wa1 = wa1 / sdiag;
    temp = fabs(wa1);
    const double parc = fp / delta / temp / temp;
    if (fp > 0.) parl = (parl < par) ? par : parl;
    if (fp < 0.) paru = (par < paru) ? par : paru;
    const double pc = par + parc;
    par = (parl < pc) ? pc : parl;
  }
  return x;
}

}  // namespace esvo
