// Parameter update rules (lib/QuadAlgorithm.py:454-578).
// Part of the kernel sources collected by cpdp_kernels.h (include that header, not this one).
#pragma once
#include "cpdp_common.h"

namespace lfsd {

// =====================================================================================
//  Parameter update rules (lib/QuadAlgorithm.py:454-578), one thread per (trajectory, parameter)
// =====================================================================================
template <typename T> struct OptArgs {
  int batch, n_param, method, iter_idx;      // iter_idx starts from 0 (QuadAlgorithm.py:507)
  T lr, mu, beta1, beta2, eps;
  T* theta;            // [B][p]  in/out
  const T* grad;       // [B][p]
  T* m;                // [B][p]  Nesterov velocity / first moment
  T* v;                // [B][p]  second moment
  T* vhat;             // [B][p]  AMSGrad running max
  const T* proj_lo;    // [p] lower bound applied after the step (-inf = none); examples clamp theta[0] >= 1e-8
  const int* row_active;   // [B] or nullptr: rows with 0 keep theta AND their optimizer state (a frozen trajectory)
};

template <typename T> __global__ void optimizer_kernel(OptArgs<T> a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)a.batch * a.n_param) return;
  if (a.row_active && !a.row_active[i / a.n_param]) return;
  const int j = (int)(i % a.n_param);
  const T g = a.grad[i];
  T th = a.theta[i];
  const T idx = T(a.iter_idx + 1);
  if (a.method == OPT_VANILLA) {
    th -= a.lr * g;
  } else if (a.method == OPT_NESTEROV) {
    // grad was evaluated at the look-ahead point theta + mu*v (QuadAlgorithm.py:478-486)
    const T vel = a.mu * a.m[i] - a.lr * g;
    a.m[i] = vel;
    th += vel;
  } else {
    const T mm = a.beta1 * a.m[i] + (T(1) - a.beta1) * g;
    const T vv = a.beta2 * a.v[i] + (T(1) - a.beta2) * g * g;
    a.m[i] = mm; a.v[i] = vv;
    if (a.method == OPT_AMSGRAD) {
      const T vh = t_max(a.vhat[i], vv);
      a.vhat[i] = vh;
      th -= a.lr * mm / (t_sqrt(vh) + a.eps);
    } else {
      const T c1 = T(1) - t_pow(a.beta1, idx), c2 = T(1) - t_pow(a.beta2, idx);
      const T mh = mm / c1, vh = vv / c2;
      if (a.method == OPT_ADAM) th -= a.lr * mh / (t_sqrt(vh) + a.eps);
      else th -= a.lr * (a.beta1 * mh + (T(1) - a.beta1) / c1 * g) / (t_sqrt(vh) + a.eps);
    }
  }
  if (a.proj_lo) th = t_max(th, a.proj_lo[j]);
  a.theta[i] = th;
}

// look-ahead point of Nesterov: out = theta + mu * v   (QuadAlgorithm.py:478)
template <typename T> __global__ void lookahead_kernel(long long n, T mu, const T* theta, const T* v, T* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = theta[i] + mu * v[i];
}

}  // namespace lfsd
