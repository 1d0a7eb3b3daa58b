// Closed-loop step around the QP solve, scalar per-instance code that compiles for gfx950 (K3
// kernels in mpmpc_hip.hip) and for the host (tests/emul).  Replaces, for B cars at once, what
// src/simulation.py:134-140 does per step on the host:
//   localise   SpatialBicycleModel.get_current_waypoint   src/spatial_bicycle_models.py:256-279
//              SpatialBicycleModel.t2s                    src/spatial_bicycle_models.py:183-219
//   advance    MPC.get_control's use of the solution / the infeasibility fallback  src/MPC.py:185-220
//              SpatialBicycleModel.drive                  src/spatial_bicycle_models.py:221-244
#pragma once
#include <cmath>

#ifndef MPMPC_HD
#define MPMPC_HD inline
#endif

namespace mpmpc {

constexpr double RO_PI = 3.141592653589793;

// Closest of the two waypoints enclosing arc length s (ties to the earlier one).  cum = cumulative
// segment lengths, cum[0] = 0.  Returns -1 when s is past the end of the path (the reference's loop
// guard `car.s < reference_path.length` stops before that).
MPMPC_HD int ro_current_waypoint(const double* cum, int n_wp, double s) {
  int lo = 0, hi = n_wp;                 // first index with cum > s
  while (lo < hi) {
    int mid = (lo + hi) / 2;
    if (cum[mid] > s) hi = mid; else lo = mid + 1;
  }
  const int nxt = lo;
  if (nxt >= n_wp) return -1;
  const int prv = nxt - 1;
  const double c_prev = prv >= 0 ? cum[prv] : cum[n_wp - 1];     // numpy's negative index wraps
  const bool take_next = std::fabs(s - cum[nxt]) < std::fabs(s - c_prev);
  return take_next ? nxt : (prv >= 0 ? prv : n_wp - 1);
}

MPMPC_HD void ro_t2s(double px, double py, double ppsi, double wx, double wy, double wpsi, double* x0) {
  x0[0] = std::cos(wpsi) * (py - wy) - std::sin(wpsi) * (px - wx);
  double t = std::fmod(ppsi - wpsi + RO_PI, 2.0 * RO_PI);       // np.mod: result has the divisor's sign
  if (t < 0.0) t += 2.0 * RO_PI;
  x0[1] = t - RO_PI;
  x0[2] = 0.0;
}

// One car, after the solve.  cc [2N] is MPC.current_control (updated in place), z the primal
// solution, x0 / kappa_wp the pre-step spatial state and the curvature of the current waypoint.
// Split in two so that the device can give every plan entry its own thread (ro_plan_entry for k = 0..N-1, then
// ro_drive by the thread that wrote entry 0); ro_advance is the same thing for one thread per car.
MPMPC_HD bool ro_usable(int status) { return status == 1 || status == 2 || status == -2; }
MPMPC_HD void ro_plan_entry(int N, double L, const double* z, double* cc, int k) {
  const double* uu = z + 3 * (N + 1);
  cc[2 * k] = uu[2 * k];
  cc[2 * k + 1] = std::atan(uu[2 * k + 1] * L);
}
// Returns false when the run ends (N-1 consecutive infeasible steps: the reference calls exit(1)).
MPMPC_HD bool ro_drive(int N, double L, double Ts, int status, const double* cc, int* counter, const double* x0,
                       double kappa_wp, double* pose, double* s, double* u_out) {
  double v, delta;
  if (ro_usable(status)) {
    v = cc[0];
    delta = cc[1];
    *counter = 0;
  } else {
    const int i = 2 * (*counter + 1);
    v = cc[i];
    delta = cc[i + 1];
    *counter += 1;
  }
  u_out[0] = v;
  u_out[1] = delta;
  if (*counter == N - 1) return false;
  const double psi = pose[2];
  pose[0] += (v * std::cos(psi)) * Ts;
  pose[1] += (v * std::sin(psi)) * Ts;
  pose[2] += (v / L * std::tan(delta)) * Ts;
  const double s_dot = 1.0 / (1.0 - x0[0] * kappa_wp) * v * std::cos(x0[1]);
  *s += s_dot * Ts;
  return true;
}
MPMPC_HD bool ro_advance(int N, double L, double Ts, int status, const double* z, double* cc, int* counter,
                         const double* x0, double kappa_wp, double* pose, double* s, double* u_out) {
  if (ro_usable(status))
    for (int k = 0; k < N; ++k) ro_plan_entry(N, L, z, cc, k);
  return ro_drive(N, L, Ts, status, cc, counter, x0, kappa_wp, pose, s, u_out);
}

}  // namespace mpmpc
