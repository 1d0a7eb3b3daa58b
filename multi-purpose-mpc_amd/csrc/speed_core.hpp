// Speed profile of a reference path (K4): per-path code written against an execution policy - one wavefront per
// path on gfx950 (SpWave in mpmpc_hip.hip: lane-strided loops, shuffle reductions, cyclic reduction for the
// tridiagonal systems), one thread per path for very long paths and on the host (SpSerial; tests/emul).
//
// Replaces ReferencePath.compute_speed_profile (src/reference_path.py:289-354), the reference's second
// OSQP call site:
//     min 1/2 |v|^2 - vmax' v      s.t.  a_min <= (v[i+1] - v[i]) / (2 l_i) <= a_max,   v_min <= v[i] <= vmax_i
// with vmax_i = min(v_max, sqrt(ay_max / (|kappa_i| + eps)))  (src/reference_path.py:326-329), n = n_wp - 1.
// The reference takes whatever OSQP returns at eps = 1e-3; here the problem is solved to a KKT point:
// Mehrotra predictor-corrector on the normal equations - a scalar tridiagonal system, because the
// difference rows couple neighbours only - followed by primal-dual active-set rounds and a KKT
// certificate, the same construction as the polish of K2 (mpmpc_core.hpp).
//
// Workspace: SP_ARRAYS arrays of n doubles per path, element (a, i) of path p at w[(a * n + i) * stride + p]
// (stride = number of paths: consecutive threads touch consecutive addresses).
#pragma once
#include <cmath>

#ifndef MPMPC_HD
#define MPMPC_HD inline
#endif

namespace mpmpc {

enum {
  SP_X, SP_Q, SP_HI, SP_C, SP_SL1, SP_SU1, SP_ZL1, SP_ZU1, SP_SL2, SP_SU2, SP_ZL2, SP_ZU2, SP_RD, SP_MD, SP_ME, SP_DX,
  SP_PL1, SP_PU1, SP_PL2, SP_PU2, SP_RHS, SP_A1, SP_A2, SP_L1, SP_L2, SP_ARRAYS
};

struct SpWork {
  double* w;
  long n, stride;
  MPMPC_HD double& operator()(int a, int i) const { return w[((long)a * n + i) * stride]; }
};

struct SpLimits { double a_min, a_max, v_min, v_max, ay_max; };

constexpr int SP_SOLVED = 1, SP_INACCURATE = 2, SP_BAD_INPUT = -1;

// one row of the interior-point step: slack / multiplier directions from the row's A dx
struct SpRow { double dsl, dsu, dzl, dzu; };
MPMPC_HD SpRow sp_row(double sl, double su, double zl, double zu, double rl, double ru, double rcl, double rcu, double adx) {
  SpRow r;
  r.dsl = adx + rl;
  r.dsu = -adx + ru;
  r.dzl = (-rcl - zl * r.dsl) / sl;
  r.dzu = (-rcu - zu * r.dsu) / su;
  return r;
}
MPMPC_HD double sp_block(double blk, double s, double ds) { return ds < 0.0 ? std::fmax(blk, -ds / s) : blk; }

// Solve M y = rhs in place (rhs in SP_RHS -> solution in SP_DX) with the LDL' factors in SP_MD / SP_ME.
MPMPC_HD void sp_tri_solve(const SpWork& W, int n) {
  W(SP_DX, 0) = W(SP_RHS, 0);
  for (int i = 1; i < n; ++i) W(SP_DX, i) = W(SP_RHS, i) - W(SP_ME, i - 1) * W(SP_DX, i - 1);
  W(SP_DX, n - 1) = W(SP_DX, n - 1) / W(SP_MD, n - 1);
  for (int i = n - 2; i >= 0; --i) W(SP_DX, i) = W(SP_DX, i) / W(SP_MD, i) - W(SP_ME, i) * W(SP_DX, i + 1);
}
// LDL' of the tridiagonal (diag SP_MD, sub-diagonal SP_ME[i] between i and i+1), in place: ME <- l_{i+1}
MPMPC_HD void sp_tri_factor(const SpWork& W, int n) {
  for (int i = 0; i + 1 < n; ++i) {
    const double l = W(SP_ME, i) / W(SP_MD, i);
    W(SP_MD, i + 1) -= l * W(SP_ME, i);
    W(SP_ME, i) = l;
  }
}

// Execution policy of sp_solve_t.  SpSerial: one thread walks the path (host emulation, and the thread-per-path
// kernel for very long paths).  The device's SpWave (mpmpc_hip.hip) spreads the elementwise loops over the 64 lanes
// of a wavefront, reduces with shuffles and solves the tridiagonal systems by parallel cyclic reduction.
struct SpSerial {
  MPMPC_HD int first() const { return 0; }
  MPMPC_HD int step() const { return 1; }
  MPMPC_HD void sync() const {}
  MPMPC_HD double rmax(double x) const { return x; }
  MPMPC_HD double rsum(double x) const { return x; }
  MPMPC_HD bool any(bool b) const { return b; }
  MPMPC_HD void tri_factor(const SpWork& W, int n) const { sp_tri_factor(W, n); }
  MPMPC_HD void tri_solve(const SpWork& W, int n) const { sp_tri_solve(W, n); }
};

// v[n] <- optimum; returns SP_SOLVED (KKT certificate <= cert_tol), SP_INACCURATE or SP_BAD_INPUT.
// li[i] = |wp[i+1] - wp[i]|, kappa[i], i = 0..n-1; iters (optional) <- interior-point iterations.
// Every loop over the path is elementwise (it reads neighbours only from arrays the loop does not write), so a
// policy may run it strided over several lanes; P.sync() separates the loops.
template <class P>
MPMPC_HD int sp_solve_t(const P& pol, int n, const double* li, const double* kappa, long in_stride, const SpLimits& lim,
                        double eps, const SpWork& W, double* v, long v_stride, int* iters) {
  const int i0 = pol.first(), di = pol.step();
  const double reg = 1e-9, tol = 1e-10, delta = 1e-9, cert_tol = 1e-8;
  if (n < 2 || !(lim.a_min < lim.a_max)) return SP_BAD_INPUT;
  // ---- data
  bool bad_input = false;
  for (int i = i0; i < n; i += di) {
    double hi = lim.v_max;
    const double cap = std::sqrt(lim.ay_max / (std::fabs(kappa[i * in_stride]) + eps));
    if (cap < hi) hi = cap;
    if (!(lim.v_min < hi)) bad_input = true;
    W(SP_HI, i) = hi;
    W(SP_Q, i) = -hi;
    if (i + 1 < n) W(SP_C, i) = 1.0 / (2.0 * li[i * in_stride]);
  }
  if (pol.any(bad_input)) return SP_BAD_INPUT;       // (any() also orders the loop above before the next one)
  const int m1 = n - 1;
  const double nb = 2.0 * (m1 + n);
  // ---- start: x = 0, unit multipliers, slacks at least 1 (as the host solver this replaces)
  for (int i = i0; i < n; i += di) {
    W(SP_X, i) = 0.0;
    W(SP_SL2, i) = std::fmax(0.0 - lim.v_min, 1.0); W(SP_SU2, i) = std::fmax(W(SP_HI, i) - 0.0, 1.0);
    W(SP_ZL2, i) = 1.0; W(SP_ZU2, i) = 1.0;
    if (i < m1) {
      W(SP_SL1, i) = std::fmax(0.0 - lim.a_min, 1.0); W(SP_SU1, i) = std::fmax(lim.a_max - 0.0, 1.0);
      W(SP_ZL1, i) = 1.0; W(SP_ZU1, i) = 1.0;
    }
  }
  pol.sync();
  auto ax1 = [&](int a, int k) { return W(SP_C, k) * (W(a, k + 1) - W(a, k)); };
  int it = 0;
  bool converged = false;
  for (; it < 60; ++it) {
    // ---- residuals, complementarity, normal-equations matrix
    double res = 0.0, musum = 0.0;
    for (int i = i0; i < n; i += di) {
      double rd = W(SP_X, i) + W(SP_Q, i) + (W(SP_ZU2, i) - W(SP_ZL2, i));
      double md = 1.0 + reg + W(SP_ZL2, i) / W(SP_SL2, i) + W(SP_ZU2, i) / W(SP_SU2, i);
      if (i > 0) {
        const double c = W(SP_C, i - 1), w1 = W(SP_ZL1, i - 1) / W(SP_SL1, i - 1) + W(SP_ZU1, i - 1) / W(SP_SU1, i - 1);
        rd += c * (W(SP_ZU1, i - 1) - W(SP_ZL1, i - 1));
        md += w1 * c * c;
      }
      if (i < m1) {
        const double c = W(SP_C, i), w1 = W(SP_ZL1, i) / W(SP_SL1, i) + W(SP_ZU1, i) / W(SP_SU1, i);
        rd -= c * (W(SP_ZU1, i) - W(SP_ZL1, i));
        md += w1 * c * c;
        W(SP_ME, i) = -w1 * c * c;
        const double a = ax1(SP_X, i);
        res = std::fmax(res, std::fmax(std::fabs(a - lim.a_min - W(SP_SL1, i)), std::fabs(lim.a_max - a - W(SP_SU1, i))));
        musum += W(SP_SL1, i) * W(SP_ZL1, i) + W(SP_SU1, i) * W(SP_ZU1, i);
      }
      W(SP_RD, i) = rd;
      W(SP_MD, i) = md;
      res = std::fmax(res, std::fabs(rd));
      res = std::fmax(res, std::fmax(std::fabs(W(SP_X, i) - lim.v_min - W(SP_SL2, i)), std::fabs(W(SP_HI, i) - W(SP_X, i) - W(SP_SU2, i))));
      musum += W(SP_SL2, i) * W(SP_ZL2, i) + W(SP_SU2, i) * W(SP_ZU2, i);
    }
    pol.sync();
    res = pol.rmax(res);
    const double mu = pol.rsum(musum) / nb;
    if (res < tol && mu < tol) { converged = true; break; }
    if (!(res < 1e300)) break;
    pol.tri_factor(W, n);
    double sigmu = 0.0, alpha = 1.0;
    for (int pass = 0; pass < 2; ++pass) {
      // ---- right-hand side  -rd - A' t,  t = (rcl + zl rl) / sl - (rcu + zu ru) / su  per row
      auto t1 = [&](int k) {
        const double a = ax1(SP_X, k), sl = W(SP_SL1, k), su = W(SP_SU1, k), zl = W(SP_ZL1, k), zu = W(SP_ZU1, k);
        const double rcl = sl * zl - sigmu + (pass ? W(SP_PL1, k) : 0.0), rcu = su * zu - sigmu + (pass ? W(SP_PU1, k) : 0.0);
        return (rcl + zl * (a - lim.a_min - sl)) / sl - (rcu + zu * (lim.a_max - a - su)) / su;
      };
      for (int i = i0; i < n; i += di) {
        const double sl = W(SP_SL2, i), su = W(SP_SU2, i), zl = W(SP_ZL2, i), zu = W(SP_ZU2, i), x = W(SP_X, i);
        const double rcl = sl * zl - sigmu + (pass ? W(SP_PL2, i) : 0.0), rcu = su * zu - sigmu + (pass ? W(SP_PU2, i) : 0.0);
        double r = -W(SP_RD, i) - ((rcl + zl * (x - lim.v_min - sl)) / sl - (rcu + zu * (W(SP_HI, i) - x - su)) / su);
        if (i > 0) r -= W(SP_C, i - 1) * t1(i - 1);
        if (i < m1) r += W(SP_C, i) * t1(i);
        W(SP_RHS, i) = r;
      }
      pol.sync();
      pol.tri_solve(W, n);
      // ---- row directions: fraction to the boundary; the predictor also leaves the second-order terms
      double blk = 0.0;
      for (int i = i0; i < n; i += di) {
        {
          const double sl = W(SP_SL2, i), su = W(SP_SU2, i), zl = W(SP_ZL2, i), zu = W(SP_ZU2, i), x = W(SP_X, i);
          const double rcl = sl * zl - sigmu + (pass ? W(SP_PL2, i) : 0.0), rcu = su * zu - sigmu + (pass ? W(SP_PU2, i) : 0.0);
          const SpRow d = sp_row(sl, su, zl, zu, x - lim.v_min - sl, W(SP_HI, i) - x - su, rcl, rcu, W(SP_DX, i));
          blk = sp_block(sp_block(sp_block(sp_block(blk, sl, d.dsl), su, d.dsu), zl, d.dzl), zu, d.dzu);
          if (!pass) { W(SP_PL2, i) = d.dsl * d.dzl; W(SP_PU2, i) = d.dsu * d.dzu; }
        }
        if (i < m1) {
          const double sl = W(SP_SL1, i), su = W(SP_SU1, i), zl = W(SP_ZL1, i), zu = W(SP_ZU1, i), a = ax1(SP_X, i);
          const double rcl = sl * zl - sigmu + (pass ? W(SP_PL1, i) : 0.0), rcu = su * zu - sigmu + (pass ? W(SP_PU1, i) : 0.0);
          const SpRow d = sp_row(sl, su, zl, zu, a - lim.a_min - sl, lim.a_max - a - su, rcl, rcu, ax1(SP_DX, i));
          blk = sp_block(sp_block(sp_block(sp_block(blk, sl, d.dsl), su, d.dsu), zl, d.dzl), zu, d.dzu);
          if (!pass) { W(SP_PL1, i) = d.dsl * d.dzl; W(SP_PU1, i) = d.dsu * d.dzu; }
        }
      }
      pol.sync();
      blk = pol.rmax(blk);
      const double ratio = blk > 0.0 ? 1.0 / blk : 1e300;
      alpha = pass ? std::fmin(1.0, 0.995 * ratio) : std::fmin(1.0, ratio);
      // ---- predictor: centring from the affine complementarity.  corrector: take the step.
      double aff = 0.0;
      for (int i = i0; i < n; i += di) {
        {
          const double sl = W(SP_SL2, i), su = W(SP_SU2, i), zl = W(SP_ZL2, i), zu = W(SP_ZU2, i), x = W(SP_X, i);
          const double rcl = sl * zl - sigmu + (pass ? W(SP_PL2, i) : 0.0), rcu = su * zu - sigmu + (pass ? W(SP_PU2, i) : 0.0);
          const SpRow d = sp_row(sl, su, zl, zu, x - lim.v_min - sl, W(SP_HI, i) - x - su, rcl, rcu, W(SP_DX, i));
          if (!pass) aff += (sl + alpha * d.dsl) * (zl + alpha * d.dzl) + (su + alpha * d.dsu) * (zu + alpha * d.dzu);
          else { W(SP_SL2, i) = sl + alpha * d.dsl; W(SP_SU2, i) = su + alpha * d.dsu; W(SP_ZL2, i) = zl + alpha * d.dzl; W(SP_ZU2, i) = zu + alpha * d.dzu; }
        }
        if (i < m1) {
          const double sl = W(SP_SL1, i), su = W(SP_SU1, i), zl = W(SP_ZL1, i), zu = W(SP_ZU1, i), a = ax1(SP_X, i);
          const double rcl = sl * zl - sigmu + (pass ? W(SP_PL1, i) : 0.0), rcu = su * zu - sigmu + (pass ? W(SP_PU1, i) : 0.0);
          const SpRow d = sp_row(sl, su, zl, zu, a - lim.a_min - sl, lim.a_max - a - su, rcl, rcu, ax1(SP_DX, i));
          if (!pass) aff += (sl + alpha * d.dsl) * (zl + alpha * d.dzl) + (su + alpha * d.dsu) * (zu + alpha * d.dzu);
          else { W(SP_SL1, i) = sl + alpha * d.dsl; W(SP_SU1, i) = su + alpha * d.dsu; W(SP_ZL1, i) = zl + alpha * d.dzl; W(SP_ZU1, i) = zu + alpha * d.dzu; }
        }
      }
      if (!pass) {
        aff = pol.rsum(aff);
        const double s = mu > 0.0 ? (aff / nb) / mu : 0.0;
        sigmu = s * s * s * mu;
      } else {
        // x moves last: the rows above read the old x
        pol.sync();
        for (int i = i0; i < n; i += di) W(SP_X, i) += alpha * W(SP_DX, i);
        pol.sync();
      }
    }
  }
  if (iters && i0 == 0) *iters = it;
  // ---- active-set finish: A1 = -1 / 0 / +1 for a difference row at its lower / no / upper bound, A2 for the boxes
  for (int i = i0; i < n; i += di) {
    W(SP_A2, i) = W(SP_ZL2, i) > W(SP_SL2, i) ? -1.0 : (W(SP_ZU2, i) > W(SP_SU2, i) ? 1.0 : 0.0);
    if (i < m1) W(SP_A1, i) = W(SP_ZL1, i) > W(SP_SL1, i) ? -1.0 : (W(SP_ZU1, i) > W(SP_SU1, i) ? 1.0 : 0.0);
  }
  pol.sync();
  bool ok = false;
  for (int round = 0; round < 8 && converged && !ok; ++round) {
    // reduced system: boxes by a 1/delta penalty on the diagonal (h), active difference rows through
    // the tridiagonal Schur complement S = A1 diag(h) A1' + delta I  (SP_MD / SP_ME, m1 rows)
    for (int i = i0; i < n; i += di) {
      W(SP_RD, i) = 1.0 / (1.0 + delta + (W(SP_A2, i) != 0.0 ? 1.0 / delta : 0.0));       // h
      W(SP_L2, i) = 0.0;
      W(SP_PL2, i) = 0.0;                                                                  // x of this round
      if (i < m1) W(SP_L1, i) = 0.0;
    }
    pol.sync();
    for (int k = i0; k < m1; k += di) {
      const double c = W(SP_C, k);
      const bool on = W(SP_A1, k) != 0.0;
      W(SP_MD, k) = on ? c * c * (W(SP_RD, k) + W(SP_RD, k + 1)) + delta : 1.0;
      if (k + 1 < m1) W(SP_ME, k) = (on && W(SP_A1, k + 1) != 0.0) ? -c * W(SP_C, k + 1) * W(SP_RD, k + 1) : 0.0;
    }
    pol.sync();
    pol.tri_factor(W, m1);
    for (int rf = 0; rf < 6; ++rf) {
      // residuals of the KKT system at (x, lam1, lam2) = (PL2, L1, L2); PU2 <- rhs_x, PL1 <- r3
      for (int i = i0; i < n; i += di) {
        double r1 = -W(SP_Q, i) - W(SP_PL2, i) - W(SP_L2, i);
        if (i > 0) r1 -= W(SP_C, i - 1) * W(SP_L1, i - 1);
        if (i < m1) r1 += W(SP_C, i) * W(SP_L1, i);
        const double a2 = W(SP_A2, i);
        const double r3 = a2 != 0.0 ? (a2 > 0.0 ? W(SP_HI, i) : lim.v_min) - W(SP_PL2, i) : 0.0;
        W(SP_PL1, i) = r3;
        W(SP_PU2, i) = r1 + r3 / delta;
      }
      pol.sync();
      for (int k = i0; k < m1; k += di) {
        const double a1 = W(SP_A1, k);
        const double r2 = a1 != 0.0 ? (a1 > 0.0 ? lim.a_max : lim.a_min) - ax1(SP_PL2, k) : 0.0;
        // bv = A1 (h rhs_x) - r2 on the active rows
        const double t0 = W(SP_RD, k) * W(SP_PU2, k), t1 = W(SP_RD, k + 1) * W(SP_PU2, k + 1);
        W(SP_RHS, k) = a1 != 0.0 ? W(SP_C, k) * (t1 - t0) - r2 : 0.0;
      }
      // S dlam = bv  (the tridiagonal solver works on RHS -> DX; m1 rows)
      pol.sync();
      pol.tri_solve(W, m1);
      for (int k = i0; k < m1; k += di) { if (W(SP_A1, k) == 0.0) W(SP_DX, k) = 0.0; W(SP_L1, k) += W(SP_DX, k); }
      pol.sync();
      for (int i = i0; i < n; i += di) {
        double s = 0.0;                                   // (A1' dlam)_i
        if (i > 0) s += W(SP_C, i - 1) * W(SP_DX, i - 1);
        if (i < m1) s -= W(SP_C, i) * W(SP_DX, i);
        const double dx = W(SP_RD, i) * (W(SP_PU2, i) - s);
        if (W(SP_A2, i) != 0.0) W(SP_L2, i) += (dx - W(SP_PL1, i)) / delta;
        W(SP_PU1, i) = dx;                                // parked: DX still holds dlam for the rows below i
      }
      pol.sync();
      for (int i = i0; i < n; i += di) W(SP_PL2, i) += W(SP_PU1, i);
      pol.sync();
    }
    // primal-dual active-set update
    bool changed = false;
    const double t = 1e-9;
    for (int i = i0; i < n; i += di) {
      const double x = W(SP_PL2, i), a2 = W(SP_A2, i), l2 = W(SP_L2, i);
      double n2 = a2;
      if (a2 < 0.0 && l2 > t) n2 = 0.0;
      if (a2 > 0.0 && l2 < -t) n2 = 0.0;
      if (a2 == 0.0 && x < lim.v_min - t) n2 = -1.0;
      if (a2 == 0.0 && x > W(SP_HI, i) + t) n2 = 1.0;
      if (n2 != a2) { W(SP_A2, i) = n2; changed = true; }
      if (i < m1) {
        const double a = ax1(SP_PL2, i), a1 = W(SP_A1, i), l1 = W(SP_L1, i);
        double n1 = a1;
        if (a1 < 0.0 && l1 > t) n1 = 0.0;
        if (a1 > 0.0 && l1 < -t) n1 = 0.0;
        if (a1 == 0.0 && a < lim.a_min - t) n1 = -1.0;
        if (a1 == 0.0 && a > lim.a_max + t) n1 = 1.0;
        if (n1 != a1) { W(SP_A1, i) = n1; changed = true; }
      }
    }
    ok = !pol.any(changed);
  }
  // ---- certificate on the active-set point (or the interior-point iterate when that failed)
  double worst = 1e300;
  if (ok) {
    worst = 0.0;
    for (int i = i0; i < n; i += di) {
      const double x = W(SP_PL2, i);
      double st = x + W(SP_Q, i) + W(SP_L2, i);
      if (i > 0) st += W(SP_C, i - 1) * W(SP_L1, i - 1);
      if (i < m1) st -= W(SP_C, i) * W(SP_L1, i);
      worst = std::fmax(worst, std::fabs(st));
      worst = std::fmax(worst, std::fmax(lim.v_min - x, x - W(SP_HI, i)));
      const double l2 = W(SP_L2, i);
      worst = std::fmax(worst, std::fmax(l2, 0.0) * std::fabs(W(SP_HI, i) - x));
      worst = std::fmax(worst, std::fmax(-l2, 0.0) * std::fabs(x - lim.v_min));
      if (i < m1) {
        const double a = ax1(SP_PL2, i), l1 = W(SP_L1, i);
        worst = std::fmax(worst, std::fmax(lim.a_min - a, a - lim.a_max));
        worst = std::fmax(worst, std::fmax(l1, 0.0) * std::fabs(lim.a_max - a));
        worst = std::fmax(worst, std::fmax(-l1, 0.0) * std::fabs(a - lim.a_min));
      }
    }
  }
  worst = pol.rmax(worst);
  const bool certified = ok && worst <= cert_tol;
  for (int i = i0; i < n; i += di) v[i * v_stride] = certified ? W(SP_PL2, i) : W(SP_X, i);
  return certified ? SP_SOLVED : SP_INACCURATE;
}

// one thread per path
MPMPC_HD int sp_solve(int n, const double* li, const double* kappa, long in_stride, const SpLimits& lim, double eps,
                      const SpWork& W, double* v, long v_stride, int* iters) {
  return sp_solve_t(SpSerial{}, n, li, kappa, in_stride, lim, eps, W, v, v_stride, iters);
}

}  // namespace mpmpc
