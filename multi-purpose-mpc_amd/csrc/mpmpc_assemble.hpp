// K1 math of the batched LTV-MPC QP path: the 27 stage fields of one (instance, stage) - what MPC._init_problem builds
// (src/MPC.py:61-155) with the linearisation of src/spatial_bicycle_models.py:391-417 - written against the lane backend
// (lane_gpu.hpp on the device, lane_emu.hpp in the tests).  The stand-alone assembly kernel stores them (assemble_lane), the
// solve kernels collect them in registers (assemble_fields) and go on with mpmpc_core.hpp's Solver.
#pragma once
#include "mpmpc.h"

#ifndef MPMPC_HD
#define MPMPC_HD inline
#endif
#ifndef MPMPC_UNROLL
#define MPMPC_UNROLL _Pragma("unroll")
#endif

namespace mpmpc {

enum Field { F_DS = 0, F_A10 = 1, F_A20 = 2, F_B20 = 3, F_BEQ = 4, F_LO = 7, F_HI = 12, F_Q = 17, F_P = 22 };

constexpr double INFTY = 1e30, MIN_SCALING = 1e-4, MAX_SCALING = 1e4;
constexpr double RHO_MIN = 1e-6, RHO_MAX = 1e6, RHO_TOL = 1e-4, RHO_EQ_FACTOR = 1e3;
constexpr double INF_BOUND = INFTY * MIN_SCALING;   // a scaled bound beyond this is "infinite"
// The dynamics rows of phase 1 are as soft as OSQP's: its ADMM iteration weights the violation of an equality row
// RHO_EQ_FACTOR = 1000 times that of an inequality row, so its limit point on an infeasible QP trades a little dynamics
// violation (~1e-5) for 0.5 % less box violation - enough to decide instances within that of OSQP's threshold.  Phase 1
// minimises  sum_boxes (scaled violation)^2 + RHO_EQ_FACTOR sum_dynamics (scaled residual)^2:  the equality block of its KKT
// systems carries -1 / RHO_EQ_FACTOR, and its least-violation point leaves the dynamics rows by nu / RHO_EQ_FACTOR.
constexpr double P1_EQ_SOFT = 1.0 / RHO_EQ_FACTOR;

// lanes that hold the N + 1 stages of an instance: a power of two, 16 .. 64 inside a wavefront, 128 / 256 = a workgroup of
// 2 / 4 wavefronts (horizons above 63: lane_gpu.hpp, LaneBlock)
MPMPC_HD int stage_ld(int N) { return N + 1 <= 16 ? 16 : (N + 1 <= 32 ? 32 : (N + 1 <= 64 ? 64 : (N + 1 <= 128 ? 128 : 256))); }

// ------------------------------------------------------------------------------------------
// K1 math: the 27 fields of stage k of one instance.
// ------------------------------------------------------------------------------------------
template <class L>
struct StageIn {
  using R = typename L::real;
  using Mk = typename L::mask;
  R kap, v, ds;        // waypoint wp_id+k      (only used where has_u)
  R kap_p, v_p, ds_p;  // waypoint wp_id+k-1    (only used where !first)
  R x0[3];             // spatial state of the instance
  R cc_a, cc_last;     // previous plan entries cc[3+k] and cc[2N-1]  (src/MPC.py:86-87)
  R lbk, ubk;          // corridor at horizon waypoint k (k >= 1): lb[k-1], ub[k-1]
  Mk first, has_u, terminal;   // k == 0, k < N, k == N
};

// put(f, value) receives the fields one by one, in the order they are formed: K2 collects them in registers (assemble_stage),
// K1 stores each at once (assemble_lane) - 27 values never wait for each other in registers there.
// the curvature-dependent speed cap of src/MPC.py:84,111-113 from the previous plan's entries
template <class L>
MPMPC_HD typename L::real speed_cap(const mpmpc_config& c, const typename L::real& cc_a, const typename L::real& cc_last) {
  using R = typename L::real;
  R kp = tan_(cc_a + cc_last) / R(c.wheelbase);
  R vmax = sqrt_(R(c.ay_max) / (abs_(kp) + R(1e-12)));
  R umax0(c.umax[0]);
  return sel(vmax < umax0, vmax, umax0);
}
// hi_v_pre: the speed cap if the caller has formed it already (K1 does, before it gathers anything else: the tangent is the
// register peak of the stage), else nullptr
template <class L, class Put>
MPMPC_HD void assemble_stage_to(const mpmpc_config& c, const StageIn<L>& in, Put&& put, const typename L::real* hi_v_pre = nullptr) {
  using R = typename L::real;
  const R zero(0.0), one(1.0);
  // linearize(v_ref, kappa_ref, delta_s), same operation order as the reference
  R a10 = (-(in.kap * in.kap)) * in.ds;
  R a20 = ((-in.kap) / in.v) * in.ds;
  R b20 = ((-one) / (in.v * in.v)) * in.ds;
  put(F_DS, sel(in.has_u, in.ds, zero));
  put(F_A10, sel(in.has_u, a10, zero));
  put(F_A20, sel(in.has_u, a20, zero));
  put(F_B20, sel(in.has_u, b20, zero));
  // rhs of equality block k: -x0, or uq_{k-1} = B [v, kappa] - f   (src/MPC.py:107-108)
  R b20p = ((-one) / (in.v_p * in.v_p)) * in.ds_p;
  R f2p = (one / in.v_p) * in.ds_p;
  put(F_BEQ + 0, sel(in.first, -in.x0[0], zero));
  put(F_BEQ + 1, sel(in.first, -in.x0[1], in.ds_p * in.kap_p));
  put(F_BEQ + 2, sel(in.first, -in.x0[2], b20p * in.v_p - f2p));
  // state boxes (src/MPC.py:81-82,119-122)
  put(F_LO + 0, sel(in.first, in.x0[0], in.lbk));
  put(F_HI + 0, sel(in.first, in.x0[0], in.ubk));
  put(F_LO + 1, R(c.xmin[1]));
  put(F_HI + 1, R(c.xmax[1]));
  put(F_LO + 2, R(c.xmin[2]));
  put(F_HI + 2, R(c.xmax[2]));
  // input boxes with the curvature-dependent speed cap (src/MPC.py:84,111-113)
  R hi_v = hi_v_pre ? *hi_v_pre : speed_cap<L>(c, in.cc_a, in.cc_last);
  put(F_LO + 3, sel(in.has_u, R(c.umin[0]), R(-INFTY)));
  put(F_HI + 3, sel(in.has_u, hi_v, R(INFTY)));
  put(F_LO + 4, sel(in.has_u, R(c.umin[1]), R(-INFTY)));
  put(F_HI + 4, sel(in.has_u, R(c.umax[1]), R(INFTY)));
  // cost (src/MPC.py:125,150-155): references are the corridor centre for e_y, (v_ref, kappa_ref) for u
  R xr0 = sel(in.first, zero, (in.lbk + in.ubk) / R(2.0));
  MPMPC_UNROLL
  for (int i = 0; i < 3; ++i) {
    R xr = (i == 0) ? xr0 : zero;
    // terminal stage: -QN . xr with xr = (xr0, 0, 0), i.e. minus the first column of QN times xr0 (src/MPC.py:154)
    const double qn_i0 = i == 0 ? c.QN[0] : c.QN_offdiag[i - 1];
    put(F_Q + i, sel(in.terminal, -(R(qn_i0) * xr0), R(-c.Q[i]) * xr));
    put(F_P + i, sel(in.terminal, R(c.QN[i]), R(c.Q[i])));
  }
  put(F_Q + 3, sel(in.has_u, R(-c.R[0]) * in.v, zero));
  put(F_Q + 4, sel(in.has_u, R(-c.R[1]) * in.kap, zero));
  put(F_P + 3, sel(in.has_u, R(c.R[0]), one));
  put(F_P + 4, sel(in.has_u, R(c.R[1]), one));
}
template <class L>
MPMPC_HD void assemble_stage(const mpmpc_config& c, const StageIn<L>& in, typename L::real out[MPMPC_NUM_FIELDS]) {
  assemble_stage_to<L>(c, in, [&](int f, const typename L::real& v) { out[f] = v; });
}

// Per-path tables uploaded once per handle (device pointers in the library, host pointers in
// the emulation): what ReferencePath.get_waypoint / update_path_constraints provide.
struct PathTables {
  const double* kappa;
  const double* v_ref;
  const double* ds_next;
  int n_wp;
  const double* ub_tab;   // [n_wp x n_cols] or null
  const double* lb_tab;
  int n_cols;
};

// One (instance, stage) of K1: gather the waypoint data, build the fields, store them
// stage-blocked as qp[(field * B + inst) * ld + k] (consecutive lanes -> consecutive addresses).
// (assemble_fields: the 27 fields in registers - what the solve kernel goes on with; assemble_lane: K1, stores them)
template <class L>
MPMPC_HD void gather_stage(const mpmpc_config& c, const PathTables& t, int B, const typename L::ival& inst,
                           const typename L::ival& k, const int* wp_id, const double* x0, const double* cc,
                           const double* lb, const double* ub, StageIn<L>& in, bool with_cc = true) {
  using Mk = typename L::mask;
  using I = typename L::ival;
  const int N = c.N;
  Mk ok = (inst < B) & (k >= 0) & (k <= N);        // (K2 keeps lanes before stage 0: lane_offset)
  in.first = (k == 0);
  in.has_u = ok & (k < N);
  in.terminal = (k == N);
  I wp = L::gatheri(wp_id, inst, ok, 0);
  I ik = wp + k, ip = maxi(wp + k - 1, 0);
  if (c.circular) {
    if (t.n_wp > N) {        // 0 <= wp < n_wp (checked at upload) and k <= N < n_wp: one wrap at most - no integer division
      ik = seli(ik >= t.n_wp, ik - t.n_wp, ik);
      ip = seli(ip >= t.n_wp, ip - t.n_wp, ip);
    } else {
      ik = modi(ik, t.n_wp); ip = modi(ip, t.n_wp);
    }
  } else {
    ik = mini(ik, t.n_wp - 1); ip = mini(ip, t.n_wp - 1);
  }
  in.kap = L::gather(t.kappa, ik, ok, 0.0);
  in.v = L::gather(t.v_ref, ik, ok, 1.0);
  in.ds = L::gather(t.ds_next, ik, ok, 0.0);
  in.kap_p = L::gather(t.kappa, ip, ok, 0.0);
  in.v_p = L::gather(t.v_ref, ip, ok, 1.0);
  in.ds_p = L::gather(t.ds_next, ip, ok, 0.0);
  MPMPC_UNROLL
  for (int i = 0; i < 3; ++i) in.x0[i] = L::gather(x0, inst * 3 + i, ok, 0.0);
  if (with_cc) {
    in.cc_a = L::gather(cc, inst * (2 * N) + k + 3, in.has_u, 0.0);
    in.cc_last = L::gather(cc, inst * (2 * N) + (2 * N - 1), ok, 0.0);
  }
  Mk inner = ok & (k >= 1);
  if (lb != nullptr) {
    in.lbk = L::gather(lb, inst * N + k - 1, inner, 0.0);
    in.ubk = L::gather(ub, inst * N + k - 1, inner, 0.0);
  } else {
    in.lbk = L::gather(t.lb_tab, wp * t.n_cols + k - 1, inner, 0.0);
    in.ubk = L::gather(t.ub_tab, wp * t.n_cols + k - 1, inner, 0.0);
  }
}
template <class L>
MPMPC_HD void assemble_fields(const mpmpc_config& c, const PathTables& t, int B, const typename L::ival& inst,
                              const typename L::ival& k, const int* wp_id, const double* x0, const double* cc,
                              const double* lb, const double* ub, typename L::real* out) {
  StageIn<L> in;
  gather_stage<L>(c, t, B, inst, k, wp_id, x0, cc, lb, ub, in);
  assemble_stage<L>(c, in, out);
}
template <class L>
MPMPC_HD void assemble_lane(const mpmpc_config& c, const PathTables& t, int B, int ld, const typename L::ival& inst,
                            const typename L::ival& k, const int* wp_id, const double* x0, const double* cc,
                            const double* lb, const double* ub, double* qp) {
  using R = typename L::real;
  using I = typename L::ival;
  const typename L::mask ok = (inst < B) & (k >= 0) & (k <= c.N);
  StageIn<L> in;
  // the speed cap first, on its own: its tangent is the register peak of the stage, and nothing else is held while it runs
  in.cc_a = L::gather(cc, inst * (2 * c.N) + k + 3, ok & (k < c.N), 0.0);
  in.cc_last = L::gather(cc, inst * (2 * c.N) + (2 * c.N - 1), ok, 0.0);
  const R hi_v = speed_cap<L>(c, in.cc_a, in.cc_last);
  L::sched_barrier();
  gather_stage<L>(c, t, B, inst, k, wp_id, x0, cc, lb, ub, in, false);
  const I base = inst * ld + k;
  // The row of an instance is written up to the end of the last 128-byte line it touches (zeros behind stage N): N = 30 uses
  // 31 of a row's 32 doubles, and a line that misses its last 8 bytes is a partial write - a read-modify-write in the memory
  // system - for every second line of the output.
  const int kfill = ((c.N + 1 + 15) / 16) * 16 < ld ? ((c.N + 1 + 15) / 16) * 16 : ld;
  const typename L::mask okw = (inst < B) & (k >= 0) & (k < kfill);
  // (one divergent region around all 27 stores: a branch around each would put a wait for the store before it at every join)
  L::when(okw, [&] {
    assemble_stage_to<L>(c, in, [&](int f, const R& v) { L::store(qp, base + f * (B * ld), okw, sel(ok, v, R(0.0))); L::sched_barrier(); }, &hi_v);
  });
}

}  // namespace mpmpc
