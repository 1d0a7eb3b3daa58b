// Dynamic drivable corridor per waypoint (e_y bounds of the horizon), scalar per-thread code that
// compiles for gfx950 (K0 kernels in mpmpc_hip.hip) and for the host (tests/emul, CPU check).
//
// Replaces, for every start waypoint of the path at once,
//   ReferencePath.update_path_constraints(wp_id, N, min_width, safety_margin)   src/reference_path.py:522-648
//   ReferencePath._compute_free_segments(wp, min_width)                         src/reference_path.py:466-520
//   Map.w2m / Map.m2w                                                           src/map.py:77-101
//   skimage.draw.line_aa (third party, compiled only): Zingl's anti-aliased line with the error
//     term kept in float32; the ORDER of the produced cells matters to the segment scan.
// Quirks reproduced, not fixed: the forward projection of the previous borders adds
// delta_s*cos(psi) to BOTH coordinates of the upper border and delta_s*sin(psi) to both of the
// lower one (src/reference_path.py:559-562).
//
// Work split: the rasterised border segment of a waypoint does not depend on where the horizon
// starts, so phase 1 extracts the free segments once per waypoint (one thread each) and phase 2
// walks the horizon for every start waypoint (one thread each), reading phase 1's lists.
#pragma once
#include <cmath>
#include <cstdint>

#ifndef MPMPC_HD
#define MPMPC_HD inline
#endif
#ifndef MPMPC_HOST_DEVICE
#define MPMPC_HOST_DEVICE
#endif

namespace mpmpc {

constexpr int COR_MAXSEG = 8;       // free segments kept per waypoint (Sim_Track needs at most 3); more is an ERROR, not a cut
constexpr int COR_CELL_CAP = 1024;  // cells of one rasterised border line the device stages (Sim_Track: ~95); more is an error
constexpr int COR_MAX_SIDE = 65534; // map cells per side (cell coordinates travel as 16-bit pairs with a +1 bias)
constexpr int COR_E_SEGMENTS = -1, COR_E_CELLS = -2;      // return values of the free-segment scans
constexpr double COR_PI = 3.141592653589793;
constexpr int COR_TRIG = 6;         // per-waypoint trigonometric constants, computed ONCE ON THE HOST (cor_trig_row)

struct MapView {
  const int8_t* data;   // [height x width], 1 free / 0 occupied
  int height, width;
  double ox, oy, res;
};

MPMPC_HOST_DEVICE inline void cor_w2m(const MapView& m, double x, double y, int& dx, int& dy) {
  dx = (int)std::floor((x - m.ox) / m.res);
  dy = (int)std::floor((y - m.oy) / m.res);
}
MPMPC_HD void cor_m2w(const MapView& m, int dx, int dy, double& x, double& y) {
  x = (dx + 0.5) * m.res + m.ox;
  y = (dy + 0.5) * m.res + m.oy;
}
// Occupancy of a cell; anything outside the grid counts as occupied.  (The border cells themselves are validated on
// the host; the anti-aliased neighbours of a line along the map edge may step one cell outside.  The reference would
// raise IndexError past the upper edges and silently wrap around at -1.)
MPMPC_HD bool cor_cell_free(const MapView& m, int x, int y) {
  return (unsigned)x < (unsigned)m.width && (unsigned)y < (unsigned)m.height && m.data[(long)y * m.width + x] == 1;
}
// The six numbers of a waypoint that need libm: cos / sin of its heading (forward projection of the previous borders,
// src/reference_path.py:559-562) and of the two normals psi +- pi/2 wrapped to (-pi, pi] (src/reference_path.py:
// 624-631).  Computed by the host's libm - the same one numpy's results in golden G3 agree with bit for bit - in
// mpmpc_set_path_geometry, so that the device tables contain no device-libm result at all.
inline double cor_wrap_host(double a) {
  double t = std::fmod(a + COR_PI, 2.0 * COR_PI);
  if (t < 0.0) t += 2.0 * COR_PI;
  return t - COR_PI;
}
inline void cor_trig_row(double psi, double* t) {
  const double au = cor_wrap_host(COR_PI / 2 + psi), al = cor_wrap_host(-COR_PI / 2 + psi);
  t[0] = std::cos(psi); t[1] = std::sin(psi);
  t[2] = std::cos(au); t[3] = std::sin(au);
  t[4] = std::cos(al); t[5] = std::sin(al);
}

// Scans the anti-aliased line from the upper border cell (r0,c0) to the lower one (r1,c1) and calls
// visit(r, c) for every produced cell in skimage's order, the start cell included.
template <class F>
MPMPC_HD void cor_line_aa(int r0, int c0, int r1, int c1, F visit) {
  const int dc = c0 < c1 ? c1 - c0 : c0 - c1, dr = r0 < r1 ? r1 - r0 : r0 - r1;
  float err = (float)(dc - dr);
  const int sign_c = c0 < c1 ? 1 : -1, sign_r = r0 < r1 ? 1 : -1;
  const float ed = (dc + dr == 0) ? 1.0f : (float)std::sqrt((double)(dc * dc + dr * dr));
  const float fdc = (float)dc, fdr = (float)dr;
  int c = c0, r = r0;
  for (;;) {
    visit(r, c);
    const float ep = err;
    const int cp = c;
    if (2.0f * ep >= -fdc) {
      if (c == c1) break;
      if (ep + fdr < ed) visit(r + sign_r, c);
      err = err - fdr;
      c += sign_c;
    }
    if (2.0f * ep <= fdr) {
      if (r == r1) break;
      if (fdc - ep < ed) visit(r, cp + sign_c);
      err = err + fdc;
      r += sign_r;
    }
  }
}

// Phase 1, one waypoint: free runs of its border segment that are wider than min_width.
// seg[4*s + {0,1,2,3}] = (ub_x, ub_y, lb_x, lb_y) of segment s in world coordinates.
// The scan is a small state machine over the cells of the line in skimage's order; it is written against two
// callables so that the device can separate the three things a cell costs - producing its coordinates (serial
// arithmetic), fetching its occupancy (a memory round trip) and the state machine - and overlap the fetches:
//   n_cells                      cells of the line after the skipped first one (the reference scans x_list[1:])
//   cell(k, x, y)                coordinates of cell k
//   is_free(k)                   occupancy of cell k
template <class Cell, class Free>
MPMPC_HD int cor_scan_cells(const MapView& m, int ux, int uy, int lx, int ly, double min_width, int n_cells, Cell cell,
                            Free is_free_at, double* seg) {
  int count = 0;
  int sx = ux, sy = uy;       // start cell of the current run
  bool in_free = false;
  for (int k = 0; k < n_cells; ++k) {
    int x, y;
    cell(k, x, y);
    const bool is_free = is_free_at(k);
    if (is_free) in_free = true;
    if ((!is_free || (x == lx && y == ly)) && in_free) {
      double ax, ay, bx, by;
      cor_m2w(m, sx, sy, ax, ay);
      cor_m2w(m, x, y, bx, by);
      const double len = std::sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by));
      if (len > min_width) {
        if (count == COR_MAXSEG) return COR_E_SEGMENTS;
        seg[4 * count + 0] = ax; seg[4 * count + 1] = ay; seg[4 * count + 2] = bx; seg[4 * count + 3] = by;
        ++count;
      }
      sx = x; sy = y;
      in_free = false;
    } else if (!is_free && !in_free) {
      sx = x; sy = y;
    }
  }
  return count;
}
// cells of the line (first one skipped) as 16-bit pairs (x + 1) | (y + 1) << 16 into idx[0..cap); returns how many
// there are (possibly more than cap: the caller reports COR_E_CELLS)
MPMPC_HD int cor_line_cells(int ux, int uy, int lx, int ly, int* idx, int cap) {
  int n = -1;                                           // -1: the first cell is skipped
  cor_line_aa(ux, uy, lx, ly, [&](int x, int y) {
    if (n >= 0 && n < cap) idx[n] = (x + 1) | ((y + 1) << 16);
    ++n;
  });
  return n;
}
MPMPC_HD void cor_unpack_cell(int id, int& x, int& y) { x = (id & 0xffff) - 1; y = ((id >> 16) & 0xffff) - 1; }
// one thread walks the line and reads the grid as it goes (host emulation; the device stages the cells of a line in
// LDS and lets the lanes of a wavefront fetch the occupancies: mpmpc_free_segments_kernel).  Returns the number of
// free segments wider than min_width, or COR_E_SEGMENTS when there are more than COR_MAXSEG.
MPMPC_HD int cor_free_segments(const MapView& m, double bux, double buy, double blx, double bly, double min_width,
                               double* seg) {
  int ux, uy, lx, ly;
  cor_w2m(m, bux, buy, ux, uy);
  cor_w2m(m, blx, bly, lx, ly);
  int count = 0;
  int sx = ux, sy = uy;       // start cell of the current run
  bool in_free = false, first = true, overflow = false;
  cor_line_aa(ux, uy, lx, ly, [&](int x, int y) {
    if (first) { first = false; return; }              // the reference skips the first cell (x_list[1:])
    const bool is_free = cor_cell_free(m, x, y);
    if (is_free) in_free = true;
    if ((!is_free || (x == lx && y == ly)) && in_free) {
      double ax, ay, bx, by;
      cor_m2w(m, sx, sy, ax, ay);
      cor_m2w(m, x, y, bx, by);
      const double len = std::sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by));
      if (len > min_width) {
        if (count == COR_MAXSEG) overflow = true;
        else { seg[4 * count + 0] = ax; seg[4 * count + 1] = ay; seg[4 * count + 2] = bx; seg[4 * count + 3] = by; ++count; }
      }
      sx = x; sy = y;
      in_free = false;
    } else if (!is_free && !in_free) {
      sx = x; sy = y;
    }
  });
  return overflow ? COR_E_SEGMENTS : count;
}

struct PathGeom {
  const double *x, *y, *psi, *ds_next;   // per waypoint; ds_next[i] = |wp[i+1] - wp[i]| (circular)
  int n_wp;
  int circular;
  const double* trig;                    // [n_wp x COR_TRIG], cor_trig_row of every waypoint (host libm)
};

// sign(wrap(atan2(py - wy, px - wx) - psi)) without an arctangent: the wrapped angle between the heading and the
// direction to the point is positive iff the point lies to the left of the heading, i.e. iff the cross product
// cos(psi) dy - sin(psi) dx is positive; on the heading line itself the angle is 0 (ahead: sign 0) or pi, which
// np.mod wraps to -pi (behind: sign -1).  Same value as the reference's expression (src/reference_path.py:598-603)
// wherever that one is not itself decided by the last bit of its arctangent.
MPMPC_HD double cor_side(double wx, double wy, double cpsi, double spsi, double px, double py) {
  const double dx = px - wx, dy = py - wy;
  const double cross = cpsi * dy - spsi * dx;
  if (cross > 0.0) return 1.0;
  if (cross < 0.0) return -1.0;
  return (cpsi * dx + spsi * dy) < 0.0 ? -1.0 : 0.0;
}

// From the chosen border cells (pux, puy) / (plx, ply) of a waypoint to its bounds: sign by side, safety margin,
// collapse, and the border cells of the selection WITHOUT the margin projected onto the waypoint's normal (what the
// next waypoint of the horizon measures its candidates against).  o = ub, lb, prev_ux, prev_uy, prev_lx, prev_ly.
constexpr int COR_WPC = 6;
MPMPC_HD void cor_bounds(double wx, double wy, const double* tr, double pux, double puy, double plx, double ply,
                         double safety_margin, double* o) {
  const double su = cor_side(wx, wy, tr[0], tr[1], pux, puy);
  const double sl = cor_side(wx, wy, tr[0], tr[1], plx, ply);
  double ub = su * std::sqrt((pux - wx) * (pux - wx) + (puy - wy) * (puy - wy));
  double lb = sl * std::sqrt((plx - wx) * (plx - wx) + (ply - wy) * (ply - wy));
  ub -= safety_margin;
  lb += safety_margin;
  if (ub < lb) { ub = 0.0; lb = 0.0; }
  o[0] = ub;
  o[1] = lb;
  o[2] = wx + (ub + safety_margin) * tr[2];
  o[3] = wy + (ub + safety_margin) * tr[3];
  o[4] = wx - (lb - safety_margin) * tr[4];
  o[5] = wy - (lb - safety_margin) * tr[5];
}
// A waypoint with at most one free segment leaves nothing to choose, whatever the horizon did before it: its
// bounds are computed once per waypoint (phase 1) instead of once per (start waypoint, column) - on Sim_Track
// that is every waypoint of the free map and all but a handful with the obstacles.
MPMPC_HD void cor_forced(const PathGeom& g, const double* segs, const int* nseg, int i, double safety_margin, double* o) {
  const double wx = g.x[i], wy = g.y[i];
  const double* s = segs + (long)i * 4 * COR_MAXSEG;
  const double* tr = g.trig + (long)i * COR_TRIG;
  if (nseg[i] == 1) cor_bounds(wx, wy, tr, s[0], s[1], s[2], s[3], safety_margin, o);
  else cor_bounds(wx, wy, tr, wx, wy, wx, wy, safety_margin, o);
}

// One column of phase 2 when the waypoint has several free segments: the largest one at the first waypoint of the
// horizon, otherwise the one closest to the forward projection of the previous column's border cells (prev: rows 2..5
// of its cor_bounds output; ip = previous waypoint).  o <- this column's cor_bounds output.
MPMPC_HD void cor_choose(const PathGeom& g, const double* segs, const int* nseg, int i, int ip, bool first,
                         const double* prev, double safety_margin, double* o) {
  const double wx = g.x[i], wy = g.y[i];
  const double* s = segs + (long)i * 4 * COR_MAXSEG;
  const int cnt = nseg[i];
  int best = 0;
  if (first) {
    double best_len = -1.0;
    for (int k = 0; k < cnt; ++k) {
      const double dx = s[4 * k] - s[4 * k + 2], dy = s[4 * k + 1] - s[4 * k + 3];
      const double len = std::sqrt(dx * dx + dy * dy);
      if (len > best_len) { best_len = len; best = k; }
    }
  } else {
    const double shift = g.ds_next[ip];            // wp_prev - wp (distance)
    const double cp = g.trig[(long)ip * COR_TRIG], sp = g.trig[(long)ip * COR_TRIG + 1];
    const double qux = prev[2] + shift * cp, quy = prev[3] + shift * cp;
    const double qlx = prev[4] + shift * sp, qly = prev[5] + shift * sp;
    double best_off = 0.0;
    for (int k = 0; k < cnt; ++k) {
      const double du = std::sqrt((s[4 * k] - qux) * (s[4 * k] - qux) + (s[4 * k + 1] - quy) * (s[4 * k + 1] - quy));
      const double dl = std::sqrt((s[4 * k + 2] - qlx) * (s[4 * k + 2] - qlx) + (s[4 * k + 3] - qly) * (s[4 * k + 3] - qly));
      const double off = (du + dl) / 2;
      if (k == 0 || off < best_off) { best_off = off; best = k; }
    }
  }
  cor_bounds(wx, wy, g.trig + (long)i * COR_TRIG, s[4 * best], s[4 * best + 1], s[4 * best + 2], s[4 * best + 3], safety_margin, o);
}
MPMPC_HD int cor_wp(const PathGeom& g, int i) { return i >= g.n_wp ? (g.circular ? i % g.n_wp : g.n_wp - 1) : i; }

// Phase 2, one start waypoint `wp_id`: ub / lb for the n_cols waypoints wp_id .. wp_id+n_cols-1.
// Returns false when the first horizon waypoint has no free segment (the reference raises there).
// wpc: the per-waypoint rows of cor_forced (waypoints with at most one free segment).
MPMPC_HD bool cor_select(const PathGeom& g, const double* segs, const int* nseg, int wp_id, int n_cols,
                         double safety_margin, double* ub_out, double* lb_out, const double* wpc) {
  double o[COR_WPC] = {0, 0, 0, 0, 0, 0};
  for (int n = 0; n < n_cols; ++n) {
    const int i = cor_wp(g, wp_id + n);
    const int cnt = nseg[i];
    if (n == 0 && cnt == 0) return false;
    if (cnt <= 1) {
      for (int k = 0; k < COR_WPC; ++k) o[k] = wpc[(long)i * COR_WPC + k];
    } else {
      double prev[COR_WPC];
      for (int k = 0; k < COR_WPC; ++k) prev[k] = o[k];
      cor_choose(g, segs, nseg, i, cor_wp(g, wp_id + n - 1), n == 0, prev, safety_margin, o);
    }
    ub_out[n] = o[0];
    lb_out[n] = o[1];
  }
  return true;
}

// The same for ONE column n of the horizon that starts at wp_id, on its own (the device gives every (start
// waypoint, column) pair a thread): a column is sequential only through the run of multi-segment waypoints right
// before it, which is replayed from the last forced column (or from the first waypoint's largest-segment rule).
// Returns false when the horizon's first waypoint has no free segment.
MPMPC_HD bool cor_select_one(const PathGeom& g, const double* segs, const int* nseg, int wp_id, int n,
                             double safety_margin, const double* wpc, double* ub, double* lb) {
  if (nseg[cor_wp(g, wp_id)] == 0) return false;
  int m = n;                                   // first column of the run to replay
  while (m > 0 && nseg[cor_wp(g, wp_id + m)] >= 2) --m;
  double o[COR_WPC] = {0, 0, 0, 0, 0, 0};
  for (int c = m; c <= n; ++c) {
    const int i = cor_wp(g, wp_id + c);
    if (nseg[i] <= 1) {
      for (int k = 0; k < COR_WPC; ++k) o[k] = wpc[(long)i * COR_WPC + k];
    } else {
      double prev[COR_WPC];
      for (int k = 0; k < COR_WPC; ++k) prev[k] = o[k];
      cor_choose(g, segs, nseg, i, cor_wp(g, wp_id + c - 1), c == 0, prev, safety_margin, o);
    }
  }
  *ub = o[0];
  *lb = o[1];
  return true;
}

}  // namespace mpmpc
