"""Occupancy grid for the path module (host side; same public surface as the reference's
src/map.py:16-155: `Map(file_path, origin, resolution, threshold_occupied)`, `w2m`, `m2w`,
`add_obstacles`, `add_boundary`, `.data` with 1 = free / 0 = occupied, `Obstacle(cx, cy, radius)`).

Differences in construction, not in results: no scikit-image dependency (hole removal is a
connected-component pass with scipy.ndimage, the line rasteriser is `line_aa` below), and a grid
can be supplied directly with `Map.from_grid`, which is how the committed fixtures are loaded on
machines that do not have the reference's PNG.
"""
from __future__ import annotations

import math

import numpy as np

OBSTACLE_COLOUR = '#2E4053'


def line_aa(r0, c0, r1, c1):
    """Anti-aliased line, pixel sequence identical to skimage.draw.line_aa (Zingl's algorithm with
    the error term kept in float32, as the compiled original does).  The ORDER of the returned cells
    matters to `ReferencePath._compute_free_segments` (src/reference_path.py:484-518).

    Returns (rr, cc, val) like the original; callers in the reference pass (x_px, y_px) pairs.
    """
    f32 = np.float32
    rr, cc, val = [], [], []
    dc, dr = abs(c0 - c1), abs(r0 - r1)
    err = f32(dc - dr)
    sign_c = 1 if c0 < c1 else -1
    sign_r = 1 if r0 < r1 else -1
    ed = f32(1.0) if dc + dr == 0 else f32(math.sqrt(dc * dc + dr * dr))
    c, r = c0, r0
    fdc, fdr = f32(dc), f32(dr)
    while True:
        cc.append(c)
        rr.append(r)
        val.append(abs(err - fdc + fdr) / ed)
        err_prime = err
        c_prime = c
        if f32(2) * err_prime >= -fdc:
            if c == c1:
                break
            if err_prime + fdr < ed:
                cc.append(c)
                rr.append(r + sign_r)
                val.append(abs(err_prime + fdr) / ed)
            err = f32(err - fdr)
            c += sign_c
        if f32(2) * err_prime <= fdr:
            if r == r1:
                break
            if fdc - err_prime < ed:
                cc.append(c_prime + sign_c)
                rr.append(r)
                val.append(abs(fdc - err_prime) / ed)
            err = f32(err + fdc)
            r += sign_r
    return (np.array(rr, dtype=np.intp), np.array(cc, dtype=np.intp),
            1.0 - np.array(val, dtype=float))


def fill_small_holes(free, area_threshold=5):
    """Occupied specks smaller than `area_threshold` cells (8-connected) become free: what
    skimage.morphology.remove_small_holes(area_threshold=5, connectivity=8) does at src/map.py:113."""
    from scipy import ndimage
    free = np.asarray(free, bool)
    labels, count = ndimage.label(~free, structure=np.ones((3, 3), int))
    if count == 0:
        return free.copy()
    sizes = np.bincount(labels.ravel())
    small = sizes < area_threshold
    small[0] = False
    return free | small[labels]


class Obstacle:
    """Circular obstacle in world coordinates [m]."""

    def __init__(self, cx, cy, radius):
        self.cx, self.cy, self.radius = cx, cy, radius

    def show(self):
        import matplotlib.patches as patches
        import matplotlib.pyplot as plt
        plt.gca().add_patch(patches.Circle(xy=(self.cx, self.cy), radius=self.radius, color=OBSTACLE_COLOUR,
                                           zorder=20))


class Map:
    def __init__(self, file_path, origin, resolution, threshold_occupied=100):
        from PIL import Image
        raw = np.array(Image.open(file_path))[:, :, 0]
        self.threshold_occupied = threshold_occupied
        self.data = raw
        self.process_map()
        self._finish(origin, resolution)

    @classmethod
    def from_grid(cls, data, origin, resolution):
        """Wrap an already processed grid (int8, 1 free / 0 occupied), e.g. a committed fixture."""
        self = cls.__new__(cls)
        self.threshold_occupied = 100
        self.data = np.array(data, dtype=np.int8)
        self._finish(origin, resolution)
        return self

    def _finish(self, origin, resolution):
        self.height, self.width = self.data.shape
        self.resolution = resolution
        self.origin = origin
        self.obstacles = []
        self.boundaries = []

    # world <-> pixel (src/map.py:77-101)
    def w2m(self, x, y):
        return (int(np.floor((x - self.origin[0]) / self.resolution)),
                int(np.floor((y - self.origin[1]) / self.resolution)))

    def m2w(self, dx, dy):
        return ((dx + 0.5) * self.resolution + self.origin[0],
                (dy + 0.5) * self.resolution + self.origin[1])

    def process_map(self):
        free = self.data >= self.threshold_occupied
        self.data = fill_small_holes(free, area_threshold=5).astype(np.int8)

    def add_obstacles(self, obstacles):
        self.obstacles.extend(obstacles)
        for ob in obstacles:
            rad = int(np.ceil(ob.radius / self.resolution))
            cx, cy = self.w2m(ob.cx, ob.cy)
            yy, xx = np.ogrid[-rad:rad, -rad:rad]
            disc = xx ** 2 + yy ** 2 <= rad ** 2
            self.data[cy - rad:cy + rad, cx - rad:cx + rad][disc] = 0

    def add_boundary(self, boundaries):
        self.boundaries.extend(boundaries)
        for (p0, p1) in boundaries:
            sx, gx = self.w2m(p0[0], p0[1]), self.w2m(p1[0], p1[1])
            xs, ys, _ = line_aa(sx[0], sx[1], gx[0], gx[1])
            self.data[ys, xs] = 0
