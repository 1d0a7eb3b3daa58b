"""A stream of batches from HOST buffers through one device, several calls in flight.

A single `mpmpc_solve` from host memory is a chain - copy in, launch, copy out - and its rate is set by PCIe round trips, not
by the kernels: 4 - 9 M solves/s against 60 M with resident inputs.  The C ABI offers the call in two
halves for exactly this (`mpmpc_staged_begin` / `mpmpc_staged_end` on a handle's page-locked staging block, include/mpmpc.h);
this class is the loop around them a caller would otherwise write: `depth` handles on ONE device take the batches in turn, the
upload and launch of batch k + 1 .. k + depth - 1 are in flight while batch k's results come back.

    sb = StreamedBatches(cfg, settings, depth=3)
    sb.set_path(kappa, v_ref, ds_next)
    for sol in sb.map(batches):            # batches: iterable of (wp_id, x0, cc_prev, lb, ub); results in submission order
        ...

serves a Monte-Carlo sweep whose poses are produced on the host (the reference's loop, src/simulation.py:134-140, run for many
scenarios): every batch is a fresh upload.  Nothing here touches the arithmetic: each batch is one `mpmpc_solve_staged` call.
"""
from __future__ import annotations

from collections import deque

import numpy as np

import mpmpc


class StreamedBatches:
    """handles: mpmpc.Handle objects of one configuration on one device (or any object with the same staging / staged_begin /
    staged_end surface - the CPU tests pass an emulation-backed stand-in)."""

    def __init__(self, config=None, settings=None, depth=3, handles=None, copy=True):
        """copy = False: the Solutions are VIEWS into the handles' page-locked staging blocks - valid until `depth` more batches
        have been submitted (a 1 024-instance plan is 1.25 MB: copying it out costs more host time than the call)"""
        if handles is None:
            if config is None:
                raise ValueError("need a configuration or ready handles")
            if depth < 1:
                raise ValueError("depth must be >= 1")
            handles = [mpmpc.Handle(config, settings) for _ in range(depth)]
        self.handles = list(handles)
        if not self.handles:
            raise ValueError("need at least one handle")
        self._views = [None] * len(self.handles)          # staging views per handle, for the batch size they were laid out for
        self._busy = [None] * len(self.handles)           # (B, want_z, want_y) of the call in flight on the handle
        self._order = deque()                             # handle indices in submission order
        self._next = 0
        self._copy = bool(copy)

    @property
    def depth(self):
        return len(self.handles)

    def set_path(self, kappa, v_ref, ds_next):
        for h in self.handles:
            h.set_path(kappa, v_ref, ds_next)

    def set_corridor(self, ub, lb):
        for h in self.handles:
            h.set_corridor(ub, lb)

    def close(self):
        self.drain()
        for h in self.handles:
            if hasattr(h, "close"):
                h.close()

    # ------------------------------------------------------------------------------------------------------------
    def _collect(self, i) -> mpmpc.Solution:
        """wait for the call in flight on handle i and copy its results out of the staging block"""
        B, want_z, want_y = self._busy[i]
        self.handles[i].staged_end()
        v = self._views[i][1]
        self._busy[i] = None
        out = np.array if self._copy else (lambda a: a)
        return mpmpc.Solution(out(v["z"]) if want_z else None, out(v["u0"]), out(v["status"]), out(v["iters"]),
                              out(v["resid"]), out(v["y"]) if want_y else None)

    def submit(self, wp_id, x0, cc_prev, lb=None, ub=None, want_z=True, want_y=False):
        """Start one batch.  -> the Solution of the OLDEST batch in flight if its handle was needed for this one, else None
        (results always come back in submission order: from submit, then from drain).
        If the new batch cannot be started (bad arguments, a device error) the exception carries the Solution that had to be
        collected to make room for it as `exc.done` (or None), and the ring stays where it was: the failed batch took no slot."""
        i = self._next
        done = None
        if self._busy[i] is not None:
            assert self._order and self._order[0] == i
            self._order.popleft()
            done = self._collect(i)
        try:
            wp_id = np.asarray(wp_id)
            B = int(wp_id.size)
            if self._views[i] is None or self._views[i][0] != B:
                self._views[i] = (B, self.handles[i].staging(B))
            v = self._views[i][1]
            v["wp_id"][:] = wp_id
            v["x0"][:] = x0
            v["cc_prev"][:] = cc_prev
            rows = lb is not None
            if rows:
                v["lb"][:] = lb
                v["ub"][:] = ub
            self.handles[i].staged_begin(B, with_rows=rows, want_z=want_z, want_y=want_y)
        except Exception as exc:
            exc.done = done          # already collected: hand it to the caller instead of dropping it
            raise
        # only now does the batch own the slot
        self._next = (i + 1) % len(self.handles)
        self._busy[i] = (B, want_z, want_y)
        self._order.append(i)
        return done

    def drain(self):
        """-> the Solutions of every batch still in flight, oldest first"""
        out = []
        while self._order:
            out.append(self._collect(self._order.popleft()))
        return out

    def discard(self):
        """wait for every batch still in flight and drop its results (a consumer that left a stream early).  EVERY slot is
        drained even if one of them fails - the first error is raised once the ring is empty and back at its first handle, so
        that the next submit never meets a slot that is still marked busy."""
        first_error = None
        while self._order:
            i = self._order.popleft()
            try:
                self.handles[i].staged_end()
            except Exception as exc:          # noqa: BLE001 - kept, raised below
                first_error = first_error or exc
            finally:
                self._busy[i] = None
        self._next = 0
        if first_error is not None:
            raise first_error

    def map(self, batches, want_z=True, want_y=False):
        """generator: Solutions of `batches` (tuples wp_id, x0, cc_prev[, lb, ub]) in order, `depth` of them in flight.
        Whatever a previous, abandoned stream left in flight is discarded first, and a consumer that stops early (break,
        exception, garbage-collected generator) leaves nothing in flight either: the next stream never sees stale results."""
        self.discard()
        try:
            for b in batches:
                try:
                    done = self.submit(*b, want_z=want_z, want_y=want_y)
                except Exception as exc:
                    held = getattr(exc, "done", None)
                    if held is not None:
                        exc.done = None
                        yield held      # the batch before the one that failed is still delivered, in order
                    raise
                if done is not None:
                    yield done
            for s in self.drain():
                yield s
        finally:
            self.discard()
