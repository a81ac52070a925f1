"""Pareto solution of a sampled multi-objective constrained problem -- host mirror of mobocmf/util/moop.py
(same class surface, same results; the implementation is this build's own).

``MOOP(samples_objs, samples_cons, input_dim, grid_size, pareto_set_size, feasible_values)`` takes posterior function
samples (callables ``f(x: (n, d) ndarray, gradient=False) -> (n,)``, ``gradient=True`` -> (d,) for one point: what
``MFDGP.sample_function_from_each_layer`` returns) and extracts, on a random grid plus the training inputs plus each
objective's constrained optimum, the feasible non-dominated set (objectives are MINIMISED, constraints feasible when
``c(x) >= feasible_value``), optionally thinned to ``pareto_set_size`` points spread over the front.

How it differs inside (results equal the reference's own MOOP, pinned by tests/golden/ref_moop.npz):
  * ``compute_pareto_front``: one stable lexicographic sort, after which every dominator of a row precedes it: a running
    minimum for two objectives, a chunked sweep against the current front otherwise (the reference culls the full array
    once per surviving point);
  * ``compute_pareto_front_and_set_summary_y_space``: farthest-point selection with an incrementally updated
    min-distance vector, O(n) memory (the reference builds the n x n distance matrix).
"""
import numpy as np
import scipy.optimize as spo
import torch


class NotFeasiblePoints(ValueError):
    pass


def _weakly_dominated_by_any(front, p):
    """True if some row q of ``front`` has q <= p in every objective."""
    return front.shape[0] > 0 and bool(np.any(np.all(front <= p, axis=1)))


class MOOP:

    def __init__(self, samples_objs, samples_cons, input_dim, grid_size=1000, pareto_set_size=None, feasible_values=0.0,
                 min_distance_between_points=1e-6, rng=None):
        self.samples_objs = samples_objs
        self.samples_cons = samples_cons
        self.input_dim = input_dim
        self.bounds = [(0.0, 1.0)] * input_dim
        self.grid_size = grid_size
        self.pareto_set_size = pareto_set_size
        self.min_distance_between_points = min_distance_between_points
        self.feasible_values = feasible_values
        self.rng = rng                    # None: numpy's global generator, as the reference (moop.py:232)

    # ------------------------------------------------------------------ distances
    @staticmethod
    def fast_dist(x1, x2):
        """(len(x2), len(x1)) Euclidean distances, squeezed (reference ``fast_dist`` convention, moop.py:33-39)."""
        diff = x1[None, :, :] - x2[:, None, :]
        return np.sqrt((diff * diff).sum(-1)).squeeze()

    # ------------------------------------------------------------------ feasibility
    def _thresholds(self, n_cons, feasible_values):
        if isinstance(feasible_values, np.ndarray):
            return feasible_values
        return np.full(max(n_cons, self.input_dim), float(feasible_values))

    def find_feasible_grid(self, constraints, grid, feasible_values=0.0, allow_negative_constraints=False):
        """Rows of ``grid`` where every constraint sample is >= its threshold (moop.py:41-72).  With no feasible row:
        None, or -- ``allow_negative_constraints`` -- the rows with the smallest total violation."""
        thr = self._thresholds(len(constraints), feasible_values)
        slack = [c(grid) - thr[i] for i, c in enumerate(constraints)]
        ok = np.ones(grid.shape[0], dtype=bool)
        for s in slack:
            ok &= s >= 0
        if ok.any():
            return grid[ok, :]
        if not allow_negative_constraints:
            return None
        violation = np.zeros(grid.shape[0])
        for s in slack:
            violation += np.minimum(s, 0.0)
        return grid[violation == np.max(violation[violation != 0]), :]

    # ------------------------------------------------------------------ constrained optimum of one objective
    def _slsqp(self, obj, cons, x0, tol):
        thr = self._thresholds(len(cons), self.feasible_values)
        fun = lambda x: float(np.asarray(obj(x, gradient=False)).reshape(-1)[0])
        jac = lambda x: np.asarray(obj(x, gradient=True)).reshape(-1)
        g = lambda x: np.array([float(np.asarray(c(x, gradient=False)).reshape(-1)[0]) - tol - thr[i]
                                for i, c in enumerate(cons)])
        dg = lambda x: np.stack([np.asarray(c(x, gradient=True)).reshape(-1) for c in cons]) if cons else \
            np.zeros((0, self.input_dim))
        x = spo.fmin_slsqp(fun, x0.copy(), bounds=self.bounds, disp=0, fprime=jac, f_ieqcons=g, fprime_ieqcons=dg)
        x = np.clip(x, 0.0, 1.0)
        return x, fun(x), g(x)

    def optimize_obj_globally(self, obj, cons, obj_evals, feasible_grid, constraint_tol=1e-6):
        """SLSQP from the best grid point; a second, slightly tightened attempt if the first one does not improve on
        the grid or leaves the feasible set (moop.py:74-139).  Returns (1, d) or None."""
        assert self.input_dim == feasible_grid.shape[1]
        best = int(np.argmin(obj_evals))
        x0, f0 = feasible_grid[best, :], float(np.min(obj_evals))
        x, fx, gx = self._slsqp(obj, cons, x0, 0.0)
        if fx < f0 and np.all(gx >= 0):
            return x[None]
        x, fx, gx = self._slsqp(obj, cons, x0, constraint_tol)
        if fx < f0 and np.all(gx >= -constraint_tol):
            return x[None]
        return None

    # ------------------------------------------------------------------ non-dominated set
    @classmethod
    def compute_pareto_front(cls, pts):
        """Boolean mask of the non-dominated rows of ``pts`` (n, k), all objectives minimised.  A row is dropped when
        another row is <= in every objective; of exactly equal rows the first is kept (moop.py:141-166)."""
        pts = np.asarray(pts)
        n = pts.shape[0]
        mask = np.zeros(n, dtype=bool)
        if n == 0:
            return mask
        # lexicographic order, original index as the last key: every (weak) dominator of a row sorts before it
        k = pts.shape[1]
        order = np.lexsort(tuple([np.arange(n)] + [pts[:, j] for j in range(k - 1, -1, -1)]))
        sp = pts[order]
        if k == 1:
            mask[order[0]] = True
            return mask
        if k == 2:        # sorted by f0: a row survives iff its f1 is strictly below everything seen so far
            seen = np.concatenate(([np.inf], np.minimum.accumulate(sp[:-1, 1])))
            mask[order[sp[:, 1] < seen]] = True
            return mask
        front = np.empty((0, k), dtype=pts.dtype)
        for b0 in range(0, n, 256):
            chunk = sp[b0:b0 + 256]
            alive = np.ones(chunk.shape[0], dtype=bool)
            if front.shape[0]:
                alive &= ~(front[None, :, :] <= chunk[:, None, :]).all(2).any(1)
            keep = []
            for i in np.flatnonzero(alive):           # the few survivors against each other, in sorted order
                if not (keep and _weakly_dominated_by_any(chunk[keep], chunk[i])):
                    keep.append(i)
            if keep:
                front = np.concatenate([front, chunk[keep]], 0)
                mask[order[b0 + np.asarray(keep)]] = True
        return mask

    def obtain_indices_pareto(self, pts):
        """Mask in the order ``pts`` was given (moop.py:168-184; the reference pre-sorts for speed only)."""
        return MOOP.compute_pareto_front(pts)

    # ------------------------------------------------------------------ summary of the front
    def compute_pareto_front_and_set_summary_y_space(self, pareto_set, pareto_front, pareto_set_size):
        """At most ``pareto_set_size`` points: the best of every objective first, then repeatedly the point of the front
        farthest (in objective space) from those already chosen (moop.py:186-219)."""
        assert pareto_set_size > 0
        n, k = pareto_front.shape
        if n <= pareto_set_size:
            return pareto_set, pareto_front
        chosen = np.zeros(pareto_set_size, dtype=np.int64)
        min_dist = np.full(n, np.inf)

        def add(pos, idx):
            chosen[pos] = idx
            diff = pareto_front - pareto_front[idx]
            np.minimum(min_dist, np.sqrt((diff * diff).sum(1)), out=min_dist)

        for j in range(k):
            add(j, int(np.argmin(pareto_front[:, j])))
        for pos in range(k, pareto_set_size):
            add(pos, int(np.argmax(min_dist)))
        return pareto_set[chosen, :], pareto_front[chosen, :]

    # ------------------------------------------------------------------ the whole extraction
    def compute_pareto_solution_from_samples(self, inputs, allow_negative_constraints=False):
        """(pareto_set, pareto_front, samples_objs, samples_cons) as torch float64 / the callables, or None when the
        sampled constraints leave no feasible grid point (moop.py:221-286)."""
        inputs = np.asarray(inputs, dtype=np.float64)
        n_rand = self.input_dim * self.grid_size
        rand = np.random.uniform(size=(n_rand, self.input_dim)) if self.rng is None else \
            self.rng.uniform(size=(n_rand, self.input_dim))
        grid = np.concatenate((rand, inputs))
        grid = self.find_feasible_grid(self.samples_cons, grid, feasible_values=self.feasible_values,
                                       allow_negative_constraints=allow_negative_constraints) \
            if len(self.samples_cons) else grid
        if grid is None:
            return None
        evals = np.stack([np.asarray(obj(grid)).reshape(-1) for obj in self.samples_objs], 1)
        extra = []
        for j, obj in enumerate(self.samples_objs):
            opt = self.optimize_obj_globally(obj, self.samples_cons, evals[:, j], grid)
            if opt is not None and np.min(np.sqrt(((grid - opt) ** 2).sum(1))) > 1e-6:
                extra.append(opt)
        if extra:
            extra = np.concatenate(extra, 0)
            grid = np.vstack((grid, extra))
            evals = np.vstack((evals, np.stack([np.asarray(obj(extra)).reshape(-1) for obj in self.samples_objs], 1)))
        keep = self.obtain_indices_pareto(evals)
        pareto_set, pareto_front = grid[keep, :], evals[keep, :]
        if self.pareto_set_size is not None:
            pareto_set, pareto_front = self.compute_pareto_front_and_set_summary_y_space(pareto_set, pareto_front,
                                                                                        self.pareto_set_size)
        return torch.from_numpy(pareto_set), torch.from_numpy(pareto_front), self.samples_objs, self.samples_cons
