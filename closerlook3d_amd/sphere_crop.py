"""Sphere crops of a sub-sampled scene on the device (second half: the per-sample part).

The reference's S3DIS dataset answers every `__getitem__` with a KD-tree radius query on the host
(`datasets/S3DIS.py:296-314`: `KDTree.query_radius(pick_point, r=in_radius, return_distance=True,
sort_results=True)`, keep the `num_points` nearest, shuffle, pad by re-drawing valid points) followed by the gathers
that build the sample (`:316-337`).  With the scene resident in HBM the query is one streaming pass -- squared
distances in double, exactly the KD-tree's `rdist` (sum over x, y, z in that order of the squared differences of
the float64 copies it keeps), the inclusive test `rdist <= r*r`, an ascending sort of the survivors by distance --
and the gathers never leave the device.

    scene = SceneCropper(sub_points, colors, labels, in_radius=2.0, num_points=15000, device="cuda")
    inds = scene.query(pick_point)                       # == the KD-tree's sorted, truncated index list
    sample = scene.crop(pick_point, generator=g)         # points (centred), mask, colours, height, labels, input_inds

Equal distances: scikit-learn's sort is not stable, so among points at EXACTLY the same float64 distance from the
pick point its order is unspecified; here they come in ascending index order.  The set of returned points and every
position outside such a tie group are identical (tests/test_sphere_crop.py, against scikit-learn itself).
"""
import ctypes

import torch


class SceneCropper:
    def __init__(self, sub_points, colors=None, labels=None, in_radius=2.0, num_points=15000, device="cuda"):
        self.device = torch.device(device)
        pts = torch.as_tensor(sub_points)
        self.points64 = pts.to(self.device, torch.float64).contiguous()   # the KD-tree keeps float64 copies
        self.colors = None if colors is None else torch.as_tensor(colors).to(self.device)
        self.labels = None if labels is None else torch.as_tensor(labels).to(self.device)
        self.in_radius = float(in_radius)
        self.num_points = int(num_points)

    def _pick(self, pick_point):
        return torch.as_tensor(pick_point, dtype=torch.float64, device=self.device).reshape(3)

    # ---- the engine's kernels (csrc/sphere_crop.hip) when the scene lives on a GPU: one streaming pass + one stable
    # radix sort per query, one gather kernel per sample; the indexed torch operations below remain the CPU path
    def _native(self):
        return self.device.type == "cuda"

    def _sorted_scene(self, pick_point, cap=None):
        """(sorted_idx [cap] int32: the in-sphere scene indices by (distance, index) when there are at most `cap` of
        them; count [1] int32 on the device; the pick point as three host doubles).  Nothing is synchronised here.
        cap: host-known size of the sort (default: four samples' worth, at least 65 536, at most the scene); a caller
        that finds count > cap repeats with cap = P."""
        from . import _lib
        lib = _lib.lib()
        pick = [float(v) for v in torch.as_tensor(pick_point, dtype=torch.float64).reshape(3).cpu().tolist()]
        P = self.points64.shape[0]
        if cap is None:
            cap = min(P, max(4 * self.num_points, 65536))
        if getattr(self, "_ws", None) is None:
            nbytes = max(lib.cl3d_workspace_bytes(16, 1, P, 0, 0, 0), lib.cl3d_workspace_bytes(16, 1, self.num_points, 0, 0, 0))
            self._ws = torch.empty((nbytes,), dtype=torch.uint8, device=self.device)
        sorted_idx = torch.empty((cap,), dtype=torch.int32, device=self.device)
        count = torch.empty((1,), dtype=torch.int32, device=self.device)
        arr = (ctypes.c_double * 3)(*pick)
        with _lib.on_device(self.device):
            _lib.check(lib.cl3d_sphere_crop_query(self.points64.data_ptr(), P, ctypes.cast(arr, ctypes.c_void_p), self.in_radius,
                                                  cap, sorted_idx.data_ptr(), count.data_ptr(), self._ws.data_ptr(),
                                                  self._ws.numel(), _lib.stream_ptr(self.device)))
        return sorted_idx, count, arr

    def query(self, pick_point, limit=True):
        """Scene indices inside the sphere, nearest first, at most `num_points` of them (S3DIS.py:300-306)."""
        if self._native():
            sorted_idx, count, _ = self._sorted_scene(pick_point)
            n = int(count)  # the one host round trip: the length of the list the caller receives
            if n > sorted_idx.numel():  # more points in the sphere than the sort was sized for: the whole scene
                sorted_idx, count, _ = self._sorted_scene(pick_point, cap=self.points64.shape[0])
            return sorted_idx[:min(n, self.num_points) if limit else n].long()
        c = self._pick(pick_point)
        d = self.points64 - c
        rdist = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        inside = torch.nonzero(rdist <= self.in_radius * self.in_radius).reshape(-1)
        order = torch.argsort(torch.sqrt(rdist[inside]), stable=True)
        inds = inside[order]
        return inds[:self.num_points] if limit else inds

    def crop(self, pick_point, generator=None):
        """One sample, as S3DIS.__getitem__ assembles it (:296-329) up to its random draws, which come from
        `generator` (a torch.Generator on this device) instead of numpy's global state:
        dict(points [N,3] f32 centred on the pick point, mask [N] i32, input_inds [N] i64, height [N,1] f32,
        colors / labels gathered if the scene has them)."""
        if self._native():
            return self._crop_native(pick_point, generator)
        c = self._pick(pick_point)
        query = self.query(c)
        cur, N = int(query.numel()), self.num_points
        if cur == 0:
            raise RuntimeError("sphere crop: no scene point within in_radius of the pick point")
        mask = torch.zeros(N, dtype=torch.int32, device=self.device)
        if cur >= N:
            input_inds = query[torch.randperm(N, device=self.device, generator=generator)]
            mask[:] = 1
        else:
            query = query[torch.randperm(cur, device=self.device, generator=generator)]
            pad = torch.randint(0, cur, (N - cur,), device=self.device, generator=generator)
            input_inds = torch.cat([query, query[pad]])
            mask[:cur] = 1
        original = self.points64[input_inds]
        out = {"points": (original - c).to(torch.float32), "mask": mask, "input_inds": input_inds,
               "height": original[:, 2:].to(torch.float32)}
        if self.colors is not None:
            out["colors"] = self.colors[input_inds]
        if self.labels is not None:
            out["labels"] = self.labels[input_inds].to(torch.int64)
        return out

    def _crop_native(self, pick_point, generator, cap=None, u=None):
        """One `self._ws` scratch serves query() and crop(): a cropper is used from ONE stream / thread at a time (as a
        dataset worker does); two concurrent users need two croppers."""
        from . import _lib
        lib = _lib.lib()
        N = self.num_points
        sorted_idx, count, arr = self._sorted_scene(pick_point, cap)
        if u is None:  # the sample's draws: shuffle keys, re-draws -- made ONCE per sample: a retry with a larger sort
            u = torch.rand((2, N), device=self.device, generator=generator)  # capacity reuses them
        points = torch.empty((N, 3), dtype=torch.float32, device=self.device)
        mask = torch.empty((N,), dtype=torch.int32, device=self.device)
        inds = torch.empty((N,), dtype=torch.int64, device=self.device)
        height = torch.empty((N, 1), dtype=torch.float32, device=self.device)
        with _lib.on_device(self.device):
            _lib.check(lib.cl3d_sphere_crop_assemble(self.points64.data_ptr(), sorted_idx.data_ptr(), count.data_ptr(),
                                                     sorted_idx.numel(), N, ctypes.cast(arr, ctypes.c_void_p), u[0].data_ptr(), u[1].data_ptr(),
                                                     points.data_ptr(), mask.data_ptr(), inds.data_ptr(), height.data_ptr(),
                                                     self._ws.data_ptr(), self._ws.numel(), _lib.stream_ptr(self.device)))
        out = {"points": points, "mask": mask, "input_inds": inds, "height": height}
        if self.colors is not None:
            out["colors"] = self.colors[inds]
        if self.labels is not None:
            out["labels"] = self.labels[inds].to(torch.int64)
        n = int(count)  # checked once everything is queued
        if n == 0:
            raise RuntimeError("sphere crop: no scene point within in_radius of the pick point")
        if n > sorted_idx.numel():  # the sphere held more points than the sort was sized for: again, same draws --
            return self._crop_native(pick_point, generator, cap=self.points64.shape[0], u=u)  # the generator's stream
        return out                                                                          # does not depend on `cap`

    def project(self, points, chunk=None, budget_bytes=1 << 30):
        """Index of the nearest sub-sampled point for every ORIGINAL scene point (`datasets/S3DIS.py:262-270`:
        `search_tree.query(points, return_distance=False)`, the projection that carries votes from the sub-sampled
        cloud back to the full one) as int32, computed in float64 like the tree; among exactly equidistant candidates
        (where the tree's choice is unspecified) the smallest index."""
        q = torch.as_tensor(points).to(self.device, torch.float64)
        out = torch.empty(q.shape[0], dtype=torch.int32, device=self.device)
        p = self.points64
        if chunk is None:
            # six [chunk, P] float64 temporaries are alive at once: bound them by `budget_bytes` (a sub-sampled room of
            # 800 000 points at the old fixed chunk of 4096 would have asked for 26 GB each -- ADVICE r2)
            chunk = max(1, min(4096, budget_bytes // max(1, p.shape[0] * 8 * 6)))
        for lo in range(0, q.shape[0], chunk):
            c = q[lo:lo + chunk]
            dx = c[:, 0][:, None] - p[:, 0][None, :]
            dy = c[:, 1][:, None] - p[:, 1][None, :]
            dz = c[:, 2][:, None] - p[:, 2][None, :]
            out[lo:lo + chunk] = torch.argmin((dx * dx + dy * dy) + dz * dz, dim=1).to(torch.int32)
        return out

    # ---- a whole batch of samples at once: the same results with ~30 device operations per BATCH instead of per
    # sample (a single crop is launch-bound: measured 2.7 ms per sample against 0.6 ms for one KD-tree query on a
    # host core, scene of 800 000 points)
    def query_batch(self, picks):
        """picks [B,3] -> (cols, rows, kept): the concatenation over b of query(picks[b]) (`cols`), the sample each
        entry belongs to (`rows`, ascending) and the number of entries per sample (`kept` [B])."""
        c = torch.as_tensor(picks, dtype=torch.float64, device=self.device).reshape(-1, 3)
        B = c.shape[0]
        p = self.points64
        dx = p[:, 0][None, :] - c[:, 0][:, None]
        dy = p[:, 1][None, :] - c[:, 1][:, None]
        dz = p[:, 2][None, :] - c[:, 2][:, None]
        rdist = (dx * dx + dy * dy) + dz * dz
        rows, cols = torch.nonzero(rdist <= self.in_radius * self.in_radius, as_tuple=True)  # row-major
        dist = torch.sqrt(rdist[rows, cols])
        by_dist = torch.argsort(dist, stable=True)
        order = by_dist[torch.argsort(rows[by_dist], stable=True)]      # by (sample, distance, index)
        rows, cols = rows[order], cols[order]
        counts = torch.bincount(rows, minlength=B)
        starts = torch.cumsum(counts, 0) - counts
        rank = torch.arange(rows.numel(), device=self.device) - starts[rows]
        keep = rank < self.num_points
        return cols[keep], rows[keep], torch.clamp(counts, max=self.num_points)

    def crop_batch(self, picks, generator=None):
        """B samples as crop() builds them, stacked: points [B,N,3], mask [B,N], input_inds [B,N], height [B,N,1]
        (+ colors / labels)."""
        c = torch.as_tensor(picks, dtype=torch.float64, device=self.device).reshape(-1, 3)
        B, N = c.shape[0], self.num_points
        cols, rows, kept = self.query_batch(c)
        if bool((kept == 0).any()):
            raise RuntimeError("sphere crop: no scene point within in_radius of a pick point")
        # shuffle inside every sample: sort by (sample, uniform key)
        key = torch.rand(cols.numel(), device=self.device, generator=generator)
        by_key = torch.argsort(key)
        perm = by_key[torch.argsort(rows[by_key], stable=True)]
        cols = cols[perm]
        starts = torch.cumsum(kept, 0) - kept
        j = torch.arange(N, device=self.device)[None, :].expand(B, N)
        valid = j < kept[:, None]
        redraw = (torch.rand((B, N), device=self.device, generator=generator) * kept[:, None]).long()
        redraw = torch.minimum(redraw, kept[:, None] - 1)
        input_inds = cols[torch.where(valid, j, redraw) + starts[:, None]]
        original = self.points64[input_inds]                                   # [B,N,3]
        out = {"points": (original - c[:, None, :]).to(torch.float32), "mask": valid.to(torch.int32),
               "input_inds": input_inds, "height": original[:, :, 2:].to(torch.float32)}
        if self.colors is not None:
            out["colors"] = self.colors[input_inds]
        if self.labels is not None:
            out["labels"] = self.labels[input_inds].to(torch.int64)
        return out


class EpochPlanner:
    """The pick points of S3DIS epochs, planned on the device (`datasets/S3DIS.py:212-253`).

    The reference plans every (epoch, step) once, on the host: the scene whose smallest potential is smallest, in it
    the point of smallest potential, a Gaussian offset, the KD-tree's sorted in-radius list cut to `num_points`, a
    Tukey bump `(1 - d^2 / r^2)^2` added to the potentials of those points -- so the next pick lands elsewhere.  Each
    step depends on the previous one, so this is a sequential loop of small operations either way; what the device
    version buys is that the scenes (already resident for `SceneCropper.crop_batch`) never travel, and that the plan
    can be extended while training runs.

        planner = EpochPlanner([SceneCropper(...), ...], initial_potentials)     # potentials: float64 per scene point
        cloud_inds, point_inds, picks = planner.plan(noise)                      # noise [steps, 3] float64

    Random draws are the caller's (the reference takes them from numpy's global state: `np.random.rand(n) * 1e-3` for
    the initial potentials, `np.random.normal(scale=in_radius / 10, size=(1, 3))` per step).

    `promotion`: the Tukey weights are computed from float32 squared distances divided by `np.square(in_radius)`, a
    float64 SCALAR.  Under the NumPy the reference was written for (< 2.0, value-based casting) the quotient stays
    float32 ('legacy', the default: plans identical to the reference's own era); under NumPy >= 2 (NEP 50) it is float64
    ('nep50').  The two give different potentials in the last bits and, after enough steps, different plans.
    """

    def __init__(self, scenes, potentials, promotion="legacy"):
        if promotion not in ("legacy", "nep50"):
            raise ValueError("promotion must be 'legacy' or 'nep50'")
        self.scenes = list(scenes)
        self.promotion = promotion
        self.potentials = [torch.as_tensor(p, dtype=torch.float64).to(s.device).clone() for p, s in zip(potentials, self.scenes)]
        self.min_potentials = [float(p.min()) for p in self.potentials]

    def step(self, noise):
        """One pick: (cloud index, point index, pick point [3] float64 tensor); potentials updated."""
        cloud = min(range(len(self.scenes)), key=lambda i: (self.min_potentials[i], i))  # np.argmin: first minimum
        scene, pot = self.scenes[cloud], self.potentials[cloud]
        point = int(torch.argmin(pot))  # first occurrence of the minimum, as np.argmin
        centre = scene.points64[point]
        pick = centre + torch.as_tensor(noise, dtype=torch.float64, device=scene.device).reshape(3)
        inds = scene.query(pick)
        d = (scene.points64[inds] - pick).to(torch.float32)  # (points[query_inds] - pick_point).astype(np.float32)
        sq = d * d
        dists = (sq[:, 0] + sq[:, 1]) + sq[:, 2]             # np.sum(..., axis=1) of three float32 values
        r2 = scene.in_radius * scene.in_radius
        if self.promotion == "legacy":
            t = 1.0 - dists / torch.tensor(r2, dtype=torch.float32, device=scene.device)
            tukey = t * t
            tukey = torch.where(dists > torch.tensor(r2, dtype=torch.float32, device=scene.device), torch.zeros_like(tukey), tukey)
        else:
            t = 1.0 - dists.to(torch.float64) / r2
            tukey = t * t
            tukey = torch.where(dists.to(torch.float64) > r2, torch.zeros_like(tukey), tukey)
        pot[inds] += tukey.to(torch.float64)                   # query indices are distinct
        self.min_potentials[cloud] = float(pot.min())
        return cloud, point, pick

    def plan(self, noise):
        """noise [steps, 3] -> (cloud_inds [steps], point_inds [steps], picks [steps, 3] float64 on the host)."""
        clouds, points, picks = [], [], []
        for row in noise:
            c, p, pick = self.step(row)
            clouds.append(c)
            points.append(p)
            picks.append(pick.cpu())
        return clouds, points, torch.stack(picks) if picks else torch.zeros((0, 3), dtype=torch.float64)
