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

    def query(self, pick_point, limit=True):
        """Scene indices inside the sphere, nearest first, at most `num_points` of them (S3DIS.py:300-306)."""
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
