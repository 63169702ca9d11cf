"""Test-side simulation of the reference's octree renderer SCHEDULE (glrender/octreerenderer.go:131-218, marchcubes.go:14-34,
glrender.go:17-36) over the oracle's evaluator: which positions `Octree.ReadTriangles` hands to `Evaluate`, call by call, when
`RenderAll` drains it through its 4096-triangle buffer -- the number `Evaluations()` prints in the reference's README.

What the reference's own files fix: the prune buffer's size (Reset :94-105), one breadth-first decomposition per ReadTriangles call
and only while the prune buffer is empty (:136-146), a centre test of EVERYTHING still in the prune buffer on every call (:147-153,
prune :180-193: survivors are tested again, call after call, as long as the position buffer has room), the depth-first loop with
`currentLim = min(8 * (len(dst) - n), len(posbuf))` (:155-176: corners evaluated but not marched when dst fills up stay in the
buffer and are evaluated again), marchCubes' stop rule (len(dst) - nTri > 5).

What they do not fix ([external]: soypat/geometry ms3.Octree, not vendored) and this module takes as parameters:
  * DecomposeBFS: level by level while 8 x the frontier fits the buffer, children in corner order;
  * DecomposeDFS: pops the last cube, pushes its 8 children in corner order, a Level-2 cube's children go to the position buffer
    (64 positions) instead -- while the buffer has 64 free slots;
  * SafeSpread(cubes, prunecubes, marked): cubes marked Level 0 are replaced by cubes from the prune buffer's END or FRONT (`spread_from`);
  * SafeMove(cubes, prunecubes) when the stack is empty: `move_k` cubes at most (None: as many as leave room for a full descent of
    the deepest one, cap - 8 (Level - 1)), from the prune buffer's END or FRONT (`move_from`).
"""
import numpy as np

OX = np.array([0, 1, 1, 0, 0, 1, 1, 0]); OY = np.array([0, 0, 1, 1, 0, 0, 1, 1]); OZ = np.array([0, 0, 0, 0, 1, 1, 1, 1])
SQRT3 = np.float32(1.73205080757)


def _scale_centered(bb, s):
    bb = np.asarray(bb, np.float32)
    mn, mx = bb[:3], bb[3:]
    c = np.float32(0.5) * (mn + mx)
    sz = mx - mn
    h = np.float32(0.5) * (np.float32(s) * sz)
    return c - h, c + h


class RefSchedule:
    def __init__(self, sdf, res, evalbuf=32768, dstcap=4096, move_k=None, move_from="end", spread_from="end", spread_append=False, child_rev=False, margin=1):
        from oracle.oracle import mc_tables
        self.sdf, self.res = sdf, np.float32(res)
        mn, mx = _scale_centered(sdf.Bounds(), 1.01)
        self.origin = mn
        long_axis = np.float32((mx - mn).max())
        self.levels = int(np.ceil(np.log2(np.float32(long_axis / self.res)))) + 1
        self.n = 1 << (self.levels - 1)
        _, tri = mc_tables()
        self.ntri_of_index = np.array([(row >= 0).sum() // 3 for row in tri], np.uint8)
        self.ntri = {}  # Level-2 cube (x, y, z) -> uint8[8] triangles of its 8 leaves, corner order
        self.evalbuf, self.dstcap = evalbuf & ~7, dstcap
        self.move_k, self.move_from, self.spread_from, self.spread_append = move_k, move_from, spread_from, spread_append
        self.child_rev, self.margin = child_rev, margin
        tbl = {4: 8, 5: 72, 6: 584, 7: 4680}
        self.prune_cap = min(tbl[min(max(self.levels, 4), 7)], self.evalbuf)
        self.cubes_cap = self.levels * 8
        self.evals = 0; self.pruned = 0; self.tris = 0; self.calls = 0; self.bfs = 0

    # ---- geometry as the oracle forms it (orc_render.c: cube_size, cube_origin, CubeCenter, leaf corners)
    def _size(self, level):
        return np.float32(1 << (level - 1)) * self.res

    def _centres(self, cubes):
        c = np.array([(x, y, z) for x, y, z, _ in cubes], np.int64)
        lv = cubes[0][3]
        size = self._size(lv)
        o = self.origin[None, :] + size * (c >> (lv - 1)).astype(np.float32)
        return np.float32(0.5) * (o + (o + size))

    def _leaf_tris(self, x, y, z):
        """triangles of the 8 leaves of Level-2 cube (x, y, z) (leaf coordinates of its origin), corner order"""
        key = (x, y, z)
        t = self.ntri.get(key)
        if t is None:
            self._fill_block(x & ~31, y & ~31, z & ~31)
            t = self.ntri[key]
        return t

    def _fill_block(self, bx, by, bz):
        """evaluate a 32^3 block of leaves at once (8 corners each, Box.Vertices order)"""
        r = np.arange(32)
        zz, yy, xx = np.meshgrid(r + bz, r + by, r + bx, indexing="ij")
        lx, ly, lz = xx.ravel(), yy.ravel(), zz.ravel()
        o = np.stack([self.origin[0] + self.res * lx.astype(np.float32), self.origin[1] + self.res * ly.astype(np.float32),
                      self.origin[2] + self.res * lz.astype(np.float32)], 1).astype(np.float32)
        m = o + self.res
        pos = np.empty((o.shape[0], 8, 3), np.float32)
        for k in range(8):
            pos[:, k, 0] = np.where(OX[k], m[:, 0], o[:, 0]); pos[:, k, 1] = np.where(OY[k], m[:, 1], o[:, 1]); pos[:, k, 2] = np.where(OZ[k], m[:, 2], o[:, 2])
        d = self.sdf.Evaluate(pos.reshape(-1, 3)).reshape(-1, 8)
        idx = ((d < 0) * (1 << np.arange(8))[None, :]).sum(1)
        nt = self.ntri_of_index[idx].reshape(32, 32, 32)  # [z][y][x]
        for z2 in range(0, 32, 2):
            for y2 in range(0, 32, 2):
                for x2 in range(0, 32, 2):
                    self.ntri[(bx + x2, by + y2, bz + z2)] = np.array([nt[z2 + OZ[k], y2 + OY[k], x2 + OX[k]] for k in range(8)], np.uint8)

    # ---- the external octree operations, as assumed
    def _children(self, c):
        x, y, z, lv = c
        h = 1 << (lv - 2)
        ch = [(x + int(OX[k]) * h, y + int(OY[k]) * h, z + int(OZ[k]) * h, lv - 1) for k in range(8)]
        return ch[::-1] if self.child_rev else ch

    def _decompose_bfs(self, start, min_lvl=3):
        front, ok = [start], False
        while front[0][3] > min_lvl and 8 * len(front) <= self.prune_cap:
            front = [ch for c in front for ch in self._children(c)]
            ok = True
        return (front, True) if ok else ([], False)

    def _take(self, src, frm):
        return src.pop() if frm == "end" else src.pop(0)

    def _refill(self):
        if not self.prunecubes:
            return
        for i, c in enumerate(self.cubes):  # SafeSpread
            if c[3] == 0 and self.prunecubes and self.marked > 0:
                self.cubes[i] = self._take(self.prunecubes, self.spread_from)
                self.marked -= 1
        if self.spread_append and self.prunecubes:  # (variant: SafeSpread also uses the stack's free room, as far as a full descent stays safe)
            lv = self.prunecubes[-1][3]
            while self.prunecubes and len(self.cubes) < self.cubes_cap - 8 * (lv - self.margin):
                self.cubes.append(self._take(self.prunecubes, self.spread_from))
        if not self.cubes:  # SafeMove
            lv = self.prunecubes[-1][3]
            k = self.move_k if self.move_k is not None else max(1, self.cubes_cap - 8 * (lv - self.margin))
            while self.prunecubes and len(self.cubes) < k:
                self.cubes.append(self._take(self.prunecubes, self.move_from))

    def _decompose_dfs(self):
        while self.cubes and self.evalbuf - 8 * len(self.posbuf) >= 64:
            c = self.cubes.pop()
            if c[3] == 0:
                self.marked -= 1  # (a marked cube that was never replaced: dropped)
                continue
            if c[3] == 2:
                self.posbuf.extend(self._leaf_tris(c[0], c[1], c[2]).tolist())
            else:
                self.cubes.extend(self._children(c))

    def _prune(self):
        if self.evalbuf - 8 * len(self.posbuf) < len(self.prunecubes):
            return
        d = self.sdf.Evaluate(self._centres(self.prunecubes))
        self.evals += len(self.prunecubes)
        lv = self.prunecubes[0][3]
        max_dist = self._size(lv) * (SQRT3 / np.float32(2))
        keep = ~(np.abs(d) >= max_dist)
        self.pruned += int((~keep).sum()) * (1 << (3 * (lv - 1)))
        self.prunecubes = [c for c, k in zip(self.prunecubes, keep) if k]

    def read_triangles(self):
        """one Octree.ReadTriangles(dst[:dstcap]); returns (n, eof)"""
        self.calls += 1
        n = 0
        upi = next((i for i, c in enumerate(self.cubes) if c[3] >= 3), -1)
        if upi >= 0 and not self.prunecubes:
            self.prunecubes, ok = self._decompose_bfs(self.cubes[upi])
            if ok:
                c = self.cubes[upi]
                self.cubes[upi] = (c[0], c[1], c[2], 0)
                self.marked += 1
                self.bfs += 1
        if self.prunecubes:
            self._prune()
            self._refill()
        while self.dstcap - n > 5:
            if not self.cubes and not self.posbuf and not self.prunecubes:
                return n, True
            if not self.cubes:
                self._refill()
            self._decompose_dfs()
            lim = min(self.dstcap - n, len(self.posbuf))  # (in cubes: currentLim / 8)
            if lim == 0:
                raise RuntimeError("zero buffer")
            self.evals += 8 * lim
            t = np.asarray(self.posbuf[:lim], np.int64)
            cs = np.cumsum(t)
            room = self.dstcap - n
            j = int(np.searchsorted(cs, room - 5, side="left"))  # first cube after which len(dst) - nTri <= 5
            k = min(j + 1, lim)
            n += int(cs[k - 1])
            del self.posbuf[:k]
        return n, False

    def render_all(self):
        self.cubes = [(0, 0, 0, self.levels)]
        self.prunecubes, self.posbuf, self.marked = [], [], 0
        while True:
            n, eof = self.read_triangles()
            self.tris += n
            if eof:
                return self
