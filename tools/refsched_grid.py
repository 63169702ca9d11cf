"""Host only: the reference's octree schedule simulated (tests/refsched.py) under every assumed form of the external ms3.Octree operations;
prints, per form, evaluations minus the README's count (npt-flange at resdiv 400: 46,148,745; fibonacci-showerhead at 350: 14,646,431) and the
number of breadth-first decompositions.   python tools/refsched_grid.py"""
import sys, time, itertools
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from multiprocessing import Pool
def run(args):
    scene, resdiv, target, kw = args
    from scaffold.builder import Builder
    from oracle.oracle import OracleSDF
    from refsched import RefSchedule
    s = Builder().Scene(scene)
    res = np.float32(float(s.Diagonal()) / resdiv)
    r = RefSchedule(OracleSDF(s.tree()), res, **kw).render_all()
    return scene, kw, r.evals - target, r.bfs, r.tris
if __name__ == "__main__":
    jobs = []
    for frm, app, rev, mar in itertools.product(["end", "front"], [False, True], [False, True], [0, 1, 2]):
        kw = dict(move_from=frm, spread_from=frm, spread_append=app, child_rev=rev, margin=mar)
        jobs.append(("npt-flange", 400, 46148745, kw)); jobs.append(("fibonacci-showerhead", 350, 14646431, kw))
    res = {}
    with Pool(24) as p:
        for scene, kw, diff, bfs, tris in p.imap_unordered(run, jobs):
            key = tuple(sorted(kw.items()))
            res.setdefault(key, {})[scene] = (diff, bfs)
            if len(res[key]) == 2:
                print(dict(key), res[key], flush=True)
