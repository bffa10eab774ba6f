import os as _os, sys as _sys
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
_sys.path.insert(0, _ROOT); _sys.path.insert(0, _os.path.join(_ROOT, "tests"))
import numpy as np, sys, collections
from fictitious_domain_al_preconditioners_amd import problems
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
pb = problems.stokes3d_sphere(n_cells=N, immersed_refine=2)
m = pb.mats["A"]
for brick in ((8,4,2),(8,2,2),(16,4,2)):
    bp, rows = problems.brick_row_blocks(pb.params, brick)
    rng = np.random.default_rng(0)
    tot = 0; shared4 = 0; shared_any = 0
    for b in rng.choice(len(bp)-1, 200, replace=False):
        rr = rows[bp[b]:bp[b+1]]
        cols = np.unique(np.concatenate([m.col[m.row_ptr[r]:m.row_ptr[r+1]] for r in rr]))
        # window positions with gap merging (< 8)
        pos = {}
        W = 0; prev = None
        for c in cols:
            if prev is not None and c - prev < 8: W += c - prev
            elif prev is not None: W += 1
            pos[c] = W; prev = c
        groups = collections.defaultdict(list)
        for r in rr:
            k0, k1 = m.row_ptr[r], m.row_ptr[r+1]
            lc = np.array([pos[c] for c in m.col[k0:k1]])
            key = (k1-k0, (lc - lc[0]).tobytes(), m.val[k0:k1].tobytes())
            groups[key].append(r)
        for key, g in groups.items():
            n = key[0]
            tot += n*len(g)
            shared4 += n*(len(g)//4*4)
            if len(g) > 1: shared_any += n*len(g)
    print(brick, "nnz share in full 4-row shared batches: %.3f" % (shared4/tot), " in groups >1: %.3f" % (shared_any/tot))
