"""Row partition of the block system over the GPUs of one node.

Every block of the block vectors is split in contiguous row ranges; with the
node-major lexicographic numbering of the synthetic meshes a contiguous range
is a slab of z-planes (y-rows in 2-D) with all components of a node together.
Pressure planes are split evenly and the velocity slab follows (plane 2k of the
Q2 grid sits on plane k of the Q1 grid), so the B / B^T halos stay one plane
thick.  Multiplier rows are split evenly by index (n_lambda is tiny).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


@dataclass
class SlabPlan:
    world: int
    global_sizes: list            # [n_u, n_p, n_lambda] in dofs
    offsets: list                 # per block: int64 array of world+1 global dof offsets
    node_offsets_u: np.ndarray    # velocity NODE offsets (world+1)
    node_offsets_p: np.ndarray    # pressure node offsets
    ncomp: int
    global_nnz_A: int = 0

    def generator_ranges(self, rank: int):
        """Tuple for problems.generate(row_ranges=...); None when unpartitioned."""
        if self.world == 1:
            return None
        return (int(self.node_offsets_u[rank]), int(self.node_offsets_u[rank + 1]),
                int(self.node_offsets_p[rank]), int(self.node_offsets_p[rank + 1]),
                int(self.offsets[-1][rank]), int(self.offsets[-1][rank + 1]))

    def local_sizes(self, rank: int):
        return [int(o[rank + 1] - o[rank]) for o in self.offsets]


def slab_partition(dim: int, n_cells: int, n_lambda: int, world: int, ncomp: int, stokes=True) -> SlabPlan:
    n1u = 2 * n_cells + 1 if stokes else n_cells + 1
    n1p = n_cells + 1
    plane_u, plane_p = n1u ** (dim - 1), n1p ** (dim - 1)
    if stokes:
        zp = np.array([(r * n1p) // world for r in range(world + 1)], np.int64)
        zu = 2 * zp
        zu[-1] = n1u
    else:
        zu = np.array([(r * n1u) // world for r in range(world + 1)], np.int64)
        zp = zu
    nu, npn = zu * plane_u, zp * plane_p
    lam = np.array([(r * n_lambda) // world for r in range(world + 1)], np.int64)
    offsets = [nu * ncomp] + ([npn] if stokes else []) + [lam]
    sizes = [int(o[-1]) for o in offsets]
    return SlabPlan(world, sizes, offsets, nu, npn, ncomp)


def slab_partition_interface(dim: int, n_cells: int, n_immersed_nodes: int, world: int, ncomp: int = 1) -> SlabPlan:
    """elliptic_interface / elasticity (problems.elliptic_interface2d, problems.elasticity3d): background Q1 nodes in
    z-slabs (y-rows in 2-D), the immersed unknowns and the multipliers -- two blocks of the same size -- split evenly
    by node with the SAME offsets (M maps between them)."""
    base = slab_partition(dim, n_cells, n_immersed_nodes * ncomp, world, ncomp, stokes=False)
    fg = np.array([(r * n_immersed_nodes) // world for r in range(world + 1)], np.int64) * ncomp
    offsets = [base.offsets[0], fg, fg.copy()]
    return SlabPlan(world, [int(o[-1]) for o in offsets], offsets, base.node_offsets_u, base.node_offsets_u, ncomp)


def slab_partition_stokes3d(n_cells: int, immersed_refine: int, world: int) -> SlabPlan:
    """The bench workload: 3-D Taylor-Hood + cubed sphere (problems.stokes3d_sphere)."""
    n_lambda = 3 * (6 * 4 ** immersed_refine + 2)
    return slab_partition(3, n_cells, n_lambda, world, ncomp=3, stokes=True)


def partitioned_geometric_aggregates(params: dict, plan: SlabPlan, a: int = 2, min_coarse: int = 600,
                                     max_levels: int = 7):
    """Geometric aggregates that respect a slab partition: boxes of a^dim nodes are formed
    inside every rank's slab (z measured from the slab's first plane), coarse unknowns are
    numbered rank-major, so no aggregate spans two ranks and every level is again a
    contiguous row partition.  Returns a list over levels of
        (agg_global, n_coarse_global, coarse_offsets[world+1], fine_offsets[world+1])
    where agg_global[i] is the GLOBAL coarse id of global fine dof i (or -1).  With
    world == 1 this is problems.geometric_aggregates()."""
    dim, ncomp = params["dim"], params["ncomp"]
    n1 = params["degree"] * params["n_cells"] + 1
    world = plan.world
    idx = np.arange(n1 ** dim, dtype=np.int64)
    coords = np.stack([(idx // n1 ** d) % n1 for d in range(dim)], axis=1)
    interior = np.all((coords > 0) & (coords < n1 - 1), axis=1)
    node_off = np.asarray(plan.node_offsets_u, np.int64)
    rank_of = np.searchsorted(node_off, idx, side="right") - 1
    plane = n1 ** (dim - 1)
    zstart = node_off // plane                      # first z-plane (last axis) of every slab
    cur = coords[interior].copy()
    cur[:, dim - 1] -= zstart[rank_of[interior]]    # slab-local coordinate along the split axis
    cur_rank = rank_of[interior]
    owner = -np.ones(idx.size, np.int64)
    owner[interior] = np.arange(cur.shape[0])
    fine_off = node_off * ncomp
    levels = []
    first = True
    while len(levels) < max_levels:
        box = cur // a
        span = int(box.max()) + 1
        key = cur_rank.astype(np.int64)
        for d in reversed(range(dim)):
            key = key * span + box[:, d]
        uniq, inv = np.unique(key, return_inverse=True)       # rank-major, lexicographic inside a rank
        urank = uniq // (span ** dim)
        node_agg = np.where(owner >= 0, inv[np.maximum(owner, 0)], -1) if first else inv
        agg = node_agg[:, None] * ncomp + np.arange(ncomp)[None, :]
        agg = np.where(node_agg[:, None] < 0, -1, agg).astype(np.int32).ravel()
        n_coarse = int(uniq.size * ncomp)
        coff = np.searchsorted(urank, np.arange(world + 1), side="left").astype(np.int64) * ncomp
        levels.append((agg, n_coarse, coff, fine_off))
        if n_coarse <= min_coarse or uniq.size == cur.shape[0]:
            break
        nxt = np.zeros((uniq.size, dim), np.int64)
        nxt[inv] = box
        cur, cur_rank = nxt, urank
        fine_off = coff
        first = False
    return levels


def local_aggregates(levels, rank: int):
    """Per-rank view for Context.set_aggregates / set_aggregate_partition."""
    out = []
    for agg, n_coarse, coff, foff in levels:
        out.append((agg[int(foff[rank]):int(foff[rank + 1])], n_coarse, coff))
    return out


def local_prolongators(levels, params: dict, plan: SlabPlan, rank: int):
    """Per-rank view of problems.tensor_prolongators(...) for a slab-partitioned context (Context.set_prolongator +
    set_aggregate_partition): level 0 -> this rank's rows of P and the coarse offsets that name, for every coarse unknown,
    the rank owning the fine node it sits on (so that its fine support lies in that rank's rows + halo of A); levels >= 1
    whole (they are replicated).  Returns [(Csr, n_coarse, coarse_offsets or None), ...]."""
    from .problems import Csr
    dim, ncomp, degree = params["dim"], params["ncomp"], params["degree"]
    nf = degree * params["n_cells"]                              # cells of the fine nodal grid
    nc = params["n_cells"] if degree > 1 else (params["n_cells"] + 1) // 2
    plane_f = (nf + 1) ** (dim - 1)
    zf = np.asarray(plan.node_offsets_u, np.int64) // plane_f   # first fine plane of every slab (world + 1)
    # interior coarse planes k = 1 .. nc - 1 sit on fine plane (k * nf) // nc; interior coarse nodes per plane: (nc - 1)^(dim-1)
    k = np.arange(1, nc)
    owner = np.searchsorted(zf, (k * nf) // nc, side="right") - 1
    per_plane = (nc - 1) ** (dim - 1) * ncomp
    coff = np.array([int(np.sum(owner < r)) * per_plane for r in range(plan.world + 1)], np.int64)
    P0, n1 = levels[0][0], levels[0][1]
    assert coff[-1] == n1, (coff, n1)
    r0, r1 = int(plan.offsets[0][rank]), int(plan.offsets[0][rank + 1])
    out = [(P0.slice_rows(r0, r1), n1, coff)]
    for P, n in levels[1:]:
        out.append((P, n, None))
    return out
