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


def slab_partition_stokes3d(n_cells: int, immersed_refine: int, world: int) -> SlabPlan:
    """The bench workload: 3-D Taylor-Hood + cubed sphere (problems.stokes3d_sphere)."""
    n_lambda = 3 * (6 * 4 ** immersed_refine + 2)
    return slab_partition(3, n_cells, n_lambda, world, ncomp=3, stokes=True)
