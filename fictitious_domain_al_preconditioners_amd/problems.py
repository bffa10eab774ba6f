"""Synthetic fictitious-domain problems (inputs of the hot path).

Thin ctypes front-end of ``csrc/synth/synth.cpp``.  The reference gets these
operators from deal.II FE assembly (immersed_laplace.cc:278-496,
stokes_immersed_boundary.cc:410-820); deal.II is not available, so the
generator builds structurally faithful CSR blocks on tensor grids -- the
concrete instances are the rows of SURVEY.md section 8(d).

Nothing here does solver arithmetic; it only produces host CSR arrays.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "libalfd_synth.so")
_lib = None


class _Params(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("degree", C.c_int32), ("ncomp", C.c_int32), ("n_cells", C.c_int32),
        ("lo", C.c_double), ("hi", C.c_double),
        ("stokes", C.c_int32), ("grad_div", C.c_int32),
        ("gamma_grad_div", C.c_double), ("beta", C.c_double),
        ("center", C.c_double * 3), ("radius", C.c_double),
        ("immersed_refine", C.c_int32), ("coupling_nq", C.c_int32),
        ("body_force", C.c_double * 3), ("embedded_value", C.c_double * 3),
        ("u_node0", C.c_int64), ("u_node1", C.c_int64), ("p_node0", C.c_int64), ("p_node1", C.c_int64),
        ("l0", C.c_int64), ("l1", C.c_int64),
        ("immersed_kind", C.c_int32), ("imm_cells", C.c_int32),
        ("imm_lo", C.c_double), ("imm_hi", C.c_double), ("beta2", C.c_double),
        ("want_surface_mass", C.c_int32), ("assembly", C.c_int32),
        ("elasticity", C.c_int32), ("pad2_", C.c_int32),
        ("lame_lambda", C.c_double), ("lame_mu", C.c_double),
        ("lame2_lambda", C.c_double), ("lame2_mu", C.c_double),
        ("box_lo", C.c_double * 3), ("box_hi", C.c_double * 3),
        ("box_cells", C.c_int32 * 3), ("pad3_", C.c_int32),
    ]


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(
                f"{_LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(_LIB_PATH)
        lib.alfd_synth_generate.restype = C.c_void_p
        lib.alfd_synth_generate.argtypes = [C.POINTER(_Params), C.c_char_p, C.c_int]
        lib.alfd_synth_free.argtypes = [C.c_void_p]
        lib.alfd_synth_matrix.restype = C.c_int
        lib.alfd_synth_matrix.argtypes = [
            C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
            C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.POINTER(C.c_int32)),
            C.POINTER(C.POINTER(C.c_double))]
        lib.alfd_synth_vector.restype = C.c_int
        lib.alfd_synth_vector.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64),
                                          C.POINTER(C.POINTER(C.c_double))]
        lib.alfd_synth_permute_nodes.restype = C.c_int
        lib.alfd_synth_permute_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        lib.alfd_synth_transpose.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = lib
    return _lib


class _NativeHandle:
    """Owns one generator result; freed when the last array viewing it dies."""

    def __init__(self, ptr):
        self.ptr = C.c_void_p(ptr)

    def __del__(self):
        if self.ptr and _lib is not None:
            _lib.alfd_synth_free(self.ptr)
            self.ptr = None


def _view(ptr, n, ctype, dtype, owner):
    """Zero-copy numpy view of native memory that keeps `owner` alive."""
    if n <= 0:
        return np.zeros(0, dtype)
    buf = (ctype * n).from_address(C.addressof(ptr.contents))
    buf._owner = owner
    return np.frombuffer(buf, dtype=dtype)


@dataclass
class Csr:
    """Host CSR block: int64 row_ptr, int32 col (ascending per row), fp64 val."""
    nrows: int
    ncols: int
    row_ptr: np.ndarray
    col: np.ndarray
    val: np.ndarray

    @property
    def nnz(self) -> int:
        return int(self.row_ptr[-1])

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.val, self.col, self.row_ptr), shape=(self.nrows, self.ncols))

    @staticmethod
    def from_scipy(m) -> "Csr":
        m = m.tocsr()
        m.sort_indices()
        return Csr(m.shape[0], m.shape[1], m.indptr.astype(np.int64), m.indices.astype(np.int32),
                   m.data.astype(np.float64))

    def transpose(self) -> "Csr":
        lib = _load()
        rp = np.empty(self.ncols + 1, np.int64)
        col = np.empty(self.nnz, np.int32)
        val = np.empty(self.nnz, np.float64)
        lib.alfd_synth_transpose(self.nrows, self.ncols, self.row_ptr.ctypes.data, self.col.ctypes.data,
                                 self.val.ctypes.data, rp.ctypes.data, col.ctypes.data, val.ctypes.data)
        return Csr(self.ncols, self.nrows, rp, col, val)

    def diagonal(self, row_offset: int = 0) -> np.ndarray:
        """Diagonal of a (row slice of a) square matrix; row r is global row r + row_offset."""
        d = np.zeros(self.nrows)
        rows = np.repeat(np.arange(self.nrows, dtype=np.int64), np.diff(self.row_ptr))
        mask = (rows + row_offset) == self.col
        d[rows[mask]] = self.val[mask]
        return d

    def slice_rows(self, r0: int, r1: int) -> "Csr":
        k0, k1 = int(self.row_ptr[r0]), int(self.row_ptr[r1])
        return Csr(r1 - r0, self.ncols, (self.row_ptr[r0:r1 + 1] - k0).astype(np.int64),
                   self.col[k0:k1].copy(), self.val[k0:k1].copy())


@dataclass
class SyntheticProblem:
    """CSR blocks + right-hand sides of one fictitious-domain system.

    mats: "A" (n_u x n_u), "Ct" (n_u x n_l), "C", "M", "K" (immersed mass /
    stiffness), and for Stokes "B", "Bt", "Mp".  vecs: "f", "g" (and "rhs_p").
    """
    params: dict
    mats: dict = field(default_factory=dict)
    vecs: dict = field(default_factory=dict)
    _handle: object = None
    row_ranges: tuple = None   # (u_node0, u_node1, p_node0, p_node1, l0, l1) of this rank, or None

    @property
    def block_sizes(self):
        """LOCAL block sizes (== global when the problem is not partitioned)."""
        n_u = self.mats["A"].nrows
        n_l = self.mats["C"].nrows
        if "A2" in self.mats:       # elliptic interface: [u (background), u2 (immersed), lambda]
            return [n_u, n_l, n_l]
        if "B" in self.mats:
            return [n_u, self.mats["B"].nrows, n_l]
        return [n_u, n_l]

    @property
    def global_sizes(self):
        n_u = self.mats["A"].ncols
        n_l = self.mats["Ct"].ncols
        if "B" in self.mats:
            return [n_u, self.mats["B"].ncols and self.mats["Mp"].ncols, n_l]
        return [n_u, n_l]

    inv_w_override: object = None   # set by a caller that wants another W^-1 (e.g. 1/M_ii, operator form)

    def inv_w_diag_squared(self) -> np.ndarray:
        """W^-1 = 1 / M_ii^2 (stokes_immersed_boundary.cc:976-978, immersed_laplace.cc:866-869)."""
        if self.inv_w_override is not None:
            return np.asarray(self.inv_w_override, np.float64)
        d = self.mats["M"].diagonal(self.row_ranges[4] if self.row_ranges else 0)
        return 1.0 / (d * d)

    def inv_w_diag_of_mass_squared(self) -> np.ndarray:
        """W^-1 = 1 / (M M)_ii, the true diagonal of M^2 (utilities.h:348-374,
        elliptic_interface.cc:726)."""
        m = self.mats["M"]          # M is symmetric: (M M)_ii = sum_k M_ik^2, row-local (works on a row slice)
        if not np.all(np.diff(m.row_ptr) > 0):      # reduceat would return a neighbour's entry for an empty row
            raise ValueError("the immersed mass matrix has an empty row")
        return 1.0 / np.add.reduceat(m.val * m.val, m.row_ptr[:-1])

    def rho_bound(self) -> float:
        """||A_Gamma||_inf / min_i M_ii (immersed_laplace.cc:609-614)."""
        k = self.mats["K"]
        rows = np.add.reduceat(np.abs(k.val), k.row_ptr[:-1])
        return float(rows.max() / self.mats["M"].diagonal().min())

    def mp_lumped_inv(self) -> np.ndarray:
        """1 / (Mp 1)_i  (stokes_immersed_boundary.cc:946-954)."""
        mp = self.mats["Mp"]
        s = np.add.reduceat(mp.val, mp.row_ptr[:-1])
        return 1.0 / s

def generate(dim=2, degree=1, ncomp=1, n_cells=16, lo=0.0, hi=1.0, stokes=False, grad_div=False,
             gamma_grad_div=0.0, beta=1.0, center=(0.5, 0.5, 0.5), radius=0.2, immersed_refine=3,
             coupling_nq=3, body_force=(0.0, 0.0, 0.0), embedded_value=(1.0, 0.0, 0.0),
             row_ranges=None, immersed_box=None, beta2=0.0, surface_mass=False,
             elasticity=None, immersed_box3d=None, immersed_segments=0, assembly="kronecker") -> SyntheticProblem:
    """immersed_box = (lo, hi, cells): the immersed domain is the 2-D box [lo,hi]^2
    with cells^2 Q1 cells (volume coupling, elliptic_interface); beta2 scales "A2".
    elasticity = (lambda, mu, lambda_jump, mu_jump): vector-Q1 linear elasticity on the background
    (utilities.h:377-427), A2 = the same form with the jump parameters on the immersed box;
    immersed_box3d = (lo[3], hi[3], cells[3]): 3-D box meshed with trilinear cells (volume coupling).
    row_ranges = (u_node0, u_node1, p_node0, p_node1, l0, l1): generate only this
    rank's rows (node ranges for the background spaces, dof range for the
    multiplier); column indices stay global.  None = the whole problem."""
    lib = _load()
    p = _Params()
    p.dim, p.degree, p.ncomp, p.n_cells = dim, degree, ncomp, n_cells
    p.lo, p.hi = lo, hi
    p.stokes, p.grad_div = int(stokes), int(grad_div)
    p.gamma_grad_div, p.beta = gamma_grad_div, beta
    for i in range(3):
        p.center[i] = center[i] if i < len(center) else 0.0
        p.body_force[i] = body_force[i] if i < len(body_force) else 0.0
        p.embedded_value[i] = embedded_value[i] if i < len(embedded_value) else 0.0
    p.radius, p.immersed_refine, p.coupling_nq = radius, immersed_refine, coupling_nq
    rr = tuple(int(v) for v in row_ranges) if row_ranges is not None else (-1,) * 6
    p.u_node0, p.u_node1, p.p_node0, p.p_node1, p.l0, p.l1 = rr
    p.want_surface_mass = int(surface_mass)
    p.assembly = {"kronecker": 0, "cellwise": 1}[assembly]
    if immersed_box is not None:
        p.immersed_kind, p.imm_lo, p.imm_hi, p.imm_cells = 1, float(immersed_box[0]), float(immersed_box[1]), int(immersed_box[2])
        p.beta2 = beta2
    if immersed_segments:       # 2-D circle with exactly this many P1 segments (closed curve: as many nodes)
        p.imm_cells = int(immersed_segments)
    if immersed_box3d is not None:
        p.immersed_kind = 2
        for i in range(3):
            p.box_lo[i], p.box_hi[i] = float(immersed_box3d[0][i]), float(immersed_box3d[1][i])
            p.box_cells[i] = int(immersed_box3d[2][i])
        p.beta2 = beta2
    if elasticity is not None:
        p.elasticity = 1
        p.lame_lambda, p.lame_mu, p.lame2_lambda, p.lame2_mu = (float(v) for v in elasticity)
    err = C.create_string_buffer(256)
    h = lib.alfd_synth_generate(C.byref(p), err, 256)
    if not h:
        raise ValueError("synthetic generator: " + err.value.decode())
    params = dict(dim=dim, degree=degree, ncomp=ncomp, n_cells=n_cells, lo=lo, hi=hi, stokes=stokes,
                  grad_div=grad_div, gamma_grad_div=gamma_grad_div, beta=beta, center=tuple(center),
                  radius=radius, immersed_refine=immersed_refine, coupling_nq=coupling_nq, assembly=assembly)
    owner = _NativeHandle(h)
    pb = SyntheticProblem(params=params, _handle=owner,
                          row_ranges=rr if row_ranges is not None else None)
    _fetch_views(pb)
    return pb


def _fetch_views(pb: "SyntheticProblem"):
    """(Re)binds pb.mats / pb.vecs to the native arrays of pb._handle."""
    lib, owner = _load(), pb._handle
    pb.mats, pb.vecs = {}, {}
    for name in ("A", "B", "Bt", "Mp", "Ct", "C", "M", "K", "A2", "G"):
        nr, nc, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        rp, col, val = C.POINTER(C.c_int64)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_double)()
        if lib.alfd_synth_matrix(owner.ptr, name.encode(), C.byref(nr), C.byref(nc), C.byref(nnz),
                                 C.byref(rp), C.byref(col), C.byref(val)) != 0:
            continue
        pb.mats[name] = Csr(nr.value, nc.value, _view(rp, nr.value + 1, C.c_int64, np.int64, owner),
                            _view(col, nnz.value, C.c_int32, np.int32, owner),
                            _view(val, nnz.value, C.c_double, np.float64, owner))
    for name in ("f", "g", "rhs_p", "f2", "immersed_xyz"):
        n, data = C.c_int64(), C.POINTER(C.c_double)()
        if lib.alfd_synth_vector(owner.ptr, name.encode(), C.byref(n), C.byref(data)) != 0:
            continue
        pb.vecs[name] = _view(data, n.value, C.c_double, np.float64, owner)


def permute_background_nodes(pb: "SyntheticProblem", new_to_old) -> "SyntheticProblem":
    """Renumber the nodes of the background / velocity space IN PLACE (alfd_synth_permute_nodes): node new_to_old[k]
    becomes node k -- what a caller's DoF renumbering (DoFRenumbering::Cuthill_McKee, then component-wise blocks,
    stokes_immersed_boundary.cc:533-541) does to A, Bt / B, Ct / C and f.  pb.node_permutation records new_to_old
    (composed with an earlier one), so that support points and tensor-grid helpers can follow."""
    perm = np.ascontiguousarray(new_to_old, np.int64)
    rc = _load().alfd_synth_permute_nodes(pb._handle.ptr, perm.ctypes.data, perm.size)
    if rc != 0:
        raise ValueError(f"alfd_synth_permute_nodes failed ({rc})")
    prev = getattr(pb, "node_permutation", None)
    _fetch_views(pb)
    pb.node_permutation = perm if prev is None else prev[perm]
    return pb


# ---------------------------------------------------------------------------
# The BASELINE.json configs as concrete synthetic instances.
def laplace2d_circle(n_cells=64, immersed_refine=5, coupling_nq=3, surface_mass=False,
                     immersed_segments=0) -> SyntheticProblem:
    """cfg 1: immersed_laplace 2-D, parameters/circle/Circle_parameters_f0_g1.prm
    (f = 0, g = 1, R = 0.2, centre (0.4, 0.4)), Q1 background on [0,1]^2."""
    return generate(dim=2, degree=1, ncomp=1, n_cells=n_cells, center=(0.4, 0.4, 0.0), radius=0.2,
                    immersed_refine=immersed_refine, coupling_nq=coupling_nq,
                    body_force=(0.0,), embedded_value=(1.0,), surface_mass=surface_mass,
                    immersed_segments=immersed_segments)


def operator_form(pb: SyntheticProblem, gamma: float = 10.0):
    """The "operator form" of immersed_laplace.cc:653-705: gamma <- gamma / h_immersed,
    A <- A + gamma int_Gamma phi_i phi_j (needs generate(surface_mass=True)), W^-1 = 1/M_ii.
    Returns (A_assembled Csr, gamma_h, inv_w)."""
    xyz = pb.vecs["immersed_xyz"].reshape(-1, 3)
    h = float(np.linalg.norm(xyz[1] - xyz[0]))                      # segment length of the circle mesh
    gamma_h = gamma / h
    a = (pb.mats["A"].to_scipy() + gamma_h * pb.mats["G"].to_scipy()).tocsr()
    a.sort_indices()
    return Csr.from_scipy(a), gamma_h, 1.0 / pb.mats["M"].diagonal()


def laplace3d_sphere(n_cells=128, immersed_refine=5, coupling_nq=3) -> SyntheticProblem:
    """cfg 2: immersed_laplace 3-D, Q1 on n^3 cells, cubed-sphere surface R = 0.2."""
    return generate(dim=3, degree=1, ncomp=1, n_cells=n_cells, center=(0.5, 0.5, 0.5), radius=0.2,
                    immersed_refine=immersed_refine, coupling_nq=coupling_nq,
                    body_force=(0.0,), embedded_value=(1.0,))


def stokes3d_sphere(n_cells=64, immersed_refine=4, gamma_grad_div=10.0, coupling_nq=4,
                    row_ranges=None, assembly="kronecker") -> SyntheticProblem:
    """cfg 4 (north star): stokes_immersed_boundary + parameters_stokes_3d.prm.
    Taylor-Hood Q2/Q1 on n^3 cells, grad-div on (prm:21), sphere R = 0.1 centre
    (.5,.5,.5) (stokes_immersed_boundary.cc:427), body force (1,0,0) (prm:52),
    embedded value (-1,1,0) (prm:134)."""
    return generate(dim=3, degree=2, ncomp=3, n_cells=n_cells, stokes=True, grad_div=True,
                    gamma_grad_div=gamma_grad_div, center=(0.5, 0.5, 0.5), radius=0.1,
                    immersed_refine=immersed_refine, coupling_nq=coupling_nq,
                    body_force=(1.0, 0.0, 0.0), embedded_value=(-1.0, 1.0, 0.0), row_ranges=row_ranges,
                    assembly=assembly)


def stokes2d_circle(n_cells=32, immersed_refine=4, gamma_grad_div=10.0, coupling_nq=3) -> SyntheticProblem:
    """2-D Taylor-Hood + immersed circle (what the reference binary is compiled
    for: stokes_immersed_boundary.cc:1218-1219, parameters_stokes.prm)."""
    return generate(dim=2, degree=2, ncomp=2, n_cells=n_cells, stokes=True, grad_div=True,
                    gamma_grad_div=gamma_grad_div, center=(0.5, 0.5, 0.0), radius=0.2,
                    immersed_refine=immersed_refine, coupling_nq=coupling_nq,
                    body_force=(1.0, 0.0), embedded_value=(-1.0, 1.0))


def elliptic_interface2d(n_bg=64, n_fg=16, beta1=1.0, beta2=10.0, coupling_nq=3, row_ranges=None) -> SyntheticProblem:
    """cfg 3: elliptic_interface 2-D + parameters_elliptic_interface/parameters_modified.prm.
    Background Q1 on n_bg^2 cells of [-1,1]^2 (prm:53-54), immersed Q1 on n_fg^2 cells of
    [-0.14,0.47]^2 (prm:55-56), A = beta_1 stiffness, A2 = (beta_2 - beta_1) stiffness
    (elliptic_interface.cc:680-681), f_1 = 1, f_2 - f = 1 (prm:19-29).  beta_2 = 10 in the
    prm (:5); BASELINE.json quotes a jump of 1e3 -- both are valid inputs."""
    return generate(dim=2, degree=1, ncomp=1, n_cells=n_bg, lo=-1.0, hi=1.0, beta=beta1,
                    coupling_nq=coupling_nq, body_force=(1.0,), embedded_value=(0.0,),
                    immersed_box=(-0.14, 0.47, n_fg), beta2=beta2 - beta1, row_ranges=row_ranges)


def elasticity3d(n_bg=16, cells_fg=None, lame_bg=(2.0, 1.0), lame_fg=(20.0, 10.0), coupling_nq=2,
                 box=((-0.65, -0.3, -0.4), (0.65, 0.3, 0.4)), mesh_ratio=2.0, row_ranges=None) -> SyntheticProblem:
    """cfg 5: elliptic_interface 3-D elasticity, parameters_elliptic_interface/elasticity.prm.
    Background vector-Q1 on n_bg^3 cells of [-1.25, 1.25]^3 (prm:55) with lambda, mu = 2, 1
    (prm:25,27); immersed hyper_rectangle [-.65,.65] x [-.3,.3] x [-.4,.4] (prm:56-57) of trilinear
    cells with lambda, mu = 20, 10 (prm:26,28): A = elasticity(lambda_1, mu_1) (utilities.h:377-427),
    A2 = elasticity(lambda_2 - lambda_1, mu_2 - mu_1) on the box (the vector-valued analogue of
    (beta_2 - beta_1)(grad, grad), elliptic_interface.cc:648-663), f = 1, f_2 - f = 1 per component
    (prm:19-23).  cells_fg = None: immersed cells of about mesh_ratio x the background cell size -- the
    prm itself pairs 4^3 background cells of 0.625 with ONE immersed cell of 1.3 x 0.6 x 0.8 (prm:100-104);
    with h_fg ~ h_bg the distributed multiplier loses inf-sup stability and the outer iteration count
    grows with refinement (12 -> 25 -> 397 outer iterations at n_bg = 16 -> 32 -> 96, measured).
    The reference driver of this prm is missing from the tree (CMakeLists.txt:41), so the instance
    is synthetic by necessity; coupling_nq defaults to 2 (the prm's 5 costs 15x the generation time)."""
    h_fg = 2.5 / n_bg * mesh_ratio
    if cells_fg is None:
        cells_fg = tuple(max(1, int(round((box[1][i] - box[0][i]) / h_fg))) for i in range(3))
    jump = (lame_fg[0] - lame_bg[0], lame_fg[1] - lame_bg[1])
    return generate(dim=3, degree=1, ncomp=3, n_cells=n_bg, lo=-1.25, hi=1.25, coupling_nq=coupling_nq,
                    body_force=(1.0, 1.0, 1.0), embedded_value=(0.0, 0.0, 0.0),
                    elasticity=(lame_bg[0], lame_bg[1], jump[0], jump[1]),
                    immersed_box3d=(box[0], box[1], cells_fg), row_ranges=row_ranges)


def cuthill_mckee_nodes(pb: "SyntheticProblem"):
    """new_to_old node permutation of the background space in Cuthill-McKee order of the node graph of A (what
    DoFRenumbering::Cuthill_McKee produces before the component-wise block sort of stokes_immersed_boundary.cc:
    533-541 moves the velocity block in front: inside that block the nodes keep this order).  SciPy's routine
    returns the REVERSE ordering; deal.II's default is the plain one."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    A, nc = pb.mats["A"], pb.params["ncomp"]
    rows0 = np.arange(0, A.nrows, nc)
    starts = A.row_ptr[rows0]
    lens = A.row_ptr[rows0 + 1] - starts
    idx = np.repeat(starts, lens) + (np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens))
    cols = A.col[idx]
    keep = cols % nc == 0
    g = sp.csr_matrix((np.ones(int(keep.sum()), np.int8), (np.repeat(np.arange(rows0.size), lens)[keep], cols[keep] // nc)),
                      shape=(rows0.size, rows0.size))
    return np.ascontiguousarray(reverse_cuthill_mckee(g, symmetric_mode=True)[::-1]).astype(np.int64)


def row_support_points(params: dict, node_range=None, node_permutation=None):
    """One support point per row of the block-(0,0) operator of a tensor-grid problem (node-major
    numbering: the ncomp rows of a node share its point) -- what DoFTools::map_dofs_to_support_points
    gives a deal.II caller; input of solver.row_blocks_from_points.  node_permutation: the new_to_old node
    renumbering the problem carries (SyntheticProblem.node_permutation)."""
    dim, ncomp = params["dim"], params["ncomp"]
    n1 = params["degree"] * params["n_cells"] + 1
    n0, n_end = (0, n1 ** dim) if node_range is None else (int(node_range[0]), int(node_range[1]))
    idx = np.arange(n0, n_end, dtype=np.int64)
    if node_permutation is not None:
        idx = np.asarray(node_permutation, np.int64)
    h = (params["hi"] - params["lo"]) / (n1 - 1)
    pts = np.stack([params["lo"] + h * ((idx // n1 ** d) % n1) for d in range(dim)], axis=1)
    return np.repeat(pts, ncomp, axis=0)


def brick_row_blocks(params: dict, brick=(16, 4, 1), max_rows: int = 250, node_range=None):
    """Row blocks for Context.set_row_blocks (alfd_set_row_blocks) on the tensor-grid background space:
    the nodes of a brick[0] x brick[1] (x brick[2]) patch of the grid, all components of a node together
    (node-major, the numbering of the block-(0,0) operator).  node_range = (first, last+1) restricts the
    blocks to one rank's slab of nodes (rows are then numbered from that slab's first row).
    Returns (block_ptr, rows)."""
    dim, ncomp = params["dim"], params["ncomp"]
    n1 = params["degree"] * params["n_cells"] + 1
    n0, n_end = (0, n1 ** dim) if node_range is None else (int(node_range[0]), int(node_range[1]))
    idx = np.arange(n0, n_end, dtype=np.int64)
    key = np.zeros(idx.size, np.int64)
    for d in reversed(range(dim)):
        c = (idx // n1 ** d) % n1
        key = key * (n1 // brick[d] + 1) + c // brick[d]
    order = np.argsort(key, kind="stable")                    # nodes of a brick keep their lexicographic order
    sk = key[order]
    starts = np.flatnonzero(np.r_[True, sk[1:] != sk[:-1]])
    node_ptr = np.r_[starts, idx.size]
    per = max(1, max_rows // ncomp)
    if idx.size and np.max(np.diff(node_ptr)) > per:          # split oversized bricks
        cuts = [np.arange(a, b, per) for a, b in zip(node_ptr[:-1], node_ptr[1:])]
        node_ptr = np.r_[np.concatenate(cuts), idx.size]
    rows = (order[:, None] * ncomp + np.arange(ncomp)[None, :]).astype(np.int32).ravel()
    return (node_ptr * ncomp).astype(np.int64), rows


def geometric_aggregates(pb: SyntheticProblem, a: int = 2, min_coarse: int = 600, max_levels: int = 7):
    """Aggregates for the multilevel inner preconditioner on the tensor-grid background
    space: nodes are grouped in a^dim boxes (then boxes of boxes, ...), all components of
    a node stay together (one coarse dof per box and component -- the constant modes ML is
    given in utilities.h:304-311), Dirichlet nodes are left out (-1).  Returns
    [(agg_level0, n_coarse0), (agg_level1, n_coarse1), ...] for Context.set_aggregates."""
    if pb.row_ranges is not None:
        raise ValueError("geometric_aggregates needs the unpartitioned problem")
    P = pb.params
    dim, ncomp = P["dim"], P["ncomp"]
    n1 = P["degree"] * P["n_cells"] + 1
    idx = np.arange(n1 ** dim, dtype=np.int64)
    coords = np.stack([(idx // n1 ** d) % n1 for d in range(dim)], axis=1)     # [node, axis]
    interior = np.all((coords > 0) & (coords < n1 - 1), axis=1)
    levels = []
    # level 0: interior nodes only
    cur = coords[interior]
    owner = -np.ones(idx.size, np.int64)          # fine node -> index into `cur`
    owner[interior] = np.arange(cur.shape[0])
    first = True
    while len(levels) < max_levels:
        box = cur // a
        span = int(box.max()) + 1
        key = np.zeros(box.shape[0], np.int64)
        for d in reversed(range(dim)):
            key = key * span + box[:, d]
        uniq, inv = np.unique(key, return_inverse=True)       # coarse nodes, lexicographic in box index
        if first:
            node_agg = np.where(owner >= 0, inv[np.maximum(owner, 0)], -1)
        else:
            node_agg = inv
        agg = node_agg[:, None] * ncomp + np.arange(ncomp)[None, :]
        agg = np.where(node_agg[:, None] < 0, -1, agg).astype(np.int32).ravel()
        n_coarse = int(uniq.size * ncomp)
        levels.append((agg, n_coarse))
        if n_coarse <= min_coarse or uniq.size == cur.shape[0]:
            break
        # coordinates of the coarse nodes = box indices
        nxt = np.zeros((uniq.size, dim), np.int64)
        nxt[inv] = box
        cur = nxt
        first = False
    return levels


def _interp1d(n_fine_cells: int, n_coarse_cells: int):
    """Linear interpolation between two uniform grids of one interval: fine node i (of n_fine_cells + 1)
    from the coarse nodes either side of it.  Returns (rows, cols, vals) with exact weights 1 and 1/2 on
    nested grids."""
    rows, cols, vals = [], [], []
    for i in range(n_fine_cells + 1):
        num = i * n_coarse_cells                  # position i / n_fine_cells in coarse cells = num / n_fine_cells
        j, rem = divmod(num, n_fine_cells)
        if rem == 0:
            rows.append(i); cols.append(j); vals.append(1.0)
        else:
            fr = rem / n_fine_cells
            rows += [i, i]; cols += [j, j + 1]; vals += [1.0 - fr, fr]
    return np.asarray(rows), np.asarray(cols), np.asarray(vals)


def tensor_prolongators(params: dict, min_coarse: int = 400, max_levels: int = 7, node_permutation=None):
    """Geometric multigrid transfers of the tensor-grid background space, as CSR prolongators for
    Context.set_prolongator (alfd_set_prolongator): level 0 embeds the Q1 space of the SAME mesh into the
    nodal grid of the degree-p space (for Q2 that is linear interpolation from n to 2n cells -- the Q1
    functions are members of the Q2 space), every further level interpolates (bi/tri)linearly from a grid
    of ceil(n/2) cells (nested when n is even).  All components of a node interpolate alike; Dirichlet
    (boundary) nodes are neither interpolated to nor from, coarse unknowns are the INTERIOR nodes of the
    coarse grid in lexicographic order, node-major.  What a deal.II caller takes from MGTransferPrebuilt /
    FETools::get_interpolation_matrix.  node_permutation: new_to_old node renumbering of the fine space
    (SyntheticProblem.node_permutation).  Returns [(Csr P, n_coarse), ...]."""
    import scipy.sparse as sp
    dim, ncomp, degree = params["dim"], params["ncomp"], params["degree"]
    nf = degree * params["n_cells"]               # cells of the fine nodal grid
    nc = params["n_cells"] if degree > 1 else (params["n_cells"] + 1) // 2
    fine_is_full = True                           # level 0 rows = all nodes (boundary rows stay empty)
    levels = []
    while len(levels) < max_levels and nc >= 2:
        r, c, v = _interp1d(nf, nc)
        p1 = sp.csr_matrix((v, (r, c)), shape=(nf + 1, nc + 1))
        if fine_is_full:                          # zero the rows of boundary nodes, keep the row count
            keep = np.ones(nf + 1); keep[0] = keep[-1] = 0.0
            p1f = sp.diags(keep) @ p1
        else:
            p1f = p1[1:-1, :]                     # interior-only numbering on this level
        p1c = p1f[:, 1:-1]                        # coarse interior nodes only
        pd = p1c
        for _ in range(dim - 1):
            pd = sp.kron(p1c, pd)                 # axis 0 fastest: kron(P_z, kron(P_y, P_x))
        pv = sp.kron(pd, sp.identity(ncomp)).tocsr() if ncomp > 1 else pd.tocsr()
        pv.eliminate_zeros()
        pv.sort_indices()
        n_coarse = int(pv.shape[1])
        if not levels and node_permutation is not None:        # level-0 rows follow the problem's node numbering
            perm = np.asarray(node_permutation, np.int64)
            pv = pv[(perm[:, None] * ncomp + np.arange(ncomp)[None, :]).ravel(), :]
        levels.append((Csr.from_scipy(pv), n_coarse))
        if n_coarse <= min_coarse:
            break
        nf, nc, fine_is_full = nc, (nc + 1) // 2, False
    return levels
