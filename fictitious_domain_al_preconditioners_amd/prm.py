"""Reader for the solver-control subset of the reference's deal.II .prm files.

deal.II ParameterHandler syntax [EXT]: ``subsection NAME`` ... ``end``,
``set KEY = VALUE``, ``#`` starts a comment (also trailing, e.g.
``set Initial embedded space refinement = 4#5``, parameters_stokes_3d.prm:8).
Only the keys that steer the hot path are interpreted; everything else is
returned untouched in the nested dict so a caller can pass it through.

Key sources: stokes_immersed_boundary.cc:166-189 (Augmented Lagrangian control),
:348 (Solver), :385-389 (Outer solver control); immersed_laplace.cc:219-230 (AL
preconditioner), :264-268 (Schur solver control); elliptic_interface.cc:252-314
(AL preconditioner, Inner / Iteration number / Outer solver control).
"""
from __future__ import annotations

import re

from . import _abi


def parse(text: str) -> dict:
    """Nested dict of subsections; leaves are stripped strings."""
    root: dict = {}
    stack = [root]
    for lineno, raw in enumerate(text.splitlines(), 1):
        line = raw.split("#", 1)[0].strip()
        if not line:
            continue
        m = re.match(r"subsection\s+(.*)$", line)
        if m:
            stack.append(stack[-1].setdefault(m.group(1).strip(), {}))
            continue
        if line == "end":
            if len(stack) == 1:
                raise ValueError(f"line {lineno}: 'end' without subsection")
            stack.pop()
            continue
        m = re.match(r"set\s+(.*?)\s*=\s*(.*)$", line)
        if not m:
            raise ValueError(f"line {lineno}: cannot parse {raw!r}")
        stack[-1][m.group(1).strip()] = m.group(2).strip()
    if len(stack) != 1:
        raise ValueError("unterminated subsection")
    return root


def parse_file(path: str) -> dict:
    with open(path) as f:
        return parse(f.read())


def _b(v, default):
    return default if v is None else v.strip().lower() == "true"


def _f(v, default):
    return default if v is None else float(v)


def _i(v, default):
    return default if v is None else int(float(v))


def _find(tree: dict, prefix: str):
    for k, v in tree.items():
        if k.startswith(prefix) and isinstance(v, dict):
            return v
    return None


def _control(sec, kind, max_steps, tol, reduce):
    sec = sec or {}
    return _abi.Control(kind, _i(sec.get("Max steps"), max_steps), _f(sec.get("Tolerance"), tol),
                        _f(sec.get("Reduction"), reduce))


def config_from_prm(tree: dict) -> tuple[_abi.Config, dict]:
    """Map a parsed .prm onto alfd_config.  Returns (config, info); info records
    the driver, the branch of solve() the prm selects and any setting the GPU
    path does not implement yet (so callers can refuse instead of guessing)."""
    info = {"unsupported": []}
    top = _find(tree, "Distributed Lagrange")
    ell = tree.get("Elliptic Interface Problem")
    if top is not None and "Augmented Lagrangian control" in top:
        # stokes_immersed_boundary: ALControl defaults stokes...:166-177
        info["driver"] = "stokes_immersed_boundary"
        info["solver"] = top.get("Solver", "IBStokesAL")
        al = top["Augmented Lagrangian control"]
        spd = _b(al.get("Diagonal SPD preconditioner"), False)
        cfg = _abi.default_config(_abi.AL_STOKES_DIAG if spd else _abi.AL_STOKES)
        cfg.gamma = _f(al.get("Gamma"), 10.0)
        cfg.gamma_grad_div = _f(al.get("Gamma Grad-div"), 10.0)
        cfg.grad_div_in_A = int(_b(al.get("Grad-div stabilization"), True))
        cfg.inner = _abi.Control(_abi.CTRL_ABS, _i(al.get("Max steps"), 100),
                                 _f(al.get("Tolerance for Augmented Lagrangian"), 1e-4), 0.0)
        cfg.log_level = 1 if _b(al.get("Log result"), True) else 0
        # ReductionControl defaults stokes...:385-389: 1000 / 1e-12 / 1e-10
        cfg.outer = _control(top.get("Outer solver control"), _abi.CTRL_REDUCTION, 1000, 1e-10, 1e-12)
        info["diagonal_W"] = _b(al.get("Diagonal mass immersed"), True)
        info["amg_for_augmented_block"] = _b(al.get("AMG for augmented block"), True)
        if info["solver"] != "IBStokesAL":
            info["unsupported"].append(f"Solver = {info['solver']} (non-AL branch, out of scope)")
        if not info["diagonal_W"]:
            # stokes...:979-985: invW = M^-1 M^-1 (UMFPACK there, CG on slot M here)
            cfg.w_inverse = _abi.W_MASS_INV_SQUARED
        if not cfg.grad_div_in_A:
            info["unsupported"].append("Grad-div stabilization = false (nested Bt Mp^-1 B in Aug)")
        if spd:
            # stokes...:1055-1064: the block-diagonal SPD preconditioner is used with SolverMinRes
            cfg.outer_solver = _abi.OUTER_MINRES
        return cfg, info
    if top is not None:
        # immersed_laplace: Solver in {CG, ELMAN_triang, rational, augmented}
        info["driver"] = "immersed_laplace"
        info["solver"] = top.get("Solver", "augmented")
        cfg = _abi.default_config(_abi.AL2)
        cfg.gamma = 10.0                                           # immersed_laplace.cc:647
        cfg.inner = _abi.Control(_abi.CTRL_ABS, 100, 1e-2, 0.0)     # immersed_laplace.cc:907
        # ReductionControl "Schur solver control", defaults immersed_laplace.cc:264-268
        cfg.outer = _control(top.get("Schur solver control"), _abi.CTRL_REDUCTION, 1000, 1e-12, 1e-12)
        al = top.get("AL preconditioner", {})
        info["use_operator_form"] = _b(al.get("Use operator version"), False)
        info["diagonal_W"] = _b(al.get("Use diagonal inverse"), False)
        if info["solver"] != "augmented":
            info["unsupported"].append(f"Solver = {info['solver']} (non-AL branch, out of scope)")
        if info["use_operator_form"]:
            # immersed_laplace.cc:653-705: the caller assembles gamma/h int_Gamma phi_i phi_j into A
            # and passes gamma = 10/h_immersed (h is a mesh quantity the .prm does not hold)
            cfg.aug_assembled = 1
            info["gamma_needs_h_scaling"] = True
        if not info["diagonal_W"]:
            # immersed_laplace.cc:859-877: M^-1 in operator form, (M^-1)^2 otherwise (UMFPACK there,
            # CG on the immersed mass matrix here -- the caller uploads slot M)
            cfg.w_inverse = _abi.W_MASS_INV if info["use_operator_form"] else _abi.W_MASS_INV_SQUARED
        return cfg, info
    if ell is not None:
        info["driver"] = "elliptic_interface"
        al = ell.get("AL preconditioner", {})
        modified = _b(al.get("Use modified AL preconditioner"), True)
        cfg = _abi.default_config(_abi.AL_ELL_MODIFIED if modified else _abi.AL_ELL_IDEAL)
        cfg.gamma = _f(al.get("gamma fluid"), 10.0)
        cfg.gamma2 = _f(al.get("gamma solid"), 1e-2)
        fixed = _b(ell.get("Use fixed (inner) iterations"), False)
        if fixed:   # IterationNumberControl, elliptic_interface.cc:888-890
            cfg.inner = _control(ell.get("Iteration number control"), _abi.CTRL_FIXED_ITERS, 30, 1e-4, 0.0)
        else:       # ReductionControl, elliptic_interface.cc:891-892
            cfg.inner = _control(ell.get("Inner solver control"), _abi.CTRL_REDUCTION, 100, 1e-10, 1e-2)
        cfg.outer = _control(ell.get("Outer solver control"), _abi.CTRL_REDUCTION, 100, 1e-10, 1e-2)
        info["solver"] = "modified AL" if modified else "ideal AL"
        info["beta_1"] = _f(ell.get("Beta_1"), 1.0)
        info["beta_2"] = _f(ell.get("Beta_2"), 10.0)
        if "lambda background" in ell:
            # elasticity.prm:25-28 (vector-valued variant; utilities.h:377-427): Lame parameters
            info["elasticity"] = {"lambda_background": _f(ell.get("lambda background"), 2.0),
                                  "mu_background": _f(ell.get("mu background"), 1.0),
                                  "lambda_immersed": _f(ell.get("lambda immersed"), 20.0),
                                  "mu_immersed": _f(ell.get("mu immersed"), 10.0)}
        info["diagonal_W"] = _b(al.get("Use diagonal inverse"), True)
        # parameter sanity of the reference (elliptic_interface.cc:874-884, 912-920)
        if modified and cfg.gamma2 > 20.0:
            raise ValueError("gamma_AL_immersed is too large for modified AL preconditioner")
        if modified and abs(cfg.gamma2 - cfg.gamma) <= 1e-1:
            raise ValueError("For the modified AL preconditioner gamma_1 and gamma_2 should not be too close")
        if not modified and cfg.gamma <= 1.0:
            raise ValueError("Parameter gamma is probably too small for classical AL preconditioner")
        if not modified and abs(cfg.gamma - cfg.gamma2) >= 1e-12:
            raise ValueError("In the ideal case, gamma must be identical")
        h_scaled = _b(al.get("Use h-scaled mass"), False) or _b(al.get("Use operator version"), False)
        info["h_scaled_or_operator_form"] = h_scaled
        if not info["diagonal_W"]:
            # elliptic_interface.cc:703-737: invW = M^-1 with the h-scaled mass / operator form, M^-1 M^-1
            # otherwise (UMFPACK there, CG on slot M here)
            cfg.w_inverse = _abi.W_MASS_INV if h_scaled else _abi.W_MASS_INV_SQUARED
        return cfg, info
    raise ValueError("no known driver section in this .prm")
