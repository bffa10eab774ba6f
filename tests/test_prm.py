"""PRM subset reader against the reference's own parameter files (text embedded
here as minimal excerpts of the solver-control keys; the full files live in
/root/reference and are parsed too when that directory is available)."""
import os

import pytest

from fictitious_domain_al_preconditioners_amd import _abi, prm

STOKES_3D = """
subsection Distributed Lagrange<2,3>
set Initial embedded space refinement            = 4#5
set Solver                                       = IBStokesAL
  subsection Augmented Lagrangian control
set AMG for augmented block            = true
set Diagonal mass immersed             = true
set Gamma                              = 10
set Gamma Grad-div                     = 10
set Grad-div stabilization             = true
set Log result                         = true
set Max steps                          = 100
set Tolerance for Augmented Lagrangian = 1.e-2 # default: 1e-4
  end
  subsection Outer solver control
set Max steps     = 1000   # default: 100
set Reduction     = 1.e-12 # default: 1.e-2
set Tolerance     = 1.e-8
  end
end
"""


def test_parse_comments_and_nesting():
    t = prm.parse(STOKES_3D)
    top = t["Distributed Lagrange<2,3>"]
    assert top["Initial embedded space refinement"] == "4"          # trailing '#5' is a comment
    assert top["Augmented Lagrangian control"]["Tolerance for Augmented Lagrangian"] == "1.e-2"
    with pytest.raises(ValueError):
        prm.parse("subsection A\nset x = 1\n")
    with pytest.raises(ValueError):
        prm.parse("end\n")
    with pytest.raises(ValueError):
        prm.parse("bogus line\n")


def test_stokes_3d_prm_maps_to_the_reference_knobs():
    cfg, info = prm.config_from_prm(prm.parse(STOKES_3D))
    assert info["driver"] == "stokes_immersed_boundary" and info["solver"] == "IBStokesAL"
    assert cfg.variant == _abi.AL_STOKES and cfg.restart == 30
    assert (cfg.gamma, cfg.gamma_grad_div, cfg.grad_div_in_A) == (10.0, 10.0, 1)
    assert (cfg.inner.kind, cfg.inner.max_steps, cfg.inner.tol) == (_abi.CTRL_ABS, 100, 1e-2)
    assert (cfg.outer.kind, cfg.outer.max_steps, cfg.outer.tol, cfg.outer.reduce) == \
        (_abi.CTRL_REDUCTION, 1000, 1e-8, 1e-12)
    assert info["unsupported"] == []
    # it is exactly the default config up to the log level
    d = _abi.default_config(_abi.AL_STOKES)
    d.log_level = cfg.log_level
    assert bytes(d) == bytes(cfg)


def test_defaults_when_keys_are_missing():
    cfg, info = prm.config_from_prm(prm.parse(
        "subsection Distributed Lagrange<1,2>\n subsection Augmented Lagrangian control\n end\nend\n"))
    assert cfg.inner.tol == 1e-4 and cfg.outer.tol == 1e-10 and cfg.outer.reduce == 1e-12   # stokes...:175,385-389


def test_immersed_laplace_and_elliptic_sections():
    lap = prm.parse("subsection Distributed Lagrange<1,2>\n set Solver = augmented\n"
                    " subsection AL preconditioner\n set Use operator version = true\n set Use diagonal inverse = false\n end\n"
                    " subsection Schur solver control\n set Max steps = 1000\n set Tolerance = 1.e-10\n end\nend\n")
    cfg, info = prm.config_from_prm(lap)
    assert cfg.variant == _abi.AL2 and cfg.gamma == 10.0 and cfg.outer.tol == 1e-10 and cfg.outer.reduce == 1e-12
    assert cfg.aug_assembled == 1 and info["gamma_needs_h_scaling"]      # operator form: supported
    assert not info["unsupported"] and cfg.w_inverse == _abi.W_MASS_INV          # operator form + exact W: M^-1
    lap2 = prm.parse("subsection Distributed Lagrange<1,2>\n set Solver = augmented\n"
                     " subsection AL preconditioner\n set Use diagonal inverse = false\n end\nend\n")
    cfg2, info2 = prm.config_from_prm(lap2)
    assert cfg2.w_inverse == _abi.W_MASS_INV_SQUARED and cfg2.aug_assembled == 0  # immersed_laplace.cc:874-877
    assert (cfg2.mass.kind, cfg2.mass.max_steps, cfg2.mass.reduce) == (_abi.CTRL_REDUCTION, 1000, 1e-14)
    ell = prm.parse("subsection Elliptic Interface Problem\n set Beta_2 = 10\n subsection AL preconditioner\n"
                    " set Use modified AL preconditioner = true\n set gamma fluid = 10\n set gamma solid = 1e-2\n end\n"
                    " subsection Inner solver control\n set Max steps = 100000\n set Reduction = 1.e-20\n set Tolerance = 1.e-2\n end\n"
                    " subsection Outer solver control\n set Max steps = 1000\n set Reduction = 1.e-10\n set Tolerance = 1.e-10\n end\nend\n")
    cfg, info = prm.config_from_prm(ell)
    assert cfg.variant == _abi.AL_ELL_MODIFIED and cfg.restart == 50 and (cfg.gamma, cfg.gamma2) == (10.0, 1e-2)
    assert (cfg.inner.kind, cfg.inner.max_steps, cfg.inner.tol, cfg.inner.reduce) == (_abi.CTRL_REDUCTION, 100000, 1e-2, 1e-20)
    bad = prm.parse("subsection Elliptic Interface Problem\n subsection AL preconditioner\n"
                    " set gamma fluid = 10\n set gamma solid = 10\n end\nend\n")
    with pytest.raises(ValueError):               # elliptic_interface.cc:880-884
        prm.config_from_prm(bad)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not present (GPU box)")
def test_every_reference_prm_parses():
    n = 0
    for dirpath, _, files in os.walk("/root/reference"):
        for f in files:
            if f.endswith(".prm"):
                path = os.path.join(dirpath, f)
                if os.path.getsize(path) == 0:
                    continue
                tree = prm.parse_file(path)
                if "nitsche" in f:
                    continue     # driver outside the BASELINE configs
                cfg, info = prm.config_from_prm(tree)
                assert cfg.outer.max_steps > 0 and info["driver"]
                n += 1
    assert n >= 15
    # BASELINE cfg 5: parameters_elliptic_interface/elasticity.prm (modified AL, exact W^-1, Lame 2,1 / 20,10)
    cfg, info = prm.config_from_prm(prm.parse_file("/root/reference/parameters_elliptic_interface/elasticity.prm"))
    assert cfg.variant == _abi.AL_ELL_MODIFIED and (cfg.gamma, cfg.gamma2) == (10.0, 1e-2)
    assert cfg.w_inverse == _abi.W_MASS_INV_SQUARED and (cfg.outer.reduce, cfg.outer.tol) == (1e-6, 1e-10)
    assert info["elasticity"] == {"lambda_background": 2.0, "mu_background": 1.0, "lambda_immersed": 20.0,
                                  "mu_immersed": 10.0}
    cfg, info = prm.config_from_prm(prm.parse_file("/root/reference/parameters_stokes_3d.prm"))
    assert info["unsupported"] == [] and cfg.outer.tol == 1e-8 and cfg.inner.tol == 1e-2
