"""bench/reference_cmake: the log scraper on a synthetic deal.II log, and the export-hook injection on the real
reference sources when they are present (authoring container only; nothing is copied, the output goes to tmp)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "bench", "reference_cmake")
sys.path.insert(0, HARNESS)

LOG = """\
DEAL::gamma (Grad-div): 10
DEAL::gamma (AL): 10
DEAL:FGMRES::Starting value 2340.96
DEAL:FGMRES:cg::Starting value 0.953
DEAL:FGMRES:cg::Convergence step 4 value 0.00812
DEAL:FGMRES:cg::Starting value 1.00
DEAL:FGMRES:cg::Convergence step 23 value 0.00934
DEAL:FGMRES::Check 1\t3.70232
DEAL:FGMRES:cg::Convergence step 25 value 0.00990
DEAL:FGMRES::Check 2\t0.0197243
DEAL:FGMRES::Convergence step 3 value 4.6e-09

+---------------------------------------------+------------+------------+
| Total wallclock time elapsed since start    |      81.2s |            |
|                                             |            |            |
| Section                         | no. calls |  wall time | % of total |
+---------------------------------+-----------+------------+------------+
| Assemble Stokes terms           |         1 |      12.4s |        15% |
| Solve system                    |         1 |      61.5s |        76% |
+---------------------------------+-----------+------------+------------+
"""


def test_scraper_reads_solver_controls_and_timer_table():
    import scrape_deallog
    r = scrape_deallog.scrape(LOG)
    assert r["outer"]["solver"] == "FGMRES" and r["outer"]["steps"] == 3 and r["outer"]["converged"]
    assert r["outer"]["initial"] == 2340.96 and r["outer"]["final"] == 4.6e-09
    assert r["outer"]["history"] == [2340.96, 3.70232, 0.0197243, 4.6e-09]
    assert r["inner"]["count"] == 3 and r["inner"]["steps_total"] == 52 and r["inner"]["by_solver"] == {"cg": 52}
    assert r["timers"]["Solve system"] == {"calls": 1, "wall_s": 61.5} and r["total_wall_s"] == 81.2
    assert abs(r["outer_iterations_per_s"] - 3 / 61.5) < 1e-12


def test_scraper_reads_the_librarys_own_log_lines():
    """alfd_config::log_level >= 1 prints the same "DEAL:FGMRES::Convergence step" line."""
    import scrape_deallog
    r = scrape_deallog.scrape("DEAL:FGMRES::Convergence step 9 value 4.686774684566552e-09\n")
    assert r["outer"]["steps"] == 9 and r["outer"]["converged"]


@pytest.mark.skipif(not os.path.exists("/root/reference/stokes_immersed_boundary.cc"), reason="reference sources absent")
@pytest.mark.parametrize("driver,hook", [("stokes_immersed_boundary", "ALFD_EXPORT_STOKES_HOOK"),
                                         ("immersed_laplace", "ALFD_EXPORT_LAPLACE_HOOK")])
def test_export_hook_injection_finds_its_anchor(tmp_path, driver, hook):
    """inject_export.cmake on the real driver: exactly one hook in front of the FGMRES solver object, the #include on
    top, the 3-D switch applied -- and otherwise the file is the reference's, byte for byte."""
    out = tmp_path / f"{driver}.cc"
    p = subprocess.run(["cmake", f"-DIN=/root/reference/{driver}.cc", f"-DOUT={out}", f"-DDRIVER={driver}",
                        "-DSTOKES_3D=ON", "-P", os.path.join(HARNESS, "inject_export.cmake")],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    gen = out.read_text()
    ref = open(f"/root/reference/{driver}.cc").read()
    assert gen.startswith('#include "alfd_export_hook.hpp"\n') and gen.count(hook) == 1
    i = gen.index(hook)
    assert "SolverFGMRES<BlockVector<double>>" in gen[i:i + 200]          # right in front of the solver object
    if driver == "stokes_immersed_boundary":
        assert "const unsigned int dim = 2, spacedim = 3;" in gen
    # removing what was inserted gives back the reference
    back = gen[len('#include "alfd_export_hook.hpp"\n'):]
    j = back.index(hook)
    k = back.index("SolverFGMRES<BlockVector<double>>", j)
    back = back[:j] + back[k:]
    back = back.replace("const unsigned int dim = 2, spacedim = 3;", "const unsigned int dim = 1, spacedim = 2;", 1)
    assert back == ref


def test_hook_header_names_only_abi_symbols_the_library_exports():
    """Every alfd_* function / ALFD_* enumerator the hook header uses exists in include/alfd/alfd.h."""
    import re
    hdr = open(os.path.join(HARNESS, "alfd_export_hook.hpp")).read()
    abi = open(os.path.join(ROOT, "include", "alfd", "alfd.h")).read()
    for name in sorted(set(re.findall(r"\b(ALFD_[A-Z0-9_]+|alfd_[a-z_]+)\b", hdr))):
        if name.startswith("ALFD_EXPORT_") or name in ("alfd_export_hook", "alfd_w", "alfd_cfg", "alfd_ones", "alfd_lumped",
                                                        "alfd_x", "alfd_inv", "alfd_i"):
            continue
        assert re.search(r"\b" + name + r"\b", abi), name
