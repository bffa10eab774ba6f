#!/usr/bin/env python3
"""Scrapes what a run of a reference driver (or of this library with alfd_config::log_level >= 1) prints:

  * deal.II's solver-control lines [EXT]: "DEAL:FGMRES::Starting value R", "DEAL:FGMRES::Check k  R",
    "DEAL:FGMRES::Convergence step K value R" / "...::Failure step K value R" -- outer solver first, nested
    prefixes ("DEAL:FGMRES:cg::Convergence step 23 value 0.0093") are the inner solves in call order;
  * the TimerOutput table [EXT]: "| Solve system | calls | wall s | % |" (stokes_immersed_boundary.cc:827 and
    immersed_laplace.cc:504 include AMG setup and factorisations in that section; elliptic_interface.cc:870 only FGMRES);
  * the reference's own summary lines: "Solved in N iterations" (elliptic_interface.cc:967).

    python scrape_deallog.py run.log [--json]
Returns / prints a dict: outer {solver, steps, initial, final, converged, history}, inner {count, steps_total,
steps_per_solve[]}, timers {section: {calls, wall_s}}, dofs when printed."""
import json
import re
import sys

_LINE = re.compile(r"DEAL:([A-Za-z0-9_:]*)::(Starting value|Check|Convergence step|Failure step)\s+(\S+)(?:\s+value)?\s*(\S+)?")
_TIMER = re.compile(r"\|\s*(.+?)\s*\|\s*(\d+)\s*\|\s*([0-9.eE+-]+)s\s*\|")
_TOTAL = re.compile(r"Total wallclock time elapsed since start\s*\|\s*([0-9.eE+-]+)s")


def scrape(text: str) -> dict:
    outer = {"solver": None, "steps": None, "initial": None, "final": None, "converged": None, "history": []}
    inner_steps = []
    timers = {}
    total = None
    for line in text.splitlines():
        m = _LINE.search(line)
        if m:
            prefix, what, a, b = m.groups()
            parts = [p for p in prefix.split(":") if p]
            depth = len(parts)
            name = parts[-1] if parts else ""
            if depth == 1:                                  # the outer Krylov solver
                outer["solver"] = outer["solver"] or name
                if what == "Starting value":
                    outer["initial"] = float(a)
                    outer["history"] = [float(a)]
                elif what == "Check":
                    if b is not None:
                        outer["history"].append(float(b))
                else:
                    outer["steps"], outer["final"] = int(a), float(b)
                    outer["converged"] = what.startswith("Convergence")
                    if not outer["history"] or outer["history"][-1] != float(b):
                        outer["history"].append(float(b))
            elif what in ("Convergence step", "Failure step"):
                inner_steps.append({"solver": name, "steps": int(a), "value": float(b), "converged": what.startswith("Conv")})
            continue
        m = _TIMER.search(line)
        if m and m.group(1) not in ("Section",):
            timers[m.group(1)] = {"calls": int(m.group(2)), "wall_s": float(m.group(3))}
            continue
        m = _TOTAL.search(line)
        if m:
            total = float(m.group(1))
    out = {"outer": outer,
           "inner": {"count": len(inner_steps), "steps_total": sum(s["steps"] for s in inner_steps),
                     "steps_per_solve": [s["steps"] for s in inner_steps],
                     "by_solver": {n: sum(s["steps"] for s in inner_steps if s["solver"] == n)
                                   for n in sorted({s["solver"] for s in inner_steps})}},
           "timers": timers, "total_wall_s": total}
    m = re.search(r"Solved in (\d+) iterations", text)
    if m:
        out["solved_in"] = int(m.group(1))
    if outer["steps"] and "Solve system" in timers:
        out["outer_iterations_per_s"] = outer["steps"] / timers["Solve system"]["wall_s"]
    return out


if __name__ == "__main__":
    res = scrape(open(sys.argv[1]).read())
    if "--json" in sys.argv:
        print(json.dumps(res))
    else:
        o = res["outer"]
        print(f"outer solver {o['solver']}: {o['steps']} steps, {o['initial']} -> {o['final']} ({'converged' if o['converged'] else 'FAILED'})")
        print(f"inner solves: {res['inner']['count']} with {res['inner']['steps_total']} steps in total {res['inner']['by_solver']}")
        for k, v in res["timers"].items():
            print(f"timer {k!r}: {v['calls']} call(s), {v['wall_s']} s")
        if "outer_iterations_per_s" in res:
            print(f"outer iterations per second of 'Solve system': {res['outer_iterations_per_s']:.4f}")
