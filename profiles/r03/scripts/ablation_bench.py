"""Times the A-SpMV (N = 74, mesh bricks 16x4x1) with an alternative build of the library: ablation experiments."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))   # repo root
from fictitious_domain_al_preconditioners_amd import problems, solver, _abi
if len(sys.argv) > 1 and sys.argv[1] != "base":
    solver.LIB_PATH = os.path.abspath(sys.argv[1])
N = int(os.environ.get("ABL_N", "74"))
pb = problems.stokes3d_sphere(n_cells=N, immersed_refine=4)
m = pb.mats["A"]
ctx = solver.Context(0)
bricks = tuple(int(v) for v in os.environ.get("ABL_BRICKS", "16,4,1").split(","))
if "ABL_NW" in os.environ:
    ctx.set_tunable("batch_major_waves", int(os.environ["ABL_NW"]))
if "ABL_SHARE" in os.environ:
    ctx.set_tunable("batch_major_share", int(os.environ["ABL_SHARE"]))
if "ABL_ROWS" in os.environ:
    ctx.set_tunable("batch_major_rows", int(os.environ["ABL_ROWS"]))
ctx.set_row_blocks(_abi.A, *problems.brick_row_blocks(pb.params, bricks))
ctx.set_matrix(_abi.A, m)
info = ctx.matrix_info(_abi.A)
best = 1e9
for _ in range(3):
    ms, nbytes = ctx.bench_spmv_format(_abi.A, 30, True)
    best = min(best, ms)
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'base':40s} {best:.4f} ms  fmt {info['batch_major']} bricks {bricks} NW {os.environ.get('ABL_NW', '4')} blocks {info['batch_major_blocks']}", flush=True)
