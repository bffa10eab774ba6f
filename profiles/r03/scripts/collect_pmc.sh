#!/bin/bash
# Regenerates the PMC evidence of the A-SpMV (run on the GPU box from the repo root):
#   bash profiles/r03/scripts/collect_pmc.sh <tag> [extra env assignments ...]
# One rocprofv3 --pmc pass per counter group (no tracing options with --pmc on this pool),
# each over `bench.py --profile-only-spmv 10` (N = 74^3, 12 back-to-back y = A x launches).
# Output: gpurun_out/pmc_<tag>_<group>/ ... counter_collection.csv; summarised by
# profiles/r03/scripts/summarise_pmc.py into profiles/r03/scripts/pmc_<tag>.json.
set -u
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
declare -A groups
groups[sq_time]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
groups[sq_inst]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
groups[fetch]="FETCH_SIZE"
groups[write]="WRITE_SIZE"
groups[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
for g in sq_time sq_inst fetch write tcc; do
  out=gpurun_out/pmc_${tag}_${g}
  rm -rf "$out"
  rocprofv3 --pmc ${groups[$g]} --output-format csv -d "$out" -- python3 bench.py --inner-prec chebyshev --profile-only-spmv 10 \
      > "gpurun_out/pmc_${tag}_${g}.log" 2>&1 || echo "pass $g failed (see gpurun_out/pmc_${tag}_${g}.log)"
done
