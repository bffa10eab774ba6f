#!/bin/bash
# Second PMC set for the A-SpMV: where the waves wait (levels = in-flight instruction counts integrated over time,
# instruction fetch, scalar unit, queue-full stalls).  Same protocol as collect_pmc.sh.
#   bash profiles/r03/scripts/collect_pmc_stalls.sh <tag>
set -u
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
declare -A groups
groups[levels]="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_CYCLES"
groups[ifetch]="SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES"
groups[scalar]="SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS"
groups[queues]="SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM_RD"
groups[dcache]="SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES SQC_TC_STALL SQC_TC_REQ"
for g in levels ifetch scalar queues dcache; do
  out=gpurun_out/pmc_${tag}_${g}
  rm -rf "$out"
  timeout -k 10 240 rocprofv3 --pmc ${groups[$g]} --output-format csv -d "$out" -- python3 bench.py --inner-prec chebyshev --profile-only-spmv 10 \
      > "gpurun_out/pmc_${tag}_${g}.log" 2>&1 || echo "pass $g failed (see gpurun_out/pmc_${tag}_${g}.log)"
done
