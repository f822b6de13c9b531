#!/bin/bash
# Run the PMC passes of bench.py (one rocprofv3 run per counter group; --pmc is never combined
# with tracing).  Usage: tools/pmc_passes.sh OUTDIR [bench args...]
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$OUT/p$i" -o pmc -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --side-anchors 0 "$@" > "$OUT/p$i.log" 2>&1
  echo "pass $i [$grp] exit $?"
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
FETCH_SIZE
WRITE_SIZE
GROUPS
