#!/bin/bash
# SQ counters of a side-path kernel (substring match) over tools/bench_side_paths.py at N anchors.  Usage: tools/pmc_side_kernel.sh OUTDIR SUBSTR [N]
set -u
OUT=$1; K=$2; N=${3:-200000}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$OUT/p$i" -o pmc -- python3 tools/bench_side_paths.py $N > "$OUT/p$i.log" 2>&1
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
GROUPS
python3 tools/pmc_summary.py "$OUT" "$K"
rm -rf "$OUT"/p*/
