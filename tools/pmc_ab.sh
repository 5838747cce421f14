#!/bin/bash
# A/B PMC pass of one edge kernel family: usage tools/pmc_ab.sh <what> <tag> [env assignments...]
# (rocprofv3 needs the program itself after --; env is set in this shell before)
what=$1; tag=$2; shift 2
for kv in "$@"; do export "$kv"; done
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD"
rm -rf gpurun_out/pmc_$tag; mkdir -p gpurun_out/pmc_$tag
rocprofv3 --kernel-trace --pmc $P1 -d gpurun_out/pmc_$tag/a -o a -- python3 tools/prof_edge.py $what 4096 3 > gpurun_out/pmc_$tag/a.log 2>&1
rocprofv3 --kernel-trace --pmc $P2 -d gpurun_out/pmc_$tag/b -o b -- python3 tools/prof_edge.py $what 4096 3 > gpurun_out/pmc_$tag/b.log 2>&1
for p in a b; do
  db=$(find gpurun_out/pmc_$tag/$p -name "*.db" | head -1)
  python3 tools/pmc_summary.py $db edge_ > gpurun_out/pmc_$tag/$p.txt 2>&1
done
cat gpurun_out/pmc_$tag/a.txt gpurun_out/pmc_$tag/b.txt > gpurun_out/pmc_$tag.txt
find gpurun_out/pmc_$tag -name "*.db" -delete
