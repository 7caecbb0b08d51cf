#!/bin/bash
# issue / wait counters of the label kernels (two passes, counters only beside --kernel-trace)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_IFETCH SQ_INSTS_VALU_ADD_F64 SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rm -rf /tmp/pl_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pl_$i -o p -- python tools/bench_labels.py 1024 24 2 > /dev/null 2>/tmp/pl_$i.err || tail -3 /tmp/pl_$i.err
  python tools/rocpd_summary.py $(find /tmp/pl_$i -name '*.db' | head -1) | grep -E "${PMC_KERNEL:-k_label_cover}" | cut -c1-100
done
