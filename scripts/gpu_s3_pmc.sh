#!/bin/bash
# dev: SQ / LDS counters of one split-bf16 conv layer (scripts/conv_s3_check, no torch)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CASE=${CASE:-"3 128 64 16 128 160 3"}
rocprofv3 -L 2>/dev/null | grep -o -E "\bSQ_[A-Z_]*LDS[A-Z_]*|SQ_WAIT[A-Z_]*|SQ_INSTS_[A-Z_]*|SQ_ACTIVE_INST_[A-Z_]*|SQ_INST_CYCLES[A-Z_]*|SQ_IFETCH[A-Z_]*" | sort -u | tr '\n' ' ' > $O/s3_counter_names.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace -d $O/s3_pmc$i -- $R/scripts/conv_s3_check $CASE > $O/s3_pmc$i.log 2>&1
  for c in $set; do python $R/profiles/summarize_rocprof_pmc.py $O/s3_pmc$i $c 2>&1 | grep -E "^#|k_conv" | head -3; done
  rm -rf $O/s3_pmc$i
done > $O/s3_pmc.txt 2>&1
cat $O/s3_pmc.txt; cat $O/s3_counter_names.txt
