# Developer script (GPU box): SQ counter passes over the ring kernel's two forms (tools/bench_ring.py), one rocprofv3 run per group
#   gpurun -- 'bash tools/pmc_ring.sh'   -> gpurun_out/pmcring/counters_{sparse,dense}.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/pmcring && mkdir -p $O
for form in sparse dense; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/${form}_$i -- python3 tools/bench_ring.py 1 "" $form > $O/${form}_$i.log 2>&1 || { tail -5 $O/${form}_$i.log; exit 1; }
  done
  python3 tools/pmc_counters.py $O/${form}_1 $O/${form}_2 $O/${form}_3 $O/${form}_4 --match conv3x3_ring > $O/counters_$form.txt
  cat $O/counters_$form.txt | cut -c1-40,90-160
done
