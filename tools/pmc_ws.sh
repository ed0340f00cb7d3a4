# Developer script (GPU box): SQ counter passes over the three forms of the Block_.body[0] kernel (tools/bench_ws.py 0), one rocprofv3
# run per counter group and form:   gpurun -- 'bash tools/pmc_ws.sh'   -> gpurun_out/pmcws/counters_{old,ring32,mfma16}.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/pmcws && mkdir -p $O
for form in old ring32 mfma16; do
  case $form in old) export CDFO_WS_RING=0; unset CDFO_WS_MFMA16;; ring32) unset CDFO_WS_RING; export CDFO_WS_MFMA16=0;; mfma16) unset CDFO_WS_RING; unset CDFO_WS_MFMA16;; esac
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/${form}_$i -- python3 tools/bench_ws.py 0 > $O/${form}_$i.log 2>&1 || { tail -5 $O/${form}_$i.log; exit 1; }
  done
  python3 tools/pmc_counters.py $O/${form}_1 $O/${form}_2 $O/${form}_3 --match conv3x3_c64_ws > $O/counters_$form.txt
  echo "== $form"; cut -c1-60,90-160 $O/counters_$form.txt
done
