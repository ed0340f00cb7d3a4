# Developer script (GPU box): SQ counter passes over the three attention products (tools/bench_attn.py: modes 0 / 1 / 2 with three
# passes in both products, modes 20 / 21 / 22 with the single-fp16 second product), one rocprofv3 run per counter group:
#   gpurun -- 'R=r04 bash tools/pmc_attn.sh'   -> gpurun_out/pmcattn/${R}_pmc_counters_seq_attn.txt
set -o pipefail
R=${R:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/pmcattn && mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -- python3 tools/bench_attn.py > $O/g$i.log 2>&1 || { tail -5 $O/g$i.log; exit 1; }
done
{ echo "# rocprofv3 --pmc passes (separate runs per counter group, --kernel-trace only) over tools/bench_attn.py (24 frames of 272x480), mean per launch,"
  echo "# summed over the chip; SQ_* cycle counters in units of 4 cycles per wave (MI355X_MICROARCH.md).  Kernel template arguments: <mode, waves per"
  echo "# workgroup, PV1>: PV1 = true is the single-fp16 probabilities-x-values product the fp16x2 forward uses (round 4), false = three passes."
  python3 tools/pmc_counters.py $O/g1 $O/g2 $O/g3 --match seq_attn_mfma; } > $O/${R}_pmc_counters_seq_attn.txt
cut -c1-70,90-150 $O/${R}_pmc_counters_seq_attn.txt | head -70
