# Developer script (GPU box): regenerates the round's profiles/ inputs under gpurun_out/prof/ (separate passes: kernel trace,
# FETCH_SIZE, WRITE_SIZE).  R = round tag of the file names, COMMIT = the commit the snapshot was taken at (the box has no .git):
#   gpurun -- "R=r03 COMMIT=$(git rev-parse --short HEAD) bash tools/profile_run.sh"
set -o pipefail
R=${R:-r05}; COMMIT=${COMMIT:-unknown}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/prof && mkdir -p $O
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-extra-modes --launch eager"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1 && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1 && \
python3 tools/pmc_summary.py $O/fetch $O/write $O/${R}_pmc_traffic_c3_fp16x2.txt --families $O/traffic_${R}.json fp16x2 $COMMIT && \
cp $O/stats/*/*kernel_stats.csv $O/${R}_kernel_stats_c3_fp16x2.csv && \
D="python3 tools/bench_dcn.py --iters 10" && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dstats -- $D > $O/dstats.log 2>&1 && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/dfetch -- $D > $O/dfetch.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/dwrite -- $D > $O/dwrite.log 2>&1 && \
python3 tools/pmc_summary.py $O/dfetch $O/dwrite $O/${R}_pmc_traffic_dcn_fwd.txt --dcn $O/traffic_dcn_${R}.json $COMMIT && \
cp $O/dstats/*/*kernel_stats.csv $O/${R}_kernel_stats_dcn_fwd.csv && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bstats -- python3 tools/bench_dcn.py --backward --iters 5 > $O/bstats.log 2>&1 && \
cp $O/bstats/*/*kernel_stats.csv $O/${R}_kernel_stats_dcn_bwd.csv && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/v7stats -- python3 tools/bench_v7.py --batch 8 --precision fp16x2 --steps 2 > $O/v7stats.log 2>&1 && \
cp $O/v7stats/*/*kernel_stats.csv $O/${R}_kernel_stats_v7_fp16x2.csv && grep "^# CVSR_V7" $O/v7stats.log > $O/${R}_v7_line.txt && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tstats -- python3 tools/bench_train.py > $O/tstats.log 2>&1 && \
cp $O/tstats/*/*kernel_stats.csv $O/${R}_kernel_stats_train.csv && grep "training step" $O/tstats.log | tail -1 > $O/${R}_train_line.txt && \
python3 tools/phase_times.py 2>&1 | grep -v amdgpu | tail -1 > $O/phases.txt && python3 tools/op_table.py > $O/op_table.txt 2>&1 && cat $O/phases.txt && head -5 $O/${R}_pmc_traffic_c3_fp16x2.txt | tail -1 | cut -c1-150
