# Developer script (GPU box): full GPU suite, default bench line, smoke, and the 2-rank rehearsal of bench.py's launcher path
set -o pipefail
R=${R:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/final && mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; rc=$?; tail -3 $O/gputests.log
if [ $rc -ne 0 ]; then exit $rc; fi
python3 bench.py > $O/bench.log 2>&1 && tail -1 $O/bench.log > $O/${R}_bench_line_c3_fp16x2.json && cut -c1-400 $O/${R}_bench_line_c3_fp16x2.json
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
