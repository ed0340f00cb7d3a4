set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r2z
python3 -m pytest tests -x -q -m gpu > gpurun_out/r2z/gputests.log 2>&1; rc=$?; tail -3 gpurun_out/r2z/gputests.log
if [ $rc -ne 0 ]; then exit $rc; fi
python3 bench.py > gpurun_out/r2z/bench.log 2>&1 && tail -1 gpurun_out/r2z/bench.log > gpurun_out/r2z/r02_bench_line_c3_fp16x2.json && cut -c1-400 gpurun_out/r2z/r02_bench_line_c3_fp16x2.json
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r2z/smoke.log 2>&1; tail -1 gpurun_out/r2z/smoke.log
CDFO_BENCH_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 2 --batch 4 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-modes > gpurun_out/r2z/bench_2ranks_gloo.log 2>&1; tail -1 gpurun_out/r2z/bench_2ranks_gloo.log | cut -c1-200
