set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1f
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r1f/gpu_tests.log 2>&1 || { tail -20 gpurun_out/r1f/gpu_tests.log; exit 1; }
tail -3 gpurun_out/r1f/gpu_tests.log
timeout -k 10 400 python bench.py > gpurun_out/r1f/bench.json 2> gpurun_out/r1f/bench.err || { tail -5 gpurun_out/r1f/bench.err; exit 1; }
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1f/prof -o r1 -- python3 bench.py --no-cpu-baseline --no-stream-api > gpurun_out/r1f/prof_bench.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --output-format csv -d gpurun_out/r1f/pmcf -o p --pmc FETCH_SIZE -- python3 bench.py --no-cpu-baseline --no-stream-api --steps 2 --warmup 1 > gpurun_out/r1f/pmcf.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --output-format csv -d gpurun_out/r1f/pmcw -o p --pmc WRITE_SIZE -- python3 bench.py --no-cpu-baseline --no-stream-api --steps 2 --warmup 1 > gpurun_out/r1f/pmcw.log 2>&1 || exit 1
find gpurun_out/r1f -name "*.csv" | head
