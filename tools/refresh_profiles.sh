# One GPU call that regenerates everything under profiles/ for a round: the GPU tests, the bench line, the other
# configurations' lines, rocprofv3 kernel statistics of both, and the two PMC passes the HBM traffic figures come from.
# usage (through gpurun): bash tools/refresh_profiles.sh r02
set -o pipefail
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$tag
mkdir -p $o
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $o/gpu_tests.log 2>&1 || { tail -20 $o/gpu_tests.log; exit 1; }
tail -3 $o/gpu_tests.log
timeout -k 10 400 python bench.py > $o/bench.json 2> $o/bench.err || { tail -5 $o/bench.err; exit 1; }
timeout -k 10 300 python -u tools/bench_configs.py 5 > $o/configs.jsonl 2> $o/configs.err || { tail -5 $o/configs.err; exit 1; }
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o bench -- python3 bench.py --no-cpu-baseline --no-stream-api > $o/prof_bench.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/profc -o configs -- python3 tools/bench_configs.py 3 --no-reference > $o/prof_configs.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --output-format csv -d $o/pmcf -o p --pmc FETCH_SIZE -- python3 bench.py --no-cpu-baseline --no-stream-api --steps 2 --warmup 1 > $o/pmcf.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --output-format csv -d $o/pmcw -o p --pmc WRITE_SIZE -- python3 bench.py --no-cpu-baseline --no-stream-api --steps 2 --warmup 1 > $o/pmcw.log 2>&1 || exit 1
find $o -name "*.csv" | head -20
