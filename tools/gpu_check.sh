# One GPU call: the GPU test suite, then the rates of the other benchmark configurations.
# usage (through gpurun): bash tools/gpu_check.sh <tag>
set -o pipefail
tag=${1:-check}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/$tag/gpu_tests.log 2>&1 || { tail -30 gpurun_out/$tag/gpu_tests.log; exit 1; }
tail -3 gpurun_out/$tag/gpu_tests.log
timeout -k 10 300 python -u tools/bench_configs.py 5 > gpurun_out/$tag/configs.jsonl 2> gpurun_out/$tag/configs.err || { tail -5 gpurun_out/$tag/configs.err; exit 1; }
cat gpurun_out/$tag/configs.jsonl
