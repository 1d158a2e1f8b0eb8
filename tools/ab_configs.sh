#!/bin/bash
# tools/ab_configs.sh <config> <lib.so>...: encode kernel times of bench.py --config for the default build and each variant
cfg=$1; shift
mkdir -p gpurun_out/ab
run() {
	timeout -k 10 200 python3 bench.py --config $cfg --no-cpu-baseline --no-stream-api --steps 10 --warmup 2 > gpurun_out/ab/$1.json 2> gpurun_out/ab/$1.err || { tail -3 gpurun_out/ab/$1.err; return; }
	python3 - "$1" <<'PY'
import json, sys
r = json.loads(open("gpurun_out/ab/%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
k = r["roofline"]["kernel_ms"]
print("%-16s encode %.4f ms  %s" % (sys.argv[1], r["ms_per_step"], " ".join("%s=%.4f" % (n, v) for n, v in k.items() if v > 0.02)))
PY
}
run default
for lib in "$@"; do
	FLAC_AMD_LIB=$PWD/$lib run "$(basename "$lib" .so)"
done
