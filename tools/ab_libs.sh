#!/bin/bash
# A/B timing of experiment builds of the library: tools/ab_libs.sh <lib.so>...  (the default build runs first)
# Prints the encode kernel times of bench.py for each; FLAC_AMD_LIB selects the build.
set -e
mkdir -p gpurun_out/ab
run() {
	timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-stream-api --steps 20 --warmup 3 > gpurun_out/ab/$1.json
	python3 - "$1" <<'PY'
import json, sys
r = json.loads(open("gpurun_out/ab/%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
k = r["roofline"]["kernel_ms"]
print("%-24s encode %.4f ms  %s  decode %.3f ms" % (sys.argv[1], r["ms_per_step"], " ".join("%s=%.4f" % (n, v) for n, v in k.items() if v > 0.02), r["decode"]["ms_per_step"]), "exact" if r["decode"]["bit_exact_vs_input"] and r["decode"]["bad_frames"] == 0 else "DECODE WRONG")
PY
}
run default
for lib in "$@"; do
	FLAC_AMD_LIB=$PWD/$lib run "$(basename "$lib" .so)"
done
