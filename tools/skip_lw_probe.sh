# Decode times of bench.py --config C4 / C3 against the frames per wave of the split form's walk (FLAC_AMD_DEC_SKIP_LW).
# usage (through gpurun): bash tools/skip_lw_probe.sh
for cfg in C4 C3; do for v in 8 4 2; do
  FLAC_AMD_DEC_SKIP_LW=$v timeout -k 10 200 python3 bench.py --config $cfg --no-cpu-baseline --no-stream-api --steps 5 --warmup 2 > gpurun_out/sk.json 2> gpurun_out/sk.err || { tail -3 gpurun_out/sk.err; continue; }
  python3 - $cfg $v <<'PY'
import json, sys
r = json.loads(open("gpurun_out/sk.json").read().strip().splitlines()[-1])
d = r["decode"]
print(sys.argv[1], "skip_lw", sys.argv[2], "decode %.3f ms" % d["ms_per_step"], {k: round(v, 3) for k, v in d["roofline"]["kernel_ms"].items()}, d["bit_exact_vs_input"])
PY
done; done
