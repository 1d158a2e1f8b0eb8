#!/usr/bin/env python3
"""Print the kernel times of a bench.py JSON line: tools/show_bench.py <file>..."""
import json
import sys

for path in sys.argv[1:]:
    r = json.loads(open(path).read().strip().splitlines()[-1])
    k = r["roofline"]["kernel_ms"]
    d = r.get("decode", {})
    print("%-28s encode %.4f ms  %s  decode %.3f ms %s" % (
        path.split("/")[-1], r["ms_per_step"], " ".join("%s=%.4f" % (n, v) for n, v in k.items() if v > 0.02),
        d.get("ms_per_step", 0.0), "exact" if d.get("bit_exact_vs_input") and d.get("bad_frames") == 0 else "DECODE WRONG"))
