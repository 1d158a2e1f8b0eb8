#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py:
bytes per launch of each pipeline stage = 2 x FETCH_SIZE KB + WRITE_SIZE KB (FETCH_SIZE counts
half of the bytes of coalesced reads on gfx950).  usage: traffic_json.py fetch.csv write.csv out.json"""
import collections
import csv
import json
import sys

STAGES = (("k_analyze", "analyze"), ("k_autoc", "autoc"), ("k_lpc", "lpc"), ("k_search", "search"), ("k_decide", "decide"),
          ("k_pack", "pack"), ("k_scan", "scan"), ("k_compact", "compact"), ("k_dec_sync", "sync"), ("k_dec_frames", "frames"),
          ("k_dec_crc", "crc"), ("k_dec_output", "output"))


def per_launch(path):
    """Mean counter value per dispatch and kernel, and dispatches per kernel."""
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    fetch, write = per_launch(sys.argv[1]), per_launch(sys.argv[2])
    out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python3 bench.py --no-cpu-baseline "
                     "--steps 2 --warmup 1`; bytes = 2 x FETCH_SIZE KB (gfx950 counts half of coalesced reads; k_compact, whose "
                     "reads are known, confirms it) + WRITE_SIZE KB; a stage that runs as several kernels is their sum",
           "kernels": {}}
    for prefix, stage in STAGES:
        f = sum(v for k, (v, _n) in fetch.items() if k.startswith(prefix))
        w = sum(v for k, (v, _n) in write.items() if k.startswith(prefix))
        if f or w:
            out["kernels"][stage] = {"fetch_kb": f, "write_kb": w, "bytes_per_launch": int((2 * f + w) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out["kernels"].items():
        print("%-8s %8.1f MB" % (k, v["bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
