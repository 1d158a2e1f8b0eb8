# Kernel times and SQ instruction counters of the encode kernels on the bench workload (one GPU call).
# usage (through gpurun): bash tools/search_probe.sh <tag>
set -o pipefail
tag=${1:-probe}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$tag
mkdir -p $o
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o b -- python3 bench.py --no-cpu-baseline --no-stream-api --no-decode > $o/prof.log 2>&1 || { tail -5 $o/prof.log; exit 1; }
timeout -s KILL 120 rocprofv3 --output-format csv -d $o/pmc -o p --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -- python3 bench.py --no-cpu-baseline --no-stream-api --no-decode --steps 2 --warmup 1 > $o/pmc.log 2>&1 || { tail -5 $o/pmc.log; exit 1; }
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$o/prof/b_kernel_stats.csv")))
print("kernel avg us")
for r in rows[:12]:
    print("  %-60s %5s %9.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
c = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$o/pmc/p_counter_collection.csv")):
    c[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in c.items():
    if "search" in k or "pack" in k:
        m = {n: sum(x) / len(x) for n, x in v.items()}
        w = m["SQ_WAVES"]
        print("%s: waves %d  VALU/wave %.0f  SALU/wave %.0f  LDS/wave %.0f  SMEM/wave %.0f  wave quad-cycles %.0f  wait-any/wave %.0f" % (
            k[:40], w, m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_INSTS_LDS"] / w, m["SQ_INSTS_SMEM"] / w, m["SQ_WAVE_CYCLES"] / w, m["SQ_WAIT_INST_ANY"] / w))
PY
