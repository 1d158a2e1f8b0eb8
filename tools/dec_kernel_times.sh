# Per-launch times of the frame-decoder kernels over every benchmark configuration (one GPU call).
# usage (through gpurun): bash tools/dec_kernel_times.sh <tag>
set -o pipefail
tag=${1:-deck}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$tag
mkdir -p $o
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $o/prof -o c -- python3 tools/bench_configs.py 3 --no-reference > $o/prof.log 2>&1 || { tail -5 $o/prof.log; exit 1; }
python3 - <<PY
import csv, collections
t = collections.defaultdict(list)
for r in csv.DictReader(open("$o/prof/c_kernel_trace.csv")):
    n = r["Kernel_Name"]
    if "k_dec_frames" in n:
        t[(n.split("(")[0], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(t.items(), key=lambda kv: kv[0][1]):
    v = sorted(v)
    print("%-40s waves %6d  calls %3d  median %9.1f us" % (k[0][-40:], k[1], len(v), v[len(v) // 2]))
PY
