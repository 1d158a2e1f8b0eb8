# Per-launch times of the frame-decoder kernels over the shapes of tools/dec_mode_probe.py (one GPU call).
# usage (through gpurun): bash tools/dec_probe_kernels.sh <tag> [variants ...]
set -o pipefail
tag=${1:-probek}
shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$tag
mkdir -p $o
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $o/prof -o c -- python3 tools/dec_mode_probe.py "$@" > $o/prof.log 2>&1 || { tail -5 $o/prof.log; exit 1; }
python3 - <<PY
import csv, collections
t = collections.defaultdict(list)
for r in csv.DictReader(open("$o/prof/c_kernel_trace.csv")):
    n = r["Kernel_Name"]
    if "k_dec_" in n:
        t[(n.split("(")[0], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(t.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    v = sorted(v)
    print("%-40s groups %6d  calls %3d  median %9.1f us" % (k[0][-40:], k[1], len(v), v[len(v) // 2]))
PY
