# SQ counters of ONE micro-benchmark command, per kernel (development aid): where do a kernel's wave-cycles go?
#   bash tools/pmc_kernel.sh <tag> <counters...> -- <python tool and args>
TAG=$1; shift
CTR=""
while [ "$1" != "--" ]; do CTR="$CTR $1"; shift; done
shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/pmc -- python3 "$@" > $OUT/run.log 2> $OUT/run.err
echo "rc=$?"
C=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
python3 - "$C" <<'PY'
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(r["Kernel_Name"], r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k[:90])
    for c, v in sorted(d.items()):
        print(f"    {c:32s} {v / n[(k, c)]:16.1f} per launch")
PY
rm -rf $OUT/pmc
