#!/bin/bash
# usage: scripts/profile_final.sh <tag>   (GPU box, repo root): kernel-trace stats of the default bench run + HBM-side traffic counters of
# the forward and backward kernels (separate --pmc passes, as MI355X_MICROARCH.md prescribes) -> gpurun_out/<tag>_final/
tag=${1:-final}
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_final
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-check"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B --steps 20 --warmup 5 > $out/stats.log 2>&1
echo "stats exit $?"
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B --steps 2 --warmup 1 > $out/fetch.log 2>&1
echo "fetch exit $?"
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $B --steps 2 --warmup 1 > $out/write.log 2>&1
echo "write exit $?"
timeout -k 10 240 rocprofv3 --pmc TCC_EA0_ATOMIC_sum --output-format csv -d $out/atomic -- $B --steps 2 --warmup 1 > $out/atomic.log 2>&1
echo "atomic exit $?"
cd $GRAFT_REPO_ROOT
python3 - "$out" <<'PY'
import csv, glob, json, collections, os, sys
out = sys.argv[1]
def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
fe, wr, at = counters(out + "/fetch"), counters(out + "/write"), counters(out + "/atomic")
res = {}
for k in fe:
    f = sum(fe[k]["FETCH_SIZE"]) / len(fe[k]["FETCH_SIZE"])
    w = sum(wr[k]["WRITE_SIZE"]) / len(wr[k]["WRITE_SIZE"]) if k in wr else float("nan")
    a = sum(at[k]["TCC_EA0_ATOMIC_sum"]) / len(at[k]["TCC_EA0_ATOMIC_sum"]) if k in at else float("nan")
    res[k] = {"FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w, "TCC_EA0_ATOMIC": a, "dispatches": len(fe[k]["FETCH_SIZE"])}
json.dump(res, open(out + "/traffic_raw.json", "w"), indent=1)
for k, v in res.items():
    print("%-66s FETCH %12.0f KB  WRITE %12.0f KB  ATOMIC %12.0f (n=%d)" % (k, v["FETCH_SIZE_KB_raw"], v["WRITE_SIZE_KB"], v["TCC_EA0_ATOMIC"], v["dispatches"]))
PY
f=$(ls $out/stats/*/*kernel_stats.csv | head -1); cut -c1-150 $f | head -14
