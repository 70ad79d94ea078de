#!/bin/bash
# usage: scripts/profile_final.sh   (GPU box, repo root): kernel-trace stats + HBM-side traffic counters of the default bench run
out=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-check"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B --steps 20 --warmup 5 > $out/stats.log 2>&1
echo "stats exit $?"
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B --steps 2 --warmup 1 --no-backward > $out/fetch.log 2>&1
echo "fetch exit $?"
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $B --steps 2 --warmup 1 --no-backward > $out/write.log 2>&1
echo "write exit $?"
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, json, collections, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "final")
def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
fe, wr = counters(out + "/fetch"), counters(out + "/write")
res = {}
for k in fe:
    f = sum(fe[k]["FETCH_SIZE"]) / len(fe[k]["FETCH_SIZE"])
    w = sum(wr[k]["WRITE_SIZE"]) / len(wr[k]["WRITE_SIZE"]) if k in wr else float("nan")
    res[k] = {"FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w, "dispatches": len(fe[k]["FETCH_SIZE"])}
json.dump(res, open(out + "/traffic_raw.json", "w"), indent=1)
for k, v in res.items():
    print("%-50s FETCH %12.0f KB  WRITE %12.0f KB (n=%d)" % (k, v["FETCH_SIZE_KB_raw"], v["WRITE_SIZE_KB"], v["dispatches"]))
PY
f=$(ls $out/stats/*/*kernel_stats.csv | head -1); cut -c1-150 $f | head -12
