#!/bin/bash
# one GPU call: the GPU suite, BASELINE C1 through tools/bench_configs.py, and rocprofv3's kernel trace of the same
# command (what one loss puts on the stream).   usage: bash tools/gpu_c1.sh <tag>
tag=${1:-c1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log
tail -n 5 $out/pytest.log
for i in 1 2 3; do timeout 120 python tools/bench_configs.py c1 2>$out/c1_$i.err | tail -n 1 > $out/c1_$i.json; done
python - <<PY
import json
for i in (1, 2, 3):
    try:
        d = json.load(open("$out/c1_%d.json" % i))
        print("c1 run", i, "ms_per_loss", d["ms_per_loss"], "loss", d["loss"])
    except Exception as e:
        print("c1 run", i, "failed", e)
PY
timeout 200 python tools/bench_configs.py c2 2>$out/c2.err | tail -n 1 > $out/c2.json
python - <<PY
import json
try:
    d = json.load(open("$out/c2.json"))
    print("c2", {k: d[k] for k in d if k.startswith("ms") or k.startswith("value")})
except Exception as e:
    print("c2 failed", e)
PY
root=$(pwd)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $root/$out/prof -o c1 -- python $root/tools/bench_configs.py c1 > $root/$out/prof.log 2>&1)
find $out/prof -name "*stats*" | head
f=$(find $out/prof -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && head -n 14 "$f"
