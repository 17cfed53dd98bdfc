#!/bin/bash
set -u
O=gpurun_out/c4; mkdir -p $O
timeout 600 python -m pytest tests/test_lazy_gpu.py -m gpu -x -q -k "dense or pinned or alphabet or asg or maxplus" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python tools/bench_c4.py --steps 3 --no-cpu-baseline > $O/chain.json 2>$O/chain.err; python - <<'PY'
import json
for f in ("gpurun_out/c4/chain.json",):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, {k:d[k] for k in ("decode_ms_per_batch","fcc_forward_ms","fcc_backward_ms","fcc_score0","transitions_grad_sum","asg_criterion_fwd_bwd_ms")}, d.get("roofline",{}).get("frac"))
    except Exception as e: print(f, "ERR", e, open("gpurun_out/c4/chain.err").read()[-500:])
PY
GTNX_NO_CHAIN=1 timeout 300 python tools/bench_c4.py --steps 3 --no-cpu-baseline > $O/steps.json 2>$O/steps.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/c4/steps.json").read().strip().splitlines()[-1])
print("no chain", {k:d[k] for k in ("decode_ms_per_batch","fcc_forward_ms","fcc_backward_ms","fcc_score0","transitions_grad_sum","asg_criterion_fwd_bwd_ms")}, d.get("roofline",{}).get("frac"))
PY
