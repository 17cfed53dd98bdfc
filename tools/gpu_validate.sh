set -u
O=$PWD/gpurun_out/${1:-r5v}; mkdir -p $O; rm -f $O/*
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $O/pytest.log
grep -E "FAILED|passed|failed|ERROR" $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","value_reference_api","ms_per_step_reference_api","parity_in_run")})
print("roofline", d["roofline"]["frac"], d["roofline"]["traffic_source"][:40] if d["roofline"].get("traffic_source") else None)
c=d["configs"]
print("C3v", c["C3_viterbi"].get("value"), c["C3_viterbi"].get("reference_api"), c["C3_viterbi"].get("parity_in_run"))
print("C4", {k:c["C4"].get(k) for k in ("asg_criterion_fwd_bwd_ms","fcc_forward_ms","fcc_backward_ms","decode_ms_per_batch","parity_in_run")})
print("C2", c["C2"]["ms_per_batch"], "C1", c["C1"]["ms_per_loss"], "C5", c["C5_shard"]["value"], c["C5_shard"].get("parity_in_run"))
print("vector", d["reference_api"]["vector_overloads"]["ms_per_batch"], "built", d["built_lattice_path"]["ms_per_step"])
PY
