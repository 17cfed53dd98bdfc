set -u
O=$PWD/gpurun_out/${1:-r5v}; mkdir -p $O; rm -f $O/*
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $O/pytest.log
grep -E "FAILED|passed|failed|ERROR" $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])   # the compact line (last on stdout)
print({k:d[k] for k in ("value","ms_per_step","ms_per_step_cold")}, "reference api", d["config"].get("value_reference_api"), d["parity_in_run"])
print("roofline", d["roofline"]["frac"], d.get("roofline_other"))
f=json.load(open("bench_out/last_full.json"))                          # the full record beside it
c=f["configs"]
print("C3v", c["C3_viterbi"].get("value"), c["C3_viterbi"].get("reference_api", {}).get("viterbi_path_ms_per_batch"))
print("C4", {k:c["C4"].get(k) for k in ("asg_criterion_fwd_bwd_ms","fcc_forward_ms","fcc_backward_ms","decode_ms_per_batch")})
print("C2", c["C2"]["ms_per_batch"], "C1", c["C1"]["ms_per_loss"], "C5", c["C5_shard"]["value"])
print("vector", f["reference_api"]["vector_overloads"]["ms_per_batch"], "built", f["built_lattice_path"]["ms_per_step"])
PY
