set -u
O=$PWD/gpurun_out/r4m; mkdir -p $O; rm -f $O/*
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $O/pytest.log
grep -E "FAILED|passed|failed|ERROR" $O/pytest.log
