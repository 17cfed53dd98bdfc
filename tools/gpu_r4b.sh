set -u
O=$PWD/gpurun_out/r4b; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $O/pytest.log
GTNX_REGION_NO_SLICE_PATH=1 timeout 900 python -m pytest tests/test_batch_gpu.py -q -m gpu 2>&1 | tail -5 > $O/pytest_noslice.log
cat $O/pytest.log | grep -E "FAILED|passed|failed|ERROR"; cat $O/pytest_noslice.log
