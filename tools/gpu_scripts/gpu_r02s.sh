set -x
O=gpurun_out/r02s; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -q) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
