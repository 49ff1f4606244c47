cd /root/repo
python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "deferred" 2>&1 | tail -5
