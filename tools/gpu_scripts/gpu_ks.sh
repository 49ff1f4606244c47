cd /root/repo
python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "k_split" 2>&1 | tail -8
python tools/batch_bench.py --batches 4,8,16,32 2>&1 | tail -4
python tools/batch_bench.py --batches 4,8,16 --opts "skinny.ksplit=0" 2>&1 | tail -3 | sed 's/^/   ksplit off: /'
python tools/batch_bench.py --model mistral-7b-v0.3 --batches 8,16 --steps 64 2>&1 | tail -2
python tools/batch_bench.py --model mistral-7b-v0.3 --batches 8,16 --steps 64 --opts "skinny.ksplit=0" 2>&1 | tail -2 | sed 's/^/   ksplit off: /'
