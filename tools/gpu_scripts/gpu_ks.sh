cd /root/repo
python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "k_split" 2>&1 | tail -3
python tools/batch_bench.py --batches 8,16,24,32 2>&1 | tail -4
python tools/batch_bench.py --batches 24,32 --opts "skinny.ksplit=0" 2>&1 | tail -2 | sed 's/^/   ksplit off: /'
python tools/batch_bench.py --model mistral-7b-v0.3 --batches 32 --steps 64 2>&1 | tail -1
python tools/batch_bench.py --model mistral-7b-v0.3 --batches 32 --steps 64 --opts "skinny.ksplit=0" 2>&1 | tail -1 | sed 's/^/   ksplit off: /'
