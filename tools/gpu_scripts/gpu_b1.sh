cd /root/repo
python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['value'], l['ms_per_step'], l['roofline']['kernel_classes_avg_us'])"
python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['value'], l['ms_per_step'], l['roofline']['kernel_classes_avg_us'])"
