set -x
O=gpurun_out/r02j; mkdir -p $O
(time timeout 1200 python -m pytest tests/test_hip_prefill.py -m gpu -x -q) > $O/pytest.log 2>&1
tail -20 $O/pytest.log
for cfg in "llama-3.2-1b fp32 256" "llama-3.2-1b fp32 2048" "gpt2 fp32 1000" "gpt2 bf16 1000" "gpt2 fp32 256" "llama-3.2-1b bf16 2048" "mistral-7b-v0.3 fp32 256"; do
  set -- $cfg
  python tools/prefill_bench.py --model $1 --dtype $2 --seq $3 >> $O/prefill.log 2>&1
  python tools/prefill_bench.py --model $1 --dtype $2 --seq $3 --opts "prefill.mfma=0" --reps 1 2>&1 | sed 's/^/   decode-kernel passes: /' >> $O/prefill.log
done
cat $O/prefill.log
