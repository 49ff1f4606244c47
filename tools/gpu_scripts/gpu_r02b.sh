set -x
O=gpurun_out/r02b; mkdir -p $O
(time timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "folded or long_context or direct_attention or batched_decode or teacher or streaming") > $O/pytest_fold.log 2>&1
tail -5 $O/pytest_fold.log
python tools/sweep.py --prompt 2048 --steps 128 --grid "attn.fold_combine=0,1,0,1" > $O/sweep_fold.log 2>&1; cat $O/sweep_fold.log
python tools/sweep.py --prompt 2048 --steps 128 --grid "attn.fold_combine=1;attn.gmax=1,2,4" > $O/sweep_gmax.log 2>&1; cat $O/sweep_gmax.log
python tools/sweep.py --prompt 300 --steps 128 --grid "attn.fold_combine=0,1;attn.direct_max=0,768" > $O/sweep_300.log 2>&1; cat $O/sweep_300.log
python tools/sweep.py --prompt 6000 --steps 128 --grid "attn.fold_combine=0,1" > $O/sweep_6000.log 2>&1; cat $O/sweep_6000.log
python tools/sweep.py --model mistral-7b-v0.3 --prompt 2048 --steps 64 --grid "attn.fold_combine=0,1" > $O/sweep_m7b.log 2>&1; cat $O/sweep_m7b.log
python tools/quick_bench.py --prompt 2048 --steps 128 > $O/quick.log 2>&1; cat $O/quick.log
