set -x
O=gpurun_out/r02c; mkdir -p $O
(time timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_sampler.py tests/test_hip_host_engine.py -m gpu -x -q) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
python tools/sweep.py --prompt 2048 --steps 128 --grid "attn.fold_combine=0,1;lmhead.fuse_finalize=0,1" > $O/sweep_a.log 2>&1; cat $O/sweep_a.log
python tools/sweep.py --prompt 1000 --steps 128 --grid "attn.fold_combine=0,1" > $O/sweep_1000.log 2>&1; cat $O/sweep_1000.log
python tools/sweep.py --prompt 6000 --steps 128 --grid "attn.fold_combine=0,1" > $O/sweep_6000.log 2>&1; cat $O/sweep_6000.log
python tools/sweep.py --model mistral-7b-v0.3 --prompt 2048 --steps 64 --grid "attn.fold_combine=0,1" > $O/sweep_m7b.log 2>&1; cat $O/sweep_m7b.log
python tools/sweep.py --model qwen2.5-0.5b --prompt 2048 --steps 64 --grid "attn.fold_combine=0,1;lmhead.fuse_finalize=0,1" > $O/sweep_q05.log 2>&1; cat $O/sweep_q05.log
python tools/quick_bench.py --prompt 2048 --steps 128 > $O/quick.log 2>&1; cat $O/quick.log
