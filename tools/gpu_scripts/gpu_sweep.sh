cd /root/repo
python tools/sweep.py --model gpt2 --prompt 64 --steps 128 --grid "qkv.ks=1,2,4;oproj.ks=1,2,4" 2>&1 | tail -9
python tools/sweep.py --model gpt2 --prompt 64 --steps 128 --grid "gateup.ks=1,2,4;down.ks=1,2,4" 2>&1 | tail -9
python tools/sweep.py --model gpt2 --prompt 64 --steps 128 --grid "qkv.bpc=1,2,4;gateup.bpc=1,2,4;down.bpc=1,2,4" 2>&1 | sort -t'>' -k2 -n | head -5
