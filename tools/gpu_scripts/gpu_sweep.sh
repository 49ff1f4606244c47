cd /root/repo
python tools/sweep.py --prompt 2048 --steps 128 --grid "qkv.ks=1,2,4;qkv.bpc=2,4,8" 2>&1 | tail -10
python tools/sweep.py --prompt 2048 --steps 128 --grid "oproj.ks=1,2,4;oproj.bpc=2,4,8" 2>&1 | tail -10
python tools/sweep.py --prompt 2048 --steps 128 --grid "down.ks=1,2,4;down.bpc=2,4,8" 2>&1 | tail -10
python tools/sweep.py --prompt 2048 --steps 128 --grid "gateup.bpc=2,3,4,6,8" 2>&1 | tail -6
