cd /root/repo; mkdir -p gpurun_out
cp tinygpt_amd/lib/libtgx_mi355x.so /tmp/vf.so
for lib in vf novf vf novf; do
  if [ $lib = novf ]; then cp tinygpt_amd/lib/libtgx_novf.so tinygpt_amd/lib/libtgx_mi355x.so; else cp /tmp/vf.so tinygpt_amd/lib/libtgx_mi355x.so; fi
  for m in gpt2 qwen3-0.6b llama-3.2-1b; do echo -n "$lib $m: "; python tools/sweep.py --model $m --prompt 256 --steps 128 2>&1 | tail -1; done
done
cp /tmp/vf.so tinygpt_amd/lib/libtgx_mi355x.so
