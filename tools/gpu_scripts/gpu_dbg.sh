cd /root/repo
python tools/dbg_kv2.py 2>&1 | tail -12
echo "--- prefill.mfma=0"; OPTS="prefill.mfma=0" python tools/dbg_kv2.py 2>&1 | tail -8
echo "--- decode gemv"; OPTS="decode.mfma_min_batch=100" python tools/dbg_kv2.py 2>&1 | tail -8
echo "--- both"; OPTS="decode.mfma_min_batch=100;prefill.mfma=0" python tools/dbg_kv2.py 2>&1 | tail -8
