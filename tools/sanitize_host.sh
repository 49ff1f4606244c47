#!/bin/bash
# Runs the CPU test suite with the C++ host engine (loader, tokenizer, regex VM, engine, C ABI) and the C oracle built
# under AddressSanitizer + UBSan, then the CLI once with leak detection, and restores the normal builds.
# Usage: tools/sanitize_host.sh        (from the repo root; ~3 minutes; prints "clean" or the sanitizer reports)
set -e
cd "$(dirname "$0")/.."
H=tinygpt_amd/host
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
CXXF="-std=c++17 -fPIC -fvisibility=hidden -fno-exceptions -pthread"
SRCS="$H/loader.cpp $H/engine.cpp $H/regex.cpp $H/tokenizer.cpp"
restore() { python tinygpt_amd/build.py -f > /dev/null 2>&1; make -B -C oracle > /dev/null 2>&1; }
trap restore EXIT
mkdir -p tests/_build
g++ $SAN $CXXF $SRCS $H/engine_c.cpp -shared -o tinygpt_amd/lib/libtgx_host.so -ldl
g++ $SAN $CXXF $SRCS $H/main.cpp -o tinygpt_amd/lib/tgx_cli -ldl
# the test-hook variants (the only builds that can bind the oracle): what the CPU host tests and the CLI run below use
g++ $SAN $CXXF -DTGXH_TEST_HOOKS $SRCS $H/engine_c.cpp -shared -o tests/_build/libtgx_host_test.so -ldl
g++ $SAN $CXXF -DTGXH_TEST_HOOKS $SRCS $H/main.cpp -o tests/_build/tgx_cli_test -ldl
gcc $SAN -march=x86-64-v3 -ffp-contract=off -fopenmp -fPIC -fvisibility=hidden -fno-math-errno -std=gnu11 -shared -o oracle/liboracle.so oracle/tgx_oracle.c -lm
touch tinygpt_amd/lib/libtgx_host.so tinygpt_amd/lib/tgx_cli tests/_build/libtgx_host_test.so tests/_build/tgx_cli_test oracle/liboracle.so      # newer than their sources: the tests keep them
LOG=$(mktemp -d)
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:log_path=$LOG/asan UBSAN_OPTIONS=print_stacktrace=1:log_path=$LOG/ubsan \
  python -m pytest tests -x -q -m "not gpu" --deselect tests/test_oracle_fullsize.py
python - <<'PY'
import os, sys
sys.path.insert(0, "tests")
from conftest import load_golden
from host_util import write_model_dir
cfg, g = load_golden("llama_tiny")
os.makedirs("/tmp/tgx_san_model", exist_ok=True)
write_model_dir("/tmp/tgx_san_model", dict(cfg, vocab_size=1280), 77, 0.08)
PY
for extra in "" "--stream"; do
  ASAN_OPTIONS=detect_leaks=1:log_path=$LOG/cli OMP_NUM_THREADS=4 tests/_build/tgx_cli_test --model /tmp/tgx_san_model \
    --tokenizer tests/golden/tokenizer/llama3_style --backend-lib oracle/liboracle.so --backend-prefix tgxo_ \
    --max-tokens 8 --temperature 0.8 --top-p 0.9 $extra > /dev/null
done
if ls $LOG/* > /dev/null 2>&1; then cat $LOG/*; echo "sanitizer reports above"; exit 1; else echo "clean"; fi
