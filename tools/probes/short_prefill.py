import subprocess, sys
for S in (24, 48, 64, 100, 128):
    out = subprocess.run([sys.executable, "tools/prefill_bench.py", "--seq", str(S), "--reps", "8"], capture_output=True, text=True).stdout.strip().splitlines()
    ms = sorted(float(l.split(": ")[1].split(" ms")[0]) for l in out if "prefill S=" in l)
    print(f"S={S}: best {ms[0]:.3f} median {ms[len(ms)//2]:.3f} ms", flush=True)
