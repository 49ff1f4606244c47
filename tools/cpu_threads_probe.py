import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    from tinygpt_amd import known_desc, synth
    from tinygpt_amd.ffi import GREEDY
    from oracle.oracle_ffi import OracleModel
    d = known_desc("llama-3.2-1b"); d.max_ctx = 128
    m = OracleModel(d).load_synthetic(1234, 0.02).finalize()
    m.forward(synth.synth_prompt(d.vocab, 16, 1234)[None, :]); m.sample(GREEDY); m.decode(1, GREEDY)
    t0 = time.perf_counter(); m.decode(8, GREEDY); dt = time.perf_counter() - t0
    print(f"OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS')} OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')}: {8 / dt:.2f} tok/s", flush=True)
else:
    print("nproc", os.cpu_count())
    for n, bind in [(16, "close"), (32, "close"), (64, "spread"), (128, "spread"), (256, "false")]:
        env = dict(os.environ, OMP_NUM_THREADS=str(n), OMP_PROC_BIND=bind)
        subprocess.run([sys.executable, __file__, "x"], env=env)
