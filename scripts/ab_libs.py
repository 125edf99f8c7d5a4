"""Same-box A/B of library builds: python scripts/ab_libs.py <configs, e.g. 4,2> [tag=path/to/lib.so ...] -- runs bench.py
(--no-cpu-baseline) alternately with the in-tree library ("head") and every named build, twice each."""
import json, subprocess, sys, os
libs = [("head", None)] + [(a.split("=")[0], a.split("=")[1]) for a in sys.argv[2:]]
for c in [int(x) for x in sys.argv[1].split(",")]:
    for rep in range(2):
        for tag, lib in libs:
            env = dict(os.environ)
            if lib: env["SOBER_HIP_LIB"] = os.path.abspath(lib)
            out = subprocess.run([sys.executable, "bench.py", "--config", str(c), "--steps", "10", "--warmup", "3", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(c, tag, "ms/step %.3f" % d["ms_per_step"], "kernel ms/step %.4f" % d["roofline"]["kernel_ms_per_step"], "frac %.3f" % d["roofline"]["frac"], flush=True)
