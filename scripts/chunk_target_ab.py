import json, subprocess, sys, os
res = {}
for c in (2, 4):
    for rep in range(2):
        for tag, lib in (("1024", None), ("512", "sober_amd/csrc/build/libsober_hip_t512.so"), ("256", "sober_amd/csrc/build/libsober_hip_t256.so")):
            env = dict(os.environ)
            if lib: env["SOBER_HIP_LIB"] = os.path.abspath(lib)
            out = subprocess.run([sys.executable, "bench.py", "--config", str(c), "--steps", "10", "--warmup", "3", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(c, tag, "ms/step %.3f" % d["ms_per_step"], "kernel ms/step %.4f" % d["roofline"]["kernel_ms_per_step"], "frac %.3f" % d["roofline"]["frac"], flush=True)
