"""Same-box A/B of an environment switch: python scripts/ab_env.py <configs> VAR=value  -- bench.py (--no-cpu-baseline)
alternately without and with the variable, twice each."""
import json, subprocess, sys, os
var, val = sys.argv[2].split("=")
for c in [int(x) for x in sys.argv[1].split(",")]:
    for rep in range(2):
        for tag, on in (("default", False), (sys.argv[2], True)):
            env = dict(os.environ)
            if on: env[var] = val
            out = subprocess.run([sys.executable, "bench.py", "--config", str(c), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-sweep", "--no-others"], env=env, capture_output=True, text=True)
            lines = out.stdout.strip().splitlines()
            if not lines:
                print(c, tag, "FAILED", out.stderr[-400:]); continue
            d = json.loads(lines[-1])
            print(c, tag, "ms/step %.3f" % d["ms_per_step"], "level kernel ms/step %.4f" % d["roofline"]["kernel_ms_per_step"],
                  "launches/step %.1f" % d["roofline"]["launches_per_step"], "parity", json.dumps(d["parity"])[:160], flush=True)
