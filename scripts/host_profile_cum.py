"""cProfile of recombination steps, cumulative host time per call of the package's own functions and of the torch
built-ins they call (python scripts/host_profile_cum.py [cfg])."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--config", sys.argv[1] if len(sys.argv) > 1 else "2", "--steps", "40", "--warmup", "5", "--no-cpu-baseline"]
import bench
pr = cProfile.Profile()
import warnings
warnings.simplefilter("ignore")
pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr)
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    if nc >= 40 and ("sober_amd" in fn or "built-in" in fn or "method" in name):
        rows.append((ct / nc * 1e6, tt / nc * 1e6, nc, os.path.basename(fn), line, name))
rows.sort(reverse=True)
print("cum us/call  own us/call  calls  where")
for r in rows[:60]:
    print("%10.1f %10.1f %6d  %s:%d %s" % r)
