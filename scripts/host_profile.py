"""cProfile of the host side of recombination steps at a BASELINE configuration (python scripts/host_profile.py [cfg]):
which Python functions the host spends its time in between the GPU's kernels."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--config", sys.argv[1] if len(sys.argv) > 1 else "2", "--steps", "30", "--warmup", "3", "--no-cpu-baseline"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:6000])
