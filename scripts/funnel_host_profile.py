"""cProfile of the host side of the acquisition step (bench.py --funnel): python scripts/funnel_host_profile.py"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--funnel", "--steps", "40", "--warmup", "3", "--no-cpu-baseline"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(60)
print(s.getvalue()[:14000])
