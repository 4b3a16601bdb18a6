"""host-side cost of one search step at a host-bound size (DeiT-T bs 8): cProfile of bench.py's step loop"""
import cProfile, pstats, sys, os, io
sys.argv = ['bench.py', '--model', 'deit_tiny', '--batch', '8', '--steps', '30', '--warmup', '3', '--no-cpu-baseline', '--no-prof']
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45)
print(s.getvalue())
