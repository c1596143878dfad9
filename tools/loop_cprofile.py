import cProfile, pstats, sys, os, io
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
sys.argv = ['agent_loop_bench.py', '12000', sys.argv[1] if len(sys.argv) > 1 else 'rainbow']
import agent_loop_bench as b
pr = cProfile.Profile()
pr.enable()
b.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])
