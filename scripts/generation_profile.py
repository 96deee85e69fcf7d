import cProfile, pstats, time, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t=time.time()
pr=cProfile.Profile(); pr.enable()
lv,tr,k = bench.build_problem(sys.argv[1] if len(sys.argv)>1 else "cfg4", True)
pr.disable()
print("total", time.time()-t)
pstats.Stats(pr).sort_stats('cumulative').print_stats(38)
