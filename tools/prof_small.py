import cProfile, pstats, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armon_amd as A
p = A.ArmonParameters(test='Sod', N=(1000, 1000), silent=5)
A.armon(p)   # warm
pr = cProfile.Profile(); pr.enable()
s = A.armon(A.ArmonParameters(test='Sod', N=(1000, 1000), silent=5))
pr.disable()
print(s.cycles, s.solve_time)
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
