import sys; sys.path.insert(0, '/root/repo')
import armon_amd as A
for sp in ("Sequential", "SequentialSym", "Strang", "X_only", "Y_only"):
    for proj in ("euler_2nd", "euler"):
        p = A.ArmonParameters(test="Sod", N=(16384, 16384), axis_splitting=sp, projection=proj, maxcycle=12, maxtime=1e9, silent=5)
        s = A.armon(p)
        nsw = {"Sequential": 2, "SequentialSym": 2, "Strang": 3, "X_only": 1, "Y_only": 1}[sp]
        print(f"{sp:14s} {proj:10s} cycles {s.cycles}  {s.giga_cells_per_sec * nsw:7.2f} Gcells/s per sweep", flush=True)
