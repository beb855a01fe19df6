import ctypes, sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from peaksegdisk_amd import _native, synthetic
from peaksegdisk_amd.grid import ProblemSet
cs, ce, cnt = synthetic.poisson_coverage(100000, seed=1)
w=(ce-cs).astype(np.int32)
pens=synthetic.penalty_grid(64)
for name in sys.argv[1:]:
    lib=_native.declare(ctypes.CDLL(os.path.abspath(name)))
    ps=ProblemSet([(cnt,w)],[(0,float(p)) for p in pens],lib=lib)
    ps.solve()
    f=[ps.solve()[0] for _ in range(2)]
    print(name, "forward ms", min(f), "=> bins/s", 100000*64/(min(f)/1e3))
    ps.close()
