import sys; sys.path.insert(0, "tools")
from quick_time import run
run(1000000, 300, 0.001, 4000, True)
run(1000000, 300, 0.001, 4000, False)
run(1000000, 300, 0.001, 4000, True, model=1)
run(1000000, 300, 0.001, 4000, True, model=3, bridge=True)
run(40000, 300, 0.001, 4000, True, (1, 0, 64, 64, 0, 0), lockstep=True)
