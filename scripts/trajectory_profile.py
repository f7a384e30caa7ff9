"""GPU box: cProfile of one trajectory run (22-variable chain, 4096 shots)."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcmrf_amd import QCMRF, trajectory, workloads as wl
C = wl.chain(22)
qc = QCMRF(C, wl.theta_halfnorm(wl.dimension(C)))
trajectory.run_trajectories(qc, 512, 3)
cProfile.run("out = trajectory.run_trajectories(qc, 4096, 5)", "/tmp/traj.prof")
print(out[4])
pstats.Stats("/tmp/traj.prof").sort_stats("tottime").print_stats(14)
