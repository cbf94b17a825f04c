import sys, time, torch
sys.path.insert(0, '.'); sys.argv=['bench.py']
import bench
torch.cuda.set_device(0)
wl = bench.Workload('c2', torch.device('cuda',0), 0)
for _ in range(3): wl.step()
torch.cuda.synchronize()
import cProfile, pstats
t0=time.perf_counter()
for _ in range(10): wl.step()
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print('host enqueue per step %.3f ms; total per step %.3f ms'%((t1-t0)/10*1e3,(t2-t0)/10*1e3))
pr=cProfile.Profile(); pr.enable()
for _ in range(10): wl.step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
