import sys,time,os; sys.path.insert(0,os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from stark_symphony_amd import records
from oracle import oracle as O
p=records.load_stwo_npz(os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'tests/golden/stwo_trace20.npz'))
print('num_procs', O.num_procs(), 'sched_affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())
try:
    print('cgroup cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e: print('no cpu.max', e)
b=O.StwoBatch([p[i%2] for i in range(2048)])
for th in (1,8,32,64,128,256):
    n=2048 if th>=32 else 64*th
    bb=b if n==2048 else O.StwoBatch([p[i%2] for i in range(n)])
    bb.verify(O.MODE_FIXTURE,th)
    t=time.perf_counter(); bb.verify(O.MODE_FIXTURE,th); dt=time.perf_counter()-t
    print('threads %3d: %8.0f proofs/s  (%.1f per thread)'%(th, n/dt, n/dt/th), flush=True)
