import os, time
import sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavenet_vocoder_amd import noise, _lib
N=2048*2048
g=torch.Generator().manual_seed(1)
u=torch.empty(N,dtype=torch.float64)
for _ in range(2):
    t=time.perf_counter(); u.uniform_(generator=g); dt=time.perf_counter()-t
print("uniform_ f64: %.1f ns/draw"%(dt/N*1e9))
out=torch.empty(N)
for th in (1,4,8,16,32):
    t=time.perf_counter(); _lib.lib().wnv_exponential_from_uniform(u.data_ptr(), out.data_ptr(), N, th); dt=time.perf_counter()-t
    print("transform %d threads: %.2f ns/draw"%(th, dt/N*1e9))
t=time.perf_counter(); noise.exponential_draws(out, g); dt=time.perf_counter()-t
print("exponential_draws: %.1f ns/draw"%(dt/N*1e9))
print(os.cpu_count(), torch.get_num_threads(), len(os.sched_getaffinity(0)))
os.system("grep -m1 'model name' /proc/cpuinfo; nproc")
