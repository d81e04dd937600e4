"""Host -> device bandwidth of the GPU box by NUMA node of the pinned source, and from pageable memory (torch; 1 GiB copies).
   python tools/h2d_probe.py     (round 6: 57.4 GB/s pinned from either socket, 7.9 GB/s pageable; the programs' staged uploads reach 36-48)"""
import os, time, torch
def cores(node):
    s=open('/sys/devices/system/node/node%d/cpulist'%node).read().strip()
    out=[]
    for part in s.split(','):
        a,b=(part.split('-')+[part])[:2]; out+=list(range(int(a),int(b)+1))
    return out
nodes=[n for n in range(8) if os.path.exists('/sys/devices/system/node/node%d'%n)]
print('numa nodes',nodes, 'gpu numa', open('/sys/class/drm/card0/device/numa_node').read().strip() if os.path.exists('/sys/class/drm/card0/device/numa_node') else '?')
for f in sorted(os.listdir('/sys/class/drm')):
    p='/sys/class/drm/%s/device/numa_node'%f
    if os.path.exists(p): print(f, open(p).read().strip(), open('/sys/class/drm/%s/device/current_link_speed'%f).read().strip() if os.path.exists('/sys/class/drm/%s/device/current_link_speed'%f) else '', open('/sys/class/drm/%s/device/current_link_width'%f).read().strip() if os.path.exists('/sys/class/drm/%s/device/current_link_width'%f) else '')
d=torch.empty(1<<30,dtype=torch.uint8,device='cuda')
for node in nodes:
    os.sched_setaffinity(0,set(cores(node)))
    h=torch.empty(1<<30,dtype=torch.uint8).pin_memory(); h.fill_(1)
    torch.cuda.synchronize()
    for rep in range(2):
        t=time.perf_counter(); d.copy_(h,non_blocking=True); torch.cuda.synchronize(); dt=time.perf_counter()-t
    print('node',node,'H2D pinned 1 GiB: %.1f GB/s'%(1.0737/dt))
    del h
os.sched_setaffinity(0,set(range(256)))
p=torch.empty(1<<30,dtype=torch.uint8); p.fill_(1)
t=time.perf_counter(); d.copy_(p); torch.cuda.synchronize(); print('pageable 1 GiB: %.1f GB/s'%(1.0737/(time.perf_counter()-t)))
