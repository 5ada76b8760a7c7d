import sys, os
sys.path.insert(0, '/root/repo')
order = sys.argv[1]
def maps():
    libs = set()
    for l in open('/proc/self/maps'):
        for k in ('amdhip', 'hsa-runtime', 'rocfft', 'hiprtc'):
            if k in l: libs.add(l.split()[-1])
    return sorted(libs)
import torch
if order == 'torch_first':
    x = torch.zeros(4, device='cuda'); torch.cuda.synchronize()
    print('torch init ok')
print('before lib:', maps())
from concept_amd.mesh import PotentialMesh
print('after lib import:', maps())
try:
    m = PotentialMesh(32, 32.0)
    print('mesh ok')
except Exception as e:
    print('mesh FAILED', e)
x = torch.zeros(4, device='cuda'); torch.cuda.synchronize(); print('torch after ok')
