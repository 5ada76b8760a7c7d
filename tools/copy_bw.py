"""Practical HBM rates on this box through torch's own kernels (a calibration for the
roofline fractions in DESIGN.md): copy (read + write), fill (write only), sum (read only)."""
import torch, time, json
n = 1024*1024*1040  # doubles: one 1024^3 padded mesh, 8.7 GB
a = torch.empty(n, dtype=torch.float64, device='cuda'); b = torch.empty_like(a)
a.fill_(1.0); b.fill_(2.0)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
out = {}
ms = t(lambda: b.copy_(a)); out['copy_GBps'] = 2*8*n/ms/1e6
ms = t(lambda: b.fill_(0.0)); out['fill_GBps'] = 8*n/ms/1e6
ms = t(lambda: a.sum()); out['sum_GBps'] = 8*n/ms/1e6
ms = t(lambda: b.mul_(1.5)); out['inplace_scale_GBps'] = 2*8*n/ms/1e6
print(json.dumps(out))
