import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
if order == 'torch_first':
    import torch
    print('torch first: cuda avail', torch.cuda.is_available(), torch.cuda.device_count())
    t = torch.zeros(4, device='cuda'); print(t.sum().item())
    from ba_amd import hipapi
    e = hipapi.Engine(1, 6); print('engine ok'); import numpy as np; print(e.select_kth(np.arange(10.0), 5))
else:
    from ba_amd import hipapi
    import numpy as np
    e = hipapi.Engine(1, 6); print('engine ok', e.select_kth(np.arange(10.0), 5))
    import torch
    print('engine first: cuda avail', torch.cuda.is_available(), torch.cuda.device_count())
os.system("grep -E 'libamdhip64|libhsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
