import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t=time.time()
import torch
print("torch", torch.__version__, "import %.1fs" % (time.time()-t), flush=True)
print("cuda available", torch.cuda.is_available(), torch.cuda.device_count(), flush=True)
torch.cuda.set_device(0)
x = torch.ones(4, device="cuda:0"); print("torch tensor ok", float(x.sum()), flush=True)
import clipgen as g
from subzero_amd import capi
ctx = capi.Context(0)
ps = g.convex_pairs(2000, seed=3)
r = g.run_gpu(ctx, ps, 1); h = g.run_host(ps, 1)
print("clip after torch: bad =", len(g.compare(h, r, len(ps))), flush=True)
os.system("grep -E 'amdhip|hsa-runtime|rccl' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
ctx.close()
