import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["FLOWZ_HIP_DEBUG"] = "1"
import numpy as np, torch
import graphs as G
from zignal_amd import flowz as F
for name, g in (("identity", G.IN(1)), ("df1", G.df1())):
    for P in (1, 2):
        prog = F.compile(F.from_sexpr(g))
        x = torch.arange(64 * 4, dtype=torch.float32, device="cuda").reshape(4, 64, 1).contiguous() + 1
        y, st = prog.run_block(x, variant=F.make_variant(P, 2))
        torch.cuda.synchronize()
        print(name, P, "y[:, :4]=", y[:, :4, 0].cpu().numpy().tolist(), "state", st[:, :2].cpu().numpy().tolist())
