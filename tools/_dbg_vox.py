import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import me, build_model
for cfg, bs in (("S100k-yaw", 8), ("S100k-yaw", 1), ("S50k", 4)):
    batch = build_model.synthetic_batch(cfg, bs, device="cuda")
    pts = batch["points"]
    c = pts[:, :4].clone(); c[:, 1:] /= 0.02
    ref = torch.unique(torch.cat([c[:, :1], torch.floor(c[:, 1:])], 1).int(), dim=0)
    sp = me.SparseTensor(coordinates=c, features=pts[:, 4:].clone())
    C = sp.C
    print(cfg, bs, "points", pts.shape[0], "ref", ref.shape[0], "got", C.shape[0], "minmax", ref.min(0).values.tolist(), ref.max(0).values.tolist())
    print("  per batch ref", torch.bincount(ref[:, 0].long()).tolist(), "got", torch.bincount(C[:, 0].long()).tolist())
    got = torch.unique(C.int(), dim=0)
    print("  distinct rows among got", got.shape[0])
    me.MORTON_ROWS = False
    sp2 = me.SparseTensor(coordinates=c, features=pts[:, 4:].clone())
    print("  without morton order:", sp2.C.shape[0])
    me.MORTON_ROWS = True
