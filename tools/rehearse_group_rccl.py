"""The group's RCCL transport on a one-GPU box: RTHIP_GROUP_TRANSPORT=rccl makes a group of ONE device run its exchange
through RCCL (ncclCommInitAll of one communicator, a grouped ncclSend / ncclRecv to itself for the tiles, ncclReduce for
the sample sums); the frames must be the single context's.  Loading the library, the signatures and the call sequence
are what this covers; more than one communicator needs more than one GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RTHIP_GROUP_TRANSPORT"] = "rccl"
import numpy as np
import torch

from raytracer_amd import scenes, tracer as T

verts, indices = scenes.bunny_class_mesh(3)
w, cam, sky = scenes.cornell(extra_mesh=(verts, indices))
settings = T.RT_TracerSettings(max_bounces=6, sky=sky)
one = T.rt_make_tracer(settings, device_id=0)
T.rt_tracer_build_blas(one, w)
T.rt_tracer_build_tlas(one, w)
ok = True
for shard in ("tiles", "samples"):
    g = T.RT_TracerGroup((0,), settings, shard)
    ok = ok and "RCCL" in g.transport()
    g.build_blas(w)
    g.build_tlas(w)
    for wd, ht in ((200, 150), (65, 129)):
        c = cam(wd, ht, 2)
        want = np.zeros((ht, wd, 3), dtype=np.float32)
        T.rt_tracer_cast(one, c, want, wd, ht, seed=11)
        dev = torch.full((ht, wd, 3), -1.0, dtype=torch.float32, device="cuda:0")
        for _ in range(2):
            g.cast_device(c, dev.data_ptr(), wd, ht, seed=11)
        torch.cuda.synchronize()
        same = dev.cpu().numpy().tobytes() == want.tobytes()
        print(shard, wd, ht, "same bits" if same else "DIFFERENT", "|", g.transport())
        ok = ok and same
    g.close()
T.rt_tracer_cleanup(one)
try:  # two contexts on one GPU cannot have a communicator each
    T.RT_TracerGroup((0, 0), settings)
    ok = False
except T.RTHipError:
    pass
print("group rccl rehearsal:", "ok" if ok else "WRONG")
sys.exit(0 if ok else 1)
