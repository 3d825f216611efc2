"""Which elements of the gather pass's outputs are wrong beside bf16 contractions on a second stream (one process)?"""
import os, sys, threading, time, collections
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pwmlp_repeat_under_load as v  # noqa: E402
import bf16_repeat_under_load as g  # noqa: E402

C = int(os.environ.get("VC", "144"))
N = int(os.environ.get("VN", "4096"))
args = v.setup(C=C, N=N, radius=0.14 if N == 4096 else 0.22)
ref = {k: t.clone() for k, t in v.forward_pieces(*args, 0).items()}
torch.cuda.synchronize()
cs = list(g.cases())[5:10]
stop = [False]
sa = torch.cuda.Stream()


def loader():
    torch.cuda.set_device(0)
    with torch.cuda.stream(sa):
        while not stop[0]:
            for name, k, x, W, dy in cs:
                g.one(k, x, W, dy, 1)
            sa.synchronize()


th = threading.Thread(target=loader)
th.start()
time.sleep(1.0)
sb = torch.cuda.Stream()
chan, lane4, sub, qmod, clouds, nbad, mags = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter(), [], []
with torch.cuda.stream(sb):
    for rep in range(int(os.environ.get("REPS", "40"))):
        o = v.forward_pieces(*args, 0)
        sb.synchronize()
        d = (o["sy"].view(torch.int32) != ref["sy"].view(torch.int32)).nonzero().cpu().numpy()
        nbad.append(len(d))
        for b, j, c in d[:4000]:
            chan[int(c)] += 1
            lane4[int(c) // 4] += 1
            sub[int(c) % 4] += 1
            qmod[int(j) % 12] += 1
            clouds[int(b)] += 1
        if len(d) and len(mags) < 12:
            b, j, c = d[0]
            mags.append(((int(b), int(j), int(c)), float(ref["sy"][b, j, c]), float(o["sy"][b, j, c]),
                         [int(x) for x in (o["sy"][b, j].view(torch.int32) != ref["sy"][b, j].view(torch.int32)).nonzero().flatten().cpu().numpy()][:40]))
stop[0] = True
th.join()
print("C=%d N=%d: wrong sy elements per launch: %s" % (C, N, nbad))
print("channel %% 4:", dict(sub))
print("lane (channel // 4):", dict(sorted(lane4.items())))
print("query %% 12 (tile of 12 queries = 4 waves x 3 lane groups at 72 channels per chunk):", dict(sorted(qmod.items())))
print("clouds:", dict(sorted(clouds.items())))
for m in mags:
    print("  e.g.", m)
