"""What IS a wrong element of the gather pass beside bf16 contractions?  For wrong (cloud, query) rows of sum_k y: the difference to the
clean launch against hypotheses -- a gathered row G stale by d slots, rel stale by d slots, a term dropped / doubled, the centre row."""
import os, sys, threading, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pwmlp_repeat_under_load as v  # noqa: E402
import bf16_repeat_under_load as g  # noqa: E402

C = int(os.environ.get("VC", "64"))
N = 4096
args = v.setup(C=C, N=N, radius=0.14)
xyz, mask, feats, W, gamma, beta, K, radius = args
ref = {k: t.clone() for k, t in v.forward_pieces(*args, 0).items()}
torch.cuda.synchronize()
cs = list(g.cases())[5:10]
stop = [False]
sa = torch.cuda.Stream()


def loader():
    torch.cuda.set_device(0)
    with torch.cuda.stream(sa):
        while not stop[0]:
            for name, k, x, W_, dy in cs:
                g.one(k, x, W_, dy, 1)
            sa.synchronize()


th = threading.Thread(target=loader)
th.start()
time.sleep(1.0)
sb = torch.cuda.Stream()
shown = 0
with torch.cuda.stream(sb):
    for rep in range(200):
        o = v.forward_pieces(*args, 0)
        sb.synchronize()
        bad = (o["sy"].view(torch.int32) != ref["sy"].view(torch.int32))
        if not bool(bad.any()):
            continue
        rows = bad.any(dim=2).nonzero()
        for b, j in rows[:3].tolist():
            ch = bad[b, j].nonzero().flatten()
            idx = ref["idx"][b, j].long()
            ght = ref["ght"][b].double()
            wr = ref["wr"].double()
            sgn = torch.where(gamma < 0, -1.0, 1.0).double()
            rel = (xyz[b, idx].double() - xyz[b, j].double()) / radius  # [K,3]
            G = ght[idx][:, :C]                                          # [K,C]
            H = ght[idx[0], C:]
            y = rel @ wr.t() + H[None] + G                               # [K,C]
            d = (o["sy"][b, j].double() - ref["sy"][b, j].double())[ch]
            best = []
            for k in range(K):
                best.append(("term k=%d dropped" % k, float((-y[k, ch] - d).abs().max())))
                best.append(("term k=%d doubled" % k, float((y[k, ch] - d).abs().max())))
                for dd in (-4, -2, -1, 1, 2, 4):
                    if 0 <= k - dd < K:
                        best.append(("G of slot %d read as slot %d's" % (k, k - dd), float((G[k - dd, ch] - G[k, ch] - d).abs().max())))
                        best.append(("rel of slot %d read as slot %d's" % (k, k - dd), float((((rel[k - dd] - rel[k]) @ wr.t())[ch] - d).abs().max())))
            for a_ in range(3):  # ONE component of rel taken from another slot
                for k in range(K):
                    for k2 in range(K):
                        if k2 != k:
                            best.append(("rel.%s of slot %d read as slot %d's (%+d)" % ("xyz"[a_], k, k2, k2 - k),
                                         float((wr[ch, a_] * (rel[k2, a_] - rel[k, a_]) - d).abs().max())))
            best.sort(key=lambda t: t[1])
            ystar_bad = int((o["ystar"][b, j].view(torch.int32) != ref["ystar"][b, j].view(torch.int32)).sum())
            print("launch %d cloud %d query %d (query %% 4 = %d): %d wrong channels %s; sy got - clean = %s; |d| max %.3g; ystar wrong in %d channels"
                  % (rep, b, j, j % 4, len(ch), ch.tolist()[:8], [round(float(x), 5) for x in d[:6]], float(d.abs().max()), ystar_bad), flush=True)
            print("     best hypotheses (max abs residual over the wrong channels):", [(n_, "%.2e" % e_) for n_, e_ in best[:4]], flush=True)
            print("     |y| scale: %.3g; G scale %.3g" % (float(y[:, ch].abs().mean()), float(G[:, ch].abs().mean())), flush=True)
            shown += 1
        if shown >= 9:
            break
stop[0] = True
th.join()
