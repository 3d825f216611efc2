"""HIP-graph capture of the operators (closerlook3d_amd/fused.py, pt_utils.index_stream): the geometry work an operator
forks onto the index streams must be joined wherever a capture may legally end.

* a captured FORWARD pass with gradients enabled ends fully joined (what torch.cuda.make_graphed_callables needs:
  forward and backward are separate captures) -- hipStreamEndCapture refuses a capture with unjoined work;
* a captured whole step under closerlook3d_amd.whole_step_capture() (forward + backward in one graph, the benches'
  mode) replays to the same output and gradients as eager launches.

(Everything eager that precedes a capture runs on a side stream, as torch's own capture recipe does: a BACKWARD pass
launched on the legacy default stream before `torch.cuda.graph` made hipStreamEndCapture segfault on this stack --
with PosPool as with PointWiseMLP, so nothing of the engine's forked streams is involved.)
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(kind):
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation
    from tests.helpers import default_config
    torch.manual_seed(3)
    B, N, C = 4, 1024, 36
    s = torch.rand(B, N, 3, device="cuda")
    m = torch.ones(B, N, dtype=torch.int32, device="cuda")
    f = torch.randn(B, C, N, device="cuda", requires_grad=True)
    over = {"pointwisemlp__feature_type": "dp_fi_df"} if kind == "pointwisemlp" else {}
    la = LocalAggregation(C, C, 0.15, 24, default_config(kind, over, cl3d_impl="fused")).cuda().train()
    return la, s, m, f


def _warm(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()


@pytest.mark.parametrize("kind", ["pointwisemlp", "pospool", "adaptive_weight", "pseudo_grid"])
def test_forward_only_capture_with_gradients_enabled_ends_joined(kind):
    la, s, m, f = _setup(kind)
    _warm(lambda: la(s, s, m, m, f))
    eager = la(s, s, m, m, f).detach().clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):  # raises hipErrorStreamCaptureUnjoined if an index stream is left dangling
        out = la(s, s, m, m, f)
    g.replay()
    torch.cuda.synchronize()
    assert float((out.detach() - eager).abs().max()) <= 1e-5 * float(eager.abs().max())


@pytest.mark.parametrize("kind", ["pointwisemlp", "pospool"])
def test_whole_step_capture_matches_eager(kind):
    import closerlook3d_amd
    la, s, m, f = _setup(kind)
    probe = torch.randn(4, 36, 1024, device="cuda")
    params = [p for p in la.parameters() if p.requires_grad]

    def step():
        f.grad = None
        for p in params:
            p.grad = None
        out = la(s, s, m, m, f)
        out.backward(probe)
        return out

    _warm(step)
    side = torch.cuda.Stream()  # (eager reference on a side stream as well: nothing of the step on the legacy stream)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out = step()
        want = [out.detach().clone(), f.grad.clone()] + [p.grad.clone() for p in params]
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    del out
    g = torch.cuda.CUDAGraph()
    with closerlook3d_amd.whole_step_capture(), torch.cuda.graph(g):
        out = step()
    g.replay()
    torch.cuda.synchronize()
    got = [out.detach(), f.grad] + [p.grad for p in params]
    for a, b in zip(got, want):
        assert float((a - b).abs().max()) <= 2e-5 * (float(b.abs().max()) + 1e-12)


@pytest.mark.parametrize("kind,precision,accumulate", [("pointwisemlp", "f32", False), ("pointwisemlp", "bf16", True),
                                                       ("pospool", "f32", False), ("adaptive_weight", "f32", True)])
def test_deferred_weight_gradients_are_the_joined_ones(kind, precision, accumulate):
    """closerlook3d_amd.deferred_weight_gradients(): the weight gradients of the contractions stay on the side stream until
    the step's join_weight_gradients() instead of being joined layer by layer -- the same kernels on the same operands, so
    every gradient of a replayed step is bit for bit what the layer-by-layer joins give; with gradients ACCUMULATED into
    an existing .grad (flat gradient buffers: dp.FlatGradients) the join adds on the caller's stream behind its wait.  Two
    bottlenecks (plain, strided), whole-step capture, 20 replays each way."""
    import numpy as np
    import closerlook3d_amd
    from closerlook3d_amd import backbones, fused
    from closerlook3d_amd.backbones import Bottleneck
    from oracle import operators as oo
    from tests.helpers import default_config
    rng = np.random.default_rng(11)
    B, N, K = 4, 1024, 16
    xyz_np, mask_np = oo.make_cloud(rng, B, N, pad_frac=0.1)
    xyz, mask = torch.from_numpy(xyz_np).cuda(), torch.from_numpy(mask_np).cuda()
    f_np = rng.standard_normal((B, 36, N)).astype(np.float32)
    over = {"pointwisemlp__feature_type": "dp_fi_df"} if kind == "pointwisemlp" else {}
    old = backbones._FUSE_MIN_VALUES
    backbones._FUSE_MIN_VALUES = 0
    try:
        results = {}
        for defer in (False, True):
            torch.manual_seed(5)
            cfg = default_config(kind, over, cl3d_precision=precision)
            b1 = Bottleneck(36, 72, 2, 0.12, K, cfg, downsample=False).cuda().train(True)
            b2 = Bottleneck(72, 144, 2, 0.15, K, cfg, downsample=True, sampleDl=0.08, npoint=256).cuda().train(True)
            params = [p for m in (b1, b2) for p in m.parameters() if p.requires_grad]
            feats = torch.from_numpy(f_np).cuda().requires_grad_(True)
            delivered = []

            def step():
                feats.grad = None
                for p in params:
                    if accumulate:
                        p.grad = torch.full_like(p, 0.25)
                    else:
                        p.grad = None
                x1, m1, y1 = b1(xyz, mask, feats)
                _, _, y2 = b2(x1, m1, y1)
                y2.square().mean().backward()
                delivered.append(closerlook3d_amd.join_weight_gradients())

            st = closerlook3d_amd.step_stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                step()
                step()
            torch.cuda.current_stream().wait_stream(st)
            torch.cuda.synchronize()
            assert delivered == [0, 0]  # eager launches never defer
            g = torch.cuda.CUDAGraph()
            with closerlook3d_amd.whole_step_capture(), closerlook3d_amd.deferred_weight_gradients(defer), \
                    torch.cuda.graph(g, stream=st):
                step()
            assert (delivered[-1] > 0) == defer, delivered
            first = None
            for _ in range(20):
                g.replay()
                torch.cuda.synchronize()
                got = [feats.grad.clone()] + [p.grad.clone() for p in params]
                if first is None:
                    first = got
                else:
                    for a, b in zip(got, first):
                        assert torch.equal(a, b)
            results[defer] = first
            assert not fused._DEFERRED
        for a, b in zip(results[True], results[False]):
            assert torch.equal(a, b)
    finally:
        backbones._FUSE_MIN_VALUES = old


_FORGETFUL = r"""
import torch, closerlook3d_amd
from closerlook3d_amd import fused
torch.manual_seed(0)
conv = torch.nn.Conv1d(16, 32, 1, bias=False).cuda()
x = torch.randn(2, 16, 256, device="cuda", requires_grad=True)

def step(join):
    conv.weight.grad = None
    x.grad = None
    y = fused._Conv1x1.apply(x, conv.weight.view(32, 16), fused.PRECISIONS["f32"])
    y.square().mean().backward()
    return closerlook3d_amd.join_weight_gradients() if join else 0

st = closerlook3d_amd.step_stream()
st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    step(True)
    want = conv.weight.grad.clone()
torch.cuda.current_stream().wait_stream(st)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with closerlook3d_amd.whole_step_capture(), closerlook3d_amd.deferred_weight_gradients(), torch.cuda.graph(g, stream=st):
    n = step(True)
g.replay()
torch.cuda.synchronize()
print("JOINED", n, bool(torch.equal(conv.weight.grad, want)), flush=True)
g2 = torch.cuda.CUDAGraph()
try:
    with closerlook3d_amd.whole_step_capture(), closerlook3d_amd.deferred_weight_gradients(), torch.cuda.graph(g2, stream=st):
        step(False)
    print("CAPTURED WITHOUT THE JOIN", flush=True)
except Exception as e:
    print("RAISED", type(e).__name__, len(fused._DEFERRED), flush=True)
import os
os._exit(0)  # (a capture that failed leaves its side stream invalidated: nothing more to do in this process)
"""


def test_a_capture_that_forgets_the_join_fails():
    """A step that defers its weight gradients and never joins them cannot be captured: hipStreamEndCapture refuses the
    unjoined side stream and the error reaches the caller (a child interpreter: the failed capture leaves its streams
    invalidated); the same step WITH the join captures, delivers one gradient and replays to the eager bits."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FORGETFUL], cwd=root, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONPATH=root))
    assert "JOINED 1 True" in r.stdout, r.stdout + r.stderr[-2000:]
    assert "RAISED" in r.stdout and "CAPTURED WITHOUT THE JOIN" not in r.stdout, r.stdout + r.stderr[-2000:]
    assert r.stdout.strip().splitlines()[-1].endswith(" 0"), r.stdout  # nothing left pending after the failure
