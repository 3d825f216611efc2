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
