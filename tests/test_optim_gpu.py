"""FlatSGD (closerlook3d_amd/optim.py, csrc/optim.hip: cl3d_sgd_step) against torch.optim.SGD -- the optimizer of the
reference's training loops (function/train_modelnet_dist.py:137-141) -- over several steps of the same gradients."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.ReLU(), torch.nn.Linear(53, 11), torch.nn.BatchNorm1d(11)).cuda()


@pytest.mark.parametrize("kw", [
    dict(lr=0.05),
    dict(lr=0.05, momentum=0.9),
    dict(lr=0.01, momentum=0.98, weight_decay=1e-3),             # the reference's settings (cfgs/*.yaml)
    dict(lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-2),
    dict(lr=0.05, momentum=0.8, dampening=0.3),
])
def test_flat_sgd_follows_torch_sgd(kw):
    from closerlook3d_amd.optim import FlatSGD
    a, b = _net(), None
    b = copy.deepcopy(a)
    oa = torch.optim.SGD(a.parameters(), **kw)
    ob = FlatSGD(b.parameters(), **kw)
    g = torch.Generator(device="cuda").manual_seed(1)
    for step in range(6):
        x = torch.randn(64, 37, device="cuda", generator=g)
        for net, opt in ((a, oa), (b, ob)):
            opt.zero_grad()
            net(x).square().mean().backward()
            opt.step()
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.allclose(pa, pb, rtol=2e-6, atol=1e-7), (step, float((pa - pb).abs().max()))
    # the parameters are views of the flat buffer, the gradients are left zeroed
    assert all(p.data_ptr() >= ob.flat_params[0].data_ptr() for p in b.parameters())
    assert float(ob.flat_grads[0].abs().max()) == 0.0


def test_flat_sgd_step_in_a_captured_graph_equals_eager_steps():
    from closerlook3d_amd.optim import FlatSGD
    nets = [_net(), _net()]
    opts = [FlatSGD(n.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3) for n in nets]
    x = torch.randn(64, 37, device="cuda")

    def step(i):
        nets[i](x).square().mean().backward()
        opts[i].step()

    for _ in range(3):   # warm-up on a side stream, as torch.cuda.graph asks
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step(0)
        torch.cuda.current_stream().wait_stream(s)
        step(1)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):   # (recorded, not run)
        step(0)
    for _ in range(4):
        graph.replay()
        step(1)
    torch.cuda.synchronize()
    for pa, pb in zip(nets[0].parameters(), nets[1].parameters()):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-6)
