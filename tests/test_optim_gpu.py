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


def test_flat_sgd_resumes_with_its_momentum_and_from_a_torch_sgd_checkpoint(tmp_path):
    """the reference checkpoints `optimizer.state_dict()` and resumes from it (function/train_modelnet_dist.py:
    145,160).  FlatSGD's momentum lives in one flat buffer; its state dict exposes it as torch.optim.SGD's per-parameter
    `momentum_buffer`s, so (a) save -> load -> continue equals an uninterrupted run, (b) a torch.optim.SGD checkpoint
    resumes under FlatSGD and the other way round."""
    from closerlook3d_amd.optim import FlatSGD
    kw = dict(lr=0.05, momentum=0.9, weight_decay=1e-3)
    g = torch.Generator(device="cuda").manual_seed(2)
    xs = [torch.randn(64, 37, device="cuda", generator=g) for _ in range(6)]

    def run(net, opt, batches):
        for x in batches:
            opt.zero_grad()
            net(x).square().mean().backward()
            opt.step()

    straight = _net()
    o_straight = FlatSGD(straight.parameters(), **kw)
    run(straight, o_straight, xs)
    # (a) three steps, checkpoint through a file, three more in a fresh process-like state
    first = _net()
    o_first = FlatSGD(first.parameters(), **kw)
    run(first, o_first, xs[:3])
    torch.save({"model": first.state_dict(), "optimizer": o_first.state_dict()}, tmp_path / "ckpt.pth")
    ck = torch.load(tmp_path / "ckpt.pth")
    assert all("momentum_buffer" in v for v in ck["optimizer"]["state"].values()) and len(ck["optimizer"]["state"]) == 6
    resumed = _net()
    o_resumed = FlatSGD(resumed.parameters(), **kw)
    resumed.load_state_dict(ck["model"])
    o_resumed.load_state_dict(ck["optimizer"])
    run(resumed, o_resumed, xs[3:])
    for pa, pb in zip(straight.parameters(), resumed.parameters()):
        assert torch.equal(pa, pb), float((pa - pb).abs().max())
    # (b) torch.optim.SGD's checkpoint under FlatSGD, and FlatSGD's under torch.optim.SGD
    lib_net = _net()
    o_lib = torch.optim.SGD(lib_net.parameters(), **kw)
    run(lib_net, o_lib, xs[:3])
    cross = _net()
    o_cross = FlatSGD(cross.parameters(), **kw)
    cross.load_state_dict(lib_net.state_dict())
    o_cross.load_state_dict(o_lib.state_dict())
    run(cross, o_cross, xs[3:])
    run(lib_net, o_lib, xs[3:])
    for pa, pb in zip(lib_net.parameters(), cross.parameters()):
        assert torch.allclose(pa, pb, rtol=2e-6, atol=1e-7)
    back = _net()
    o_back = torch.optim.SGD(back.parameters(), **kw)
    o_back.load_state_dict(o_first.state_dict())  # (loads: same keys, same shapes)


def test_flat_sgd_refuses_parameters_that_left_the_flat_buffers():
    from closerlook3d_amd.optim import FlatSGD
    net = _net()
    opt = FlatSGD(net.parameters(), lr=0.05, momentum=0.9)
    net.zero_grad(set_to_none=True)  # the module's own zero_grad detaches .grad from the flat buffer
    net(torch.randn(8, 37, device="cuda")).square().mean().backward()
    with pytest.raises(RuntimeError, match="no longer lives in the flat buffers"):
        opt.step()
