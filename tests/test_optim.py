"""CPU: FlatSGD's checkpoint surface (closerlook3d_amd/optim.py).  The update itself is a HIP kernel (tests/test_optim_gpu.py);
what is checked here needs no launch: the state dict has torch.optim.SGD's layout (the reference saves and restores
`optimizer.state_dict()`, function/train_modelnet_dist.py:145,160), loads both ways, and a detached parameter is refused."""
import copy
import io

import pytest
import torch

from closerlook3d_amd.optim import FlatSGD


def _net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))


def test_state_dict_round_trip_keeps_the_momentum():
    net = _net()
    opt = FlatSGD(net.parameters(), lr=0.1, momentum=0.9)
    opt._bufs[0].copy_(torch.arange(opt._bufs[0].numel(), dtype=torch.float32))
    sd = opt.state_dict()
    assert [tuple(v["momentum_buffer"].shape) for v in sd["state"].values()] == [(7, 5), (7,), (3, 7), (3,)]
    blob = io.BytesIO()
    torch.save(sd, blob)
    blob.seek(0)
    net2 = _net()
    opt2 = FlatSGD(net2.parameters(), lr=0.1, momentum=0.9)
    opt2.load_state_dict(torch.load(blob))
    assert torch.equal(opt2._bufs[0], opt._bufs[0]) and opt2._steps[0] == 1
    # the loaded state is a view of the flat buffer again (the kernel updates it in place)
    assert opt2.state[list(net2.parameters())[1]]["momentum_buffer"].data_ptr() == opt2._bufs[0][35:].data_ptr()


def test_checkpoints_are_interchangeable_with_torch_sgd():
    lib_net = _net()
    lib = torch.optim.SGD(lib_net.parameters(), lr=0.1, momentum=0.9)
    lib_net(torch.randn(4, 5)).sum().backward()
    lib.step()
    net = _net()
    opt = FlatSGD(net.parameters(), lr=0.1, momentum=0.9)
    opt.load_state_dict(lib.state_dict())
    off = 0
    for p in lib_net.parameters():
        n = p.numel()
        assert torch.equal(opt._bufs[0][off:off + n], lib.state[p]["momentum_buffer"].reshape(-1))
        off += n
    back = torch.optim.SGD(_net().parameters(), lr=0.1, momentum=0.9)
    back.load_state_dict(opt.state_dict())
    assert all(torch.equal(a["momentum_buffer"], b["momentum_buffer"])
               for a, b in zip(back.state.values(), lib.state.values()))


def test_detached_parameters_are_refused():
    net = _net()
    opt = FlatSGD(net.parameters(), lr=0.1, momentum=0.9)
    opt._check_views(0)
    net.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError, match="no longer lives in the flat buffers"):
        opt._check_views(0)
    net2 = copy.deepcopy(_net())
    opt2 = FlatSGD(net2.parameters(), lr=0.1)
    list(net2.parameters())[0].data = torch.zeros(7, 5)
    with pytest.raises(RuntimeError, match="no longer lives in the flat buffers"):
        opt2._check_views(0)
