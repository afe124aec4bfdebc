"""Host logic of gae_dgl_amd.optim.Adam that needs no GPU: checkpoint loading keeps the tensors a captured HIP graph
has baked in (the moments, not only the step counters)."""
import torch

from gae_dgl_amd import optim


def test_load_state_dict_copies_into_the_existing_moment_tensors():
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5))]
    opt = optim.Adam(ps, lr=1e-2)
    for p in ps:                                   # state as a first step() would have created it
        m, v = opt._moments(p)
        m.fill_(1.0); v.fill_(2.0)
    held = [(opt.state[p]["exp_avg"], opt.state[p]["exp_avg_sq"]) for p in ps]
    ptrs = [(m.data_ptr(), v.data_ptr()) for m, v in held]
    # a checkpoint in torch.optim.Adam's layout with other values and a step count
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1e-2)
    for q in ref.param_groups[0]["params"]:
        q.grad = torch.ones_like(q)
    for _ in range(7):
        ref.step()
    sd = ref.state_dict()
    opt.load_state_dict(sd)
    for k, p in enumerate(ps):
        st = opt.state[p]
        assert st["exp_avg"] is held[k][0] and st["exp_avg_sq"] is held[k][1], "moment tensors were re-bound"
        assert (st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()) == ptrs[k]
        assert torch.equal(st["exp_avg"], sd["state"][k]["exp_avg"])
        assert torch.equal(st["exp_avg_sq"], sd["state"][k]["exp_avg_sq"])
        assert "step" not in st
    assert opt.steps_taken() == 7
    # and the round trip keeps torch's layout
    out = opt.state_dict()
    assert float(out["state"][0]["step"]) == 7.0


def test_load_state_dict_into_a_fresh_optimizer():
    ps = [torch.nn.Parameter(torch.randn(3, 2))]
    ref = torch.optim.Adam([torch.nn.Parameter(ps[0].detach().clone())], lr=1e-3)
    ref.param_groups[0]["params"][0].grad = torch.ones(3, 2)
    ref.step(); ref.step()
    opt = optim.Adam(ps, lr=1e-3)
    opt.load_state_dict(ref.state_dict())
    assert opt.steps_taken() == 2
    assert torch.equal(opt.state[ps[0]]["exp_avg"], ref.state_dict()["state"][0]["exp_avg"])
