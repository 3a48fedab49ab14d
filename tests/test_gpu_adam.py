"""splatco_amd.adam.FusedAdam (csrc/adam.hip) against torch.optim.Adam -- the reference's optimizer
(scene/gaussian_model.py:575 `Adam(l, lr=0.0, eps=1e-15)`, stepped at train.py:310-312) -- on the same parameters and
gradients: several groups with their own learning rates, sizes around every vector / workgroup boundary, tensors that are
only 4-byte aligned, more tensors than one launch takes, a learning-rate change between steps, parameters without a
gradient, state_dict interchange, and the optimizer surgery of densification."""
import copy

import pytest
import torch

from splatco_amd.adam import FusedAdam


def _models(dev, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    sizes = [(1,), (3,), (5, 3), (1023,), (1024,), (1025,), (4097,), (300, 71), (7, 15, 9, 9)] + [(33 + i,) for i in range(30)]
    base = [torch.randn(s, generator=g) for s in sizes]
    # a contiguous tensor that starts 4 bytes into an allocation: the scalar path
    off = torch.randn(2050, generator=g)

    def make():
        ps = [torch.nn.Parameter(b.clone().to(dev)) for b in base]
        buf = off.clone().to(dev)
        ps.append(torch.nn.Parameter(buf[1:]))
        assert ps[-1].data_ptr() % 16 == 4 and ps[-1].is_contiguous()
        groups = [{"params": ps[:3], "lr": 1e-2, "name": "a"}, {"params": ps[3:8], "lr": 3e-4, "name": "b"},
                  {"params": ps[8:], "lr": 1e-3, "name": "rest"}]
        return ps, groups
    return make


@pytest.mark.gpu
def test_fused_adam_matches_torch_adam():
    dev = torch.device("cuda:0")
    make = _models(dev)
    pa, ga = make()
    pb, gb = make()
    ours = FusedAdam(ga, lr=0.0, eps=1e-15)
    ref = torch.optim.Adam(gb, lr=0.0, eps=1e-15, foreach=False, fused=False)
    gen = torch.Generator(device=dev).manual_seed(3)
    for it in range(12):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 4 and it % 3 == 0:          # a parameter that gets no gradient in some steps: its step count lags
                a.grad = b.grad = None
                continue
            scale = 10.0 ** ((i % 7) - 4)       # gradients from 1e-4 to 1e2
            g = torch.randn(a.shape, device=dev, generator=gen) * scale
            a.grad, b.grad = g.clone(), g.clone()
        if it == 6:
            for grp in list(ours.param_groups) + list(ref.param_groups):
                grp["lr"] *= 0.5
        ours.step()
        ref.step()
    for i, (a, b) in enumerate(zip(pa, pb)):
        # same formulas, same order of operations up to the compiler's choice of division / sqrt expansions: a few ulp
        # per step on the update, which is lr-sized
        torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-7, msg=lambda m: f"param {i} {tuple(a.shape)}: {m}")
        sa, sb = ours.state[a], ref.state[b]
        assert float(sa["step"]) == float(sb["step"])
        # the first moment is a signed running sum (elements near zero are cancelled): against the tensor's scale
        assert float((sa["exp_avg"] - sb["exp_avg"]).abs().max()) <= 2e-6 * float(sb["exp_avg"].abs().max()), i
        torch.testing.assert_close(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=2e-6, atol=1e-30)
    # bit-reproducible
    pc, gc = make()
    again = FusedAdam(gc, lr=0.0, eps=1e-15)
    gen = torch.Generator(device=dev).manual_seed(3)
    for it in range(12):
        for i, c in enumerate(pc):
            if i == 4 and it % 3 == 0:
                c.grad = None
                continue
            c.grad = torch.randn(c.shape, device=dev, generator=gen) * 10.0 ** ((i % 7) - 4)
        if it == 6:
            for grp in again.param_groups:
                grp["lr"] *= 0.5
        again.step()
    assert all(torch.equal(a, c) for a, c in zip(pa, pc))


@pytest.mark.gpu
def test_fused_adam_state_interchanges_with_torch_adam():
    dev = torch.device("cuda:0")
    make = _models(dev, seed=1)
    pa, ga = make()
    pb, gb = make()
    ours = FusedAdam(ga, lr=0.0, eps=1e-15)
    ref = torch.optim.Adam(gb, lr=0.0, eps=1e-15, foreach=False, fused=False)
    gen = torch.Generator(device=dev).manual_seed(4)

    def grads():
        for a, b in zip(pa, pb):
            g = torch.randn(a.shape, device=dev, generator=gen)
            a.grad, b.grad = g.clone(), g.clone()
    for _ in range(3):
        grads()
        ours.step()
        ref.step()
    # swap the states through state_dict: torch continues from ours, ours from torch's
    sd_ours, sd_ref = copy.deepcopy(ours.state_dict()), copy.deepcopy(ref.state_dict())
    ours.load_state_dict(sd_ref)
    ref.load_state_dict(sd_ours)
    for _ in range(3):
        grads()
        ours.step()
        ref.step()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-7)


@pytest.mark.gpu
def test_densifier_surgery_on_fused_adam_equals_torch_adam():
    """adjust_anchor grows / prunes the per-anchor parameters and their moments inside the optimizer
    (scene/gaussian_model.py:738-818): the same anchors, parameters and moments whichever optimizer holds them."""
    from splatco_amd.densify import AnchorDensifier
    from splatco_amd.synthetic import synthetic_anchor_model
    dev = torch.device("cuda:0")
    out = []
    for kind in ("ours", "torch"):
        pc = synthetic_anchor_model(20_000, 9, dev, plane_size=64)
        groups = [{"params": [getattr(pc, "_" + n)], "lr": 1e-3, "name": n} for n in ("anchor", "offset", "anchor_feat", "scaling")]
        opt = FusedAdam(groups, eps=1e-15) if kind == "ours" else torch.optim.Adam(groups, eps=1e-15, foreach=False, fused=False)
        den = AnchorDensifier(pc, opt, voxel_size=0.01, seed=5)
        gen = torch.Generator(device=dev).manual_seed(6)
        for _ in range(2):
            for grp in groups:
                p = grp["params"][0]
                p.grad = torch.randn(p.shape, device=dev, generator=gen) * 1e-2
            opt.step()
        N, k = pc._anchor.shape[0], pc.n_offsets
        den.offset_gradient_accum[:] = torch.rand(N * k, 1, device=dev, generator=gen)
        den.offset_denom[:] = 60
        den.opacity_accum[:] = torch.rand(N, 1, device=dev, generator=gen) * 2
        den.anchor_demon[:] = 100
        den.adjust_anchor(iteration=100, check_interval=100, grad_threshold=0.012)
        for grp in opt.param_groups:       # and a step on the grown tensors
            p = grp["params"][0]
            p.grad = torch.full_like(p, 1e-3)
        opt.step()
        out.append({grp["name"]: (grp["params"][0].detach().clone(), opt.state[grp["params"][0]]["exp_avg"].clone(),
                                  opt.state[grp["params"][0]]["exp_avg_sq"].clone()) for grp in opt.param_groups})
    assert out[0]["anchor"][0].shape[0] != 20_000
    for name in out[0]:
        for a, b in zip(out[0][name], out[1][name]):
            assert a.shape == b.shape, name
            torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-7)


def test_fused_adam_refuses_what_the_reference_does_not_use():
    p = torch.nn.Parameter(torch.zeros(4))
    for kw in ({"weight_decay": 0.1}, {"amsgrad": True}, {"maximize": True}):
        with pytest.raises(NotImplementedError):
            FusedAdam([p], **kw)
    with pytest.raises(ValueError):
        FusedAdam([p], betas=(1.0, 0.999))
    opt = FusedAdam([p], lr=1e-3)
    p.grad = torch.ones(4)
    with pytest.raises(ValueError, match="no CPU path"):
        opt.step()


def test_scr_adam_step_rejects_bad_arguments_without_a_gpu():
    """The C-ABI entry validates its host-side arguments before anything is launched."""
    import ctypes as C
    from splatco_amd import _C
    lib = _C.lib
    err = lambda: lib.scr_last_error().decode()
    assert lib.scr_adam_step(-1, None, 0.9, 0.999, 1e-15, None) != 0 and "n_tensors" in err()
    assert lib.scr_adam_step(0, None, 0.9, 0.999, 1e-15, None) == 0              # nothing to do
    assert lib.scr_adam_step(1, None, 0.9, 0.999, 1e-15, None) != 0 and "NULL" in err()
    t = (_C.AdamTensor * 1)()
    t[0].numel, t[0].step_size, t[0].bias_correction2_sqrt = 4, 1e-2, 0.03
    assert lib.scr_adam_step(1, t, 0.9, 0.999, 1e-15, None) != 0 and "NULL tensor" in err()
    t[0].param = t[0].grad = t[0].exp_avg = t[0].exp_avg_sq = 16                  # non-null: the checks below come first
    assert lib.scr_adam_step(1, t, 1.0, 0.999, 1e-15, None) != 0 and "beta" in err()
    t[0].step_size = float("inf")                                                 # step 0: lr / (1 - beta1^0) does not exist
    assert lib.scr_adam_step(1, t, 0.9, 0.999, 1e-15, None) != 0 and "bias" in err()


@pytest.mark.gpu
def test_sharded_adam_survives_densification_through_full_state():
    """The documented densification path of adam.ShardedFusedAdam (its `.state` raises): full_state() -> the optimizer surgery
    of adjust_anchor on a torch.optim.Adam built from it (scene/gaussian_model.py:738-818) -> a NEW GradArena +
    ShardedFusedAdam + load_full_state().  Same parameters and moments afterwards as FusedAdam carried through the same
    densification; and a sharded optimizer refuses to step once a parameter's storage was re-pointed away from its flat
    buffer (it would update memory nobody reads)."""
    from splatco_amd.adam import ShardedFusedAdam
    from splatco_amd.densify import AnchorDensifier
    from splatco_amd.multiview import GradArena
    from splatco_amd.synthetic import synthetic_anchor_model
    dev = torch.device("cuda:0")
    names = ("anchor", "offset", "anchor_feat", "scaling")

    def groups_of(pc):
        return [{"params": [getattr(pc, "_" + n)], "lr": 1e-3 * (j + 1), "name": n} for j, n in enumerate(names)]

    def fill_stats(den, pc, gen):
        N, k = pc._anchor.shape[0], pc.n_offsets
        den.offset_gradient_accum[:] = torch.rand(N * k, 1, device=dev, generator=gen)
        den.offset_denom[:] = 60
        den.opacity_accum[:] = torch.rand(N, 1, device=dev, generator=gen) * 2
        den.anchor_demon[:] = 100

    def grads(params, gen, scale):
        return [torch.randn(p.shape, device=dev, generator=gen) * scale for p in params]

    # ---- sharded side
    pc_a = synthetic_anchor_model(20_000, 9, dev, plane_size=64)
    groups_a = groups_of(pc_a)
    params_a = [g["params"][0] for g in groups_a]
    arena = GradArena(params_a, mode="rs_ag")
    opt_a = ShardedFusedAdam(groups_a, arena, eps=1e-15)
    # ---- replicated side
    pc_b = synthetic_anchor_model(20_000, 9, dev, plane_size=64)
    groups_b = groups_of(pc_b)
    params_b = [g["params"][0] for g in groups_b]
    opt_b = FusedAdam(groups_b, eps=1e-15)
    gen = torch.Generator(device=dev).manual_seed(6)
    for _ in range(2):
        for v, pb, g in zip(arena.views, params_b, grads(params_b, gen, 1e-2)):
            v.copy_(g)
            pb.grad = g
        opt_a.step()
        opt_b.step()
    for a, b in zip(params_a, params_b):
        assert torch.equal(a, b)
    # ---- densification: the sharded moments travel through torch.optim.Adam's per-parameter layout
    full = opt_a.full_state()
    tmp = torch.optim.Adam(groups_a, eps=1e-15, foreach=False, fused=False)
    for i, p in enumerate(params_a):
        tmp.state[p] = {"step": full[i]["step"].clone(), "exp_avg": full[i]["exp_avg"].clone(), "exp_avg_sq": full[i]["exp_avg_sq"].clone()}
    den_a, den_b = AnchorDensifier(pc_a, tmp, voxel_size=0.01, seed=5), AnchorDensifier(pc_b, opt_b, voxel_size=0.01, seed=5)
    for den, pc in ((den_a, pc_a), (den_b, pc_b)):
        fill_stats(den, pc, torch.Generator(device=dev).manual_seed(7))
        den.adjust_anchor(iteration=100, check_interval=100, grad_threshold=0.012)
    n_new = pc_a._anchor.shape[0]
    assert n_new != 20_000 and n_new == pc_b._anchor.shape[0]
    # adjust_anchor put NEW Parameter objects into the model; the old arena / optimizer still hold the old ones (a step
    # through train_step.collaborative_step refuses an arena whose parameters are not the model's).  What the optimizer can
    # see by itself is a parameter whose storage was re-pointed away from its flat buffer: that it refuses
    assert params_a[0] is not pc_a._anchor
    params_a[0].data = params_a[0].data.clone()
    with pytest.raises(RuntimeError, match="no longer lives"):
        opt_a.step()
    new_groups = [{"params": list(g["params"]), "lr": g["lr"], "name": g["name"]} for g in tmp.param_groups]
    new_params = [g["params"][0] for g in new_groups]
    assert all(p is getattr(pc_a, "_" + n) for p, n in zip(new_params, names))
    arena.close()
    arena2 = GradArena(new_params, mode="rs_ag")
    opt_a2 = ShardedFusedAdam(new_groups, arena2, eps=1e-15)
    opt_a2.load_full_state({i: tmp.state[p] for i, p in enumerate(new_params)})
    params_b2 = [g["params"][0] for g in opt_b.param_groups]
    gen = torch.Generator(device=dev).manual_seed(8)
    for v, pb, g in zip(arena2.views, params_b2, grads(params_b2, gen, 1e-3)):
        v.copy_(g)
        pb.grad = g
    opt_a2.step()
    opt_b.step()
    after = opt_a2.full_state()
    for i, (a, b) in enumerate(zip(new_params, params_b2)):
        assert a.shape == b.shape and torch.equal(a, b), (names[i], float((a - b).abs().max()))
        assert torch.equal(after[i]["exp_avg"], opt_b.state[b]["exp_avg"]) and torch.equal(after[i]["exp_avg_sq"], opt_b.state[b]["exp_avg_sq"])
        assert float(after[i]["step"]) == float(opt_b.state[b]["step"]) == 3.0
    arena2.close()
