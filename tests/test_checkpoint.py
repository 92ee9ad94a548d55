"""CPU: Adam-state / parameter round trips through the engines' flat layouts (resume fidelity).
Needs only the built library (layout queries are host-side), no GPU and no reference checkout."""
import pytest
import torch


# ------------------------------------------------------------------------------------ checkpoint fidelity (N4)
def _populate(opt, params, seed):
    g = torch.Generator().manual_seed(seed)
    for _ in range(3):
        for p in params:
            p.grad = torch.randn(p.shape, generator=g)
        opt.step()


@pytest.mark.parametrize("family", ["dqn", "sac_actor", "sac_critic", "td3_actor", "ppo_cnn"])
def test_adam_state_round_trip_through_engine_layouts(family):
    """torch.optim.Adam state -> engine flat layout -> back must be the identity (resume fidelity):
    exp_avg / exp_avg_sq are pure permutations (+ zero padding) of the reference's tensors, the step is shared."""
    from tianshou_amd import dqn as D, ppo_cnn as PC, sac as S, td3 as T
    from tianshou_amd.checkpoint import adam_state, store_adam_state

    torch.manual_seed(0)
    if family == "dqn":
        mods = [torch.nn.Conv2d(4, 32, 8, 4), torch.nn.Conv2d(32, 64, 4, 2), torch.nn.Conv2d(64, 64, 3, 1),
                torch.nn.Linear(3136, 512), torch.nn.Linear(512, 6)]
        to_flat = lambda ts: D.flat_from_torch(ts, 4, 84, 84, 6, device="cpu")      # noqa: E731
        from_flat = lambda f: D.flat_to_torch(f, 4, 84, 84, 6)                       # noqa: E731
    elif family == "ppo_cnn":
        mods = [torch.nn.Conv2d(4, 32, 8, 4), torch.nn.Conv2d(32, 64, 4, 2), torch.nn.Conv2d(64, 64, 3, 1),
                torch.nn.Linear(3136, 512), torch.nn.Linear(512, 6), torch.nn.Linear(512, 1)]
        to_flat = lambda ts: PC.flat_from_torch(ts, 4, 84, 84, 6, device="cpu")     # noqa: E731
        from_flat = lambda f: PC.flat_to_torch(f, 4, 84, 84, 6)                      # noqa: E731
    elif family == "sac_actor":
        mods = [torch.nn.Linear(376, 256), torch.nn.Linear(256, 256), torch.nn.Linear(256, 17), torch.nn.Linear(256, 17)]
        to_flat = lambda ts: S.actor_flat_from_torch(ts, 376, 17, device="cpu")     # noqa: E731
        from_flat = lambda f: S.actor_flat_to_torch(f, 376, 17)                      # noqa: E731
    elif family == "sac_critic":
        mods = [torch.nn.Linear(393, 256), torch.nn.Linear(256, 256), torch.nn.Linear(256, 1)]
        to_flat = lambda ts: S.critic_flat_from_torch(ts, 376, 17, device="cpu")    # noqa: E731
        from_flat = lambda f: S.critic_flat_to_torch(f, 376, 17)                     # noqa: E731
    else:
        mods = [torch.nn.Linear(11, 256), torch.nn.Linear(256, 256), torch.nn.Linear(256, 3)]
        to_flat = lambda ts: T.actor_flat_from_torch(ts, 11, 3, device="cpu")       # noqa: E731
        from_flat = lambda f: T.actor_flat_to_torch(f, 11, 3)                        # noqa: E731
    params = [t for m in mods for t in (m.weight, m.bias)]
    opt = torch.optim.Adam(params, lr=1e-3)
    _populate(opt, params, 1)
    ms, vs, step = adam_state(opt, params)
    assert step == 3
    fm, fv, fp = to_flat(ms), to_flat(vs), to_flat([p.detach() for p in params])
    fresh = torch.optim.Adam(params, lr=1e-3)
    store_adam_state(fresh, params, from_flat(fm), from_flat(fv), step)
    for p, back in zip(params, from_flat(fp)):
        assert torch.equal(p.detach(), back)
    for p in params:
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(opt.state[p][k], fresh.state[p][k]), (family, k)
        assert float(fresh.state[p]["step"]) == 3.0
    # and a fresh optimizer reads as zeros / step 0
    ms0, vs0, step0 = adam_state(torch.optim.Adam(params, lr=1e-3), params)
    assert step0 == 0 and all(float(m.abs().sum()) == 0.0 for m in ms0 + vs0)
