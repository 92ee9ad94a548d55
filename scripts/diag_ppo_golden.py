"""Diagnostic: engine vs oracle vs reference-golden, per step, per parameter group."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import oracle as O, oracle_ppo as OP
from tianshou_amd import ppo as P
from tests.test_gpu_ppo import _cfg_from_golden, load, dev

tag = sys.argv[1] if len(sys.argv) > 1 else "mujoco"
g = load(f"ppo_{tag}.npz")
E, T, obs_dim, act_dim, batch_size, repeat, n_updates = [int(x) for x in g["dims"]]
cfg = _cfg_from_golden(g)
c = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
ocfg = OP.PPOConfig(gamma=c["gamma"], gae_lambda=c["gae_lambda"], eps_clip=c["eps_clip"], dual_clip=(c["dual_clip"] or None),
                    value_clip=bool(c["value_clip"]), advantage_normalization=bool(c["advantage_normalization"]),
                    recompute_advantage=bool(c["recompute_advantage"]), vf_coef=c["vf_coef"], ent_coef=c["ent_coef"],
                    max_grad_norm=(c["max_grad_norm"] or None), return_scaling=bool(c["return_scaling"]), lr=c["lr"],
                    max_batchsize=int(c["max_batchsize"]))
shapes = OP.param_shapes(obs_dim, act_dim)
bounds, off = {}, 0
for k in OP.PARAM_ORDER:
    n = int(np.prod(shapes[k])); bounds[k] = (off, off + n); off += n

def groups(a, b, label):
    out = []
    for k, (lo, hi) in bounds.items():
        d = np.abs(a[lo:hi] - b[lo:hi]).max(); s = np.abs(b[lo:hi]).max()
        out.append(f"{k}:{d:.1e}/{s:.1e}")
    print(label, " ".join(out))

eng = P.PPOEngine(obs_dim, act_dim, dev(g["flat_params0"]), cfg)
st = OP.PPOState(params=OP.unflatten_params(torch.from_numpy(g["flat_params0"]), obs_dim, act_dim))
for u in range(n_updates):
    pre_ = "" if u == 0 else f"u{u}_"
    bs = O.BufferState(g["buf_offset"], g["buf_last_index"], g["buf_lengths"], g["buf_insertion"],
                       g[pre_ + "rew"], g[pre_ + "terminated"], g[pre_ + "truncated"])
    idx, unf = bs.sample_indices_all(), bs.unfinished_index()
    args = (torch.from_numpy(g[pre_ + "obs"])[idx], torch.from_numpy(g[pre_ + "obs_next"])[idx], torch.from_numpy(g[pre_ + "act"])[idx],
            g[pre_ + "rew"][idx], g[pre_ + "terminated"][idx], g[pre_ + "truncated"][idx], idx, unf)
    pre = OP.preprocess(st, ocfg, *args)
    cutpos = np.nonzero(np.isin(idx, unf))[0]
    b = eng.preprocess(dev(args[0].numpy()), dev(args[1].numpy()), dev(args[2].numpy()), dev(args[3]), dev(args[4]), dev(args[5]), dev(cutpos))
    for k in ("v_s", "returns", "adv", "logp_old"):
        print(u, k, "max|eng-oracle|", float((b[k].cpu() - pre[k]).abs().max()))
    perms = list(g[f"u{u}_perms"])
    n = len(idx)
    offs = P.split_offsets(n, batch_size)
    step = 0
    for r in range(repeat):
        for lo, hi in zip(offs[:-1], offs[1:]):
            rows = perms[r][lo:hi]
            lo_ = OP.update(st, ocfg, {"obs": args[0][rows], "act": args[2][rows]},
                            {k: pre[k][rows] for k in ("v_s", "returns", "adv", "logp_old")}, None, 1,
                            [np.arange(len(rows))], collect_grads=True)
            losses_o, g_o = lo_
            l, g_e = eng._run_steps(b, dev(rows), [0, len(rows)], want_grad=True)
            print(f"u{u} step{step} loss eng {l.cpu().numpy()[0]} oracle {losses_o[0]} ref {g[f'u{u}_losses'][step]}")
            groups(g_e.cpu().numpy(), g_o.numpy(), "   grad |eng-oracle|/scale")
            groups(eng.params.cpu().numpy(), OP.flatten_params(st.params).numpy(), "   param|eng-oracle|/scale")
            step += 1
    groups(eng.params.cpu().numpy(), g[f"u{u}_flat_params"], f"u{u} param|eng-REF|")
    groups(OP.flatten_params(st.params).numpy(), g[f"u{u}_flat_params"], f"u{u} param|oracle-REF|")
