"""bench_init.py (the bench scripts' own parameter initialisers) produces exactly the tensors the product's converters and
the oracle's states expect: same keys, same order, same shapes as the oracle's initialisers -- so the measured legs of
bench_*.py need nothing from oracle/ and the cpu_baseline legs can start from the same weights."""
import ast
import os

import torch

import bench_init as BI
from oracle import oracle_dqn as ODQ
from oracle import oracle_drqn as ORQ
from oracle import oracle_dsac as ODS
from oracle import oracle_distq as OQ
from oracle import oracle_ppo as OP
from oracle import oracle_ppo_cnn as OC
from oracle import oracle_ppo_discrete as OD
from oracle import oracle_rainbow as ORB
from oracle import oracle_redq as OR
from oracle import oracle_sac as OS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def same(mine: dict, ref: dict, order=None):
    assert list(mine) == list(order if order is not None else ref), (list(mine), list(order if order is not None else ref))
    for k in mine:
        assert tuple(mine[k].shape) == tuple(ref[k].shape), k
        assert mine[k].dtype == torch.float32 and mine[k].is_contiguous() and bool(torch.isfinite(mine[k]).all())


def test_off_policy_mlp_nets():
    a, c1, c2 = OS.init_td3_params(376, 17, 0, twin=True)
    ma, m1, m2 = BI.td3_nets(376, 17, 0, twin=True)
    same(ma, a, OS.DET_ACTOR_ORDER), same(m1, c1, OS.CRITIC_ORDER), same(m2, c2, OS.CRITIC_ORDER)
    assert BI.td3_nets(376, 17, 0, twin=False)[2] is None and not torch.equal(m1["w1"], m2["w1"])
    a, c1, c2 = OS.init_sac_params(376, 17, 0)
    ma, m1, m2 = BI.sac_nets(376, 17, 0)
    same(ma, a, OS.ACTOR_ORDER), same(m1, c1, OS.CRITIC_ORDER), same(m2, c2, OS.CRITIC_ORDER)
    a, ens = OR.init_params(376, 17, 10, 0)
    same(BI.sac_actor(376, 17), a, OS.ACTOR_ORDER), same(BI.redq_ensemble(376, 17, 10), ens, OR.CRITIC_ORDER)
    nets = ODS.init_params(128, 18, 256, 0)
    mine = BI.dsac_nets(128, 18, 256)
    assert len(mine) == len(nets) == 3
    for m, r in zip(mine, nets):
        same(m, r, ODS.NET_ORDER)


def test_on_policy_nets():
    same(BI.ppo_nets(17, 6), OP.init_params(17, 6, seed=0), OP.PARAM_ORDER)
    same(BI.ppo_discrete_net(4, 64, 2), OD.init_params(4, 64, 2, 0), OD.PARAM_ORDER)
    same(BI.cnn_actor_critic(4, 84, 84, 6), OC.init_params(4, 84, 84, 6, 0), OC.PARAM_ORDER)


def test_atari_q_nets():
    same(BI.dqnet(4, 84, 84, 6), ODQ.init_params(4, 84, 84, 6, 0), ODQ.PARAM_ORDER)
    same(BI.dqnet(4, 84, 84, 6 * 200), OQ.init_params(4, 84, 84, 6, 200, 0), ODQ.PARAM_ORDER)
    same(BI.dqnet(2, 44, 36, 3), ODQ.init_params(2, 44, 36, 3, 0), ODQ.PARAM_ORDER)
    p, n0 = ORB.init_params(4, 84, 84, 6, 51, 0)
    mp, mn = BI.rainbow_net(4, 84, 84, 6, 51)
    same(mp, p, ORB.PARAM_ORDER)
    same(mn, n0, [f"{L}.{t}" for L in ORB.NOISY for t in ("eps_p", "eps_q")])


def test_recurrent_net():
    p = BI.recurrent_net(4, 128, 2, 2)
    shapes = ORQ.param_shapes(4, 128, 2, 2)
    assert list(p) == ORQ.param_keys(2) and all(tuple(p[k].shape) == shapes[k] for k in p)


def test_measured_legs_do_not_touch_the_oracle():
    """In every bench script, anything under oracle/ is imported only inside a baseline function (`cpu_baseline*`,
    `*_baseline`) or an `if with_cpu:` block -- never at module level or on the measured path."""
    for name in ("bench.py", "bench_dqn.py", "bench_sac.py", "bench_ppo_cnn.py", "bench_next.py", "bench_init.py"):
        tree = ast.parse(open(os.path.join(ROOT, name)).read())

        def visit(node, allowed):
            for child in ast.iter_child_nodes(node):
                ok = allowed
                if isinstance(child, ast.FunctionDef) and (child.name.startswith("cpu_baseline") or child.name.endswith("_baseline")):
                    ok = True                     # baseline legs: the CPU port and the ROCm-eager port of the reference path
                if isinstance(child, ast.If) and isinstance(child.test, ast.Name) and child.test.id == "with_cpu":
                    ok = True
                if isinstance(child, ast.ImportFrom) and (child.module or "").split(".")[0] == "oracle":
                    assert allowed, f"{name}:{child.lineno}: oracle import on the measured path"
                if isinstance(child, ast.Import):
                    assert allowed or all(a.name.split(".")[0] != "oracle" for a in child.names), f"{name}:{child.lineno}"
                visit(child, ok)

        visit(tree, False)


def test_bench_scripts_have_no_undefined_names():
    """A static check standing in for running the bench scripts here (they need a GPU): every name a function reads is a
    local, an enclosing-scope variable, a module global or a builtin."""
    import builtins
    import symtable

    for name in ("bench.py", "bench_dqn.py", "bench_sac.py", "bench_ppo_cnn.py", "bench_next.py", "bench_init.py"):
        src = open(os.path.join(ROOT, name)).read()
        top = symtable.symtable(src, name, "exec")
        module_names = {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()}

        def walk(tab):
            for sym in tab.get_symbols():
                if sym.is_global() and sym.is_referenced() and not sym.is_assigned():
                    n = sym.get_name()
                    assert n in module_names or hasattr(builtins, n), f"{name}: `{n}` is undefined in {tab.get_name()}()"
            for child in tab.get_children():
                walk(child)

        for child in top.get_children():
            walk(child)
