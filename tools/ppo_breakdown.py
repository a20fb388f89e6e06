"""Where a PPO training step's time goes (pendulum swing-up example shape): collection, GAE glue, updates, evaluation."""
import functools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ambersim_amd.rl import ppo
from ambersim_amd.rl.helpers import make_ppo_networks
from ambersim_amd.rl.pendulum.swingup import PendulumSwingupEnv
from ambersim_amd.rl.wrappers import EpisodeAutoReset
dev = torch.device("cuda", 0)
N = 1024
env = PendulumSwingupEnv(num_envs=N, device=dev); w = EpisodeAutoReset(env, 200); w_plain = EpisodeAutoReset(env, 200, fused=False)
nets = make_ppo_networks(3, 1, policy_hidden_layer_sizes=(64,) * 3)
nets.policy_network.to(dev); nets.value_network.to(dev); nets.policy_network(torch.zeros(1, 3, device=dev)); nets.value_network(torch.zeros(1, 3, device=dev))
norm = ppo.RunningStatistics(3, dev); norm.update(torch.randn(100, 3, device=dev))
gen = torch.Generator(device=dev).manual_seed(0)
def timeit(f, k):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k
with torch.no_grad():
    s0 = w.reset(gen)
    for ug in (False, True):
        c = ppo.Collector(w, nets, norm, 10, 0.1, s0, use_graph=ug)
        t = timeit(lambda: c(gen), 20)
        print(f"collector unroll of 10 steps x {N} envs, graph={ug}: {1e3*t:.3f} ms -> {10*N/t/1e6:.2f} M env-steps/s")
    ev = ppo.Evaluator(PendulumSwingupEnv(num_envs=128, device=dev), nets, norm, 200, True, True)
    t = timeit(lambda: ev(1), 3); print(f"evaluation (200 steps x 128 envs, graphs): {1e3*t:.2f} ms")
    # the raw physics call for scale
    from ambersim_amd import mjx
    d = s0.pipeline_state
    t = timeit(lambda: mjx.step(env.sys, d.replace(ctrl=torch.zeros(N, 1, device=dev)), 1), 200); print(f"mjx.step eager (one amx_step_to + shim): {1e6*t:.1f} us")
    t = timeit(lambda: env.step(s0, torch.zeros(N, 1, device=dev)), 100); print(f"env.step eager: {1e6*t:.1f} us")
    t = timeit(lambda: w_plain.step(s0, torch.zeros(N, 1, device=dev)), 100); print(f"wrapped env.step eager, torch path (env + wrapper as torch code): {1e6*t:.1f} us")
    t = timeit(lambda: w.step(s0, torch.zeros(N, 1, device=dev)), 200); print(f"wrapped env.step eager, fused (amx_env_step: physics + one elementwise launch): {1e6*t:.1f} us")
    for name, ww in (("torch path", w_plain), ("fused", w)):
        c = ppo.Collector(ww, nets, norm, 10, 0.1, s0, use_graph=True)
        t = timeit(lambda: c(gen), 50)
        print(f"collector unroll of 10 steps x {N} envs as a graph, {name}: {1e3*t:.3f} ms -> {10*N/t/1e6:.2f} M env-steps/s")
# ---- the update step
import math
obs_size, act_size, n, mb = 3, 1, 163840, 5120
dist = nets.parametric_action_distribution
obs_f, raw_f, logp_f, adv_f, tgt_f = torch.randn(n, 3, device=dev), torch.randn(n, 1, device=dev), torch.randn(n, device=dev), torch.randn(n, device=dev), torch.randn(n, device=dev)
idx = torch.randperm(n, device=dev)[:mb]
opt = torch.optim.Adam(list(nets.policy_network.parameters()) + list(nets.value_network.parameters()), lr=3e-4, capturable=True)
def update_body():
    o = obs_f.index_select(0, idx)
    params = nets.policy_network(o)
    logp = dist.log_prob(params, raw_f.index_select(0, idx))
    ratio = torch.exp(logp - logp_f.index_select(0, idx))
    a = adv_f.index_select(0, idx)
    policy_loss = -torch.minimum(ratio * a, ratio.clamp(0.7, 1.3) * a).mean()
    v = nets.value_network(o).squeeze(-1)
    loss = policy_loss + 0.25 * ((tgt_f.index_select(0, idx) - v) ** 2).mean()
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
t = timeit(update_body, 10); print(f"update eager: {1e3*t:.3f} ms")
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): update_body()
torch.cuda.current_stream().wait_stream(side)
opt.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): update_body()
t = timeit(g.replay, 50); print(f"update graph replay: {1e3*t:.3f} ms")
a_, b_ = torch.randn(5120, 256, device=dev), torch.randn(256, 256, device=dev)
t = timeit(lambda: a_ @ b_, 200); print(f"fp32 GEMM 5120x256x256: {1e6*t:.1f} us = {2*5120*256*256/t/1e12:.1f} TFLOP/s")
with torch.autocast("cuda", dtype=torch.bfloat16):
    t = timeit(lambda: a_ @ b_, 200); print(f"bf16 autocast GEMM: {1e6*t:.1f} us")
