"""Headline benchmark: collocation-points/sec of one residual+grad evaluation, Burgers 1D (BASELINE.json).

A "step" = zero the gradient buffer, then ONE fused launch that computes r = compute_residual(model, x, t),
sum r^2 and d(mean r^2)/d(theta) for this rank's 49 729 points (weights and points resident in HBM), then —
with more than one rank — one RCCL all-reduce of [flat gradient || loss sum].  Weak scaling: every rank owns
its own 49 729-point batch.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of THIS command
    (profiles/r01_bench.json, made by tools/profile_summary.py): FETCH_SIZE (x2: gfx950 counts 64 B per 128-B
    request) + WRITE_SIZE, both KiB.  None when no profile has been committed."""
    path = os.path.join(ROOT, "profiles", "r01_bench.json")
    try:
        with open(path) as f:
            h = json.load(f).get("hbm_bytes_per_launch")
        return None if not h else float(h["read_x2"] + h["write"])
    except (OSError, ValueError, KeyError):
        return None


def host_threads() -> int:
    """Threads for the CPU leg: this process's CPU share (a 1-GPU box grants 16 cores), never the whole host."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("PINN_CPU_THREADS", "16"))))


def cpu_baseline(model, x_cpu, t_cpu, seconds: float):
    """The oracle (= the reference's CPU op sequence: 2 forwards + chained autograd.grad + backward) on host cores."""
    import oracle as O

    torch.set_num_threads(host_threads())
    pspec = O.PdeSpec(name="burgers", parameters={"nu": 0.01 / math.pi})
    aspec = O.ArchSpec("fourier", hidden_dim=128, num_layers=4)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    N = x_cpu.shape[0]
    for _ in range(2):
        r, L, g = O.residual_loss_and_grad(pspec, aspec, sd, x_cpu, t_cpu)
    best, n, t_end = float("inf"), 0, time.perf_counter() + seconds
    while time.perf_counter() < t_end or n < 3:
        t0 = time.perf_counter()
        r, L, g = O.residual_loss_and_grad(pspec, aspec, sd, x_cpu, t_cpu)
        best = min(best, time.perf_counter() - t0)
        n += 1
    flat = torch.cat([g[k].flatten() for k, _ in model.named_parameters()])
    return {"value": N / best, "unit": "points/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"best of {n} evaluations of the same {N}-point batch (~{seconds:.0f} s), fp32, "
                      f"torch {torch.__version__} CPU"}, float(L), flat


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--points", type=int, default=50000, help="requested collocation points per GPU (uniform -> floor(sqrt)^2)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # PINN_BENCH_REHEARSAL=1: rehearse the N > 1 code path on a ONE-GPU box (all ranks on cuda:0, gloo instead of
    # RCCL, which refuses two ranks on one device).  Not a measurement mode.
    rehearsal = os.environ.get("PINN_BENCH_REHEARSAL") == "1"
    dev = torch.device("cuda:0" if rehearsal else f"cuda:{local}")
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from __graft_entry__ import _burgers
    from pinnrl_amd import engine as E

    cfg, model, pde = _burgers(dev, hidden=128, layers=4, mapping=32, scale=10.0, seed=0)  # identical theta_0 on every rank
    torch.manual_seed(1 + rank)
    import oracle as O  # only for the seeded CPU sampler shared with the cpu_baseline leg (points are synthetic)

    x_cpu, t_cpu = O.sample_uniform(O.PdeSpec(name="burgers"), args.points)
    x, t = x_cpu.to(dev), t_cpu.to(dev)
    N = x.shape[0]
    n_global = N * world
    prog = model.program()
    pd = pde._pde_desc()
    _, n_grad = prog.grad_layout()
    buf = torch.zeros(n_grad + 4, dtype=torch.float32, device=dev)  # [flat gradient || loss sum || pad]
    flat, loss_sum = buf[:n_grad], buf[n_grad : n_grad + 1]

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        buf.zero_()
        if i is not None:
            ev[i][0].record()
        E.residual_loss_grad(prog, pd, x, t, 1.0 / n_global, flat, loss_sum=loss_sum)
        if i is not None:
            ev[i][1].record()
        if world > 1:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps

    if rank == 0:
        K = 4
        flops_pt = 3 * K * prog.flops_per_point()  # forward jets + delta-propagation + weight-gradient GEMMs
        achieved = flops_pt * N / (kern_ms * 1e-3) / 1e12
        out = {
            "metric": "collocation-points/sec (residual+grad), Burgers 1D",
            "value": n_global * args.steps / elapsed,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: Burgers 1D nu=0.01/pi, fourier 4x128 tanh (41 473 params), "
                            f"{N} collocation points per GPU (50 000 requested, uniform -> 223^2), "
                            "residual + mean(r^2) + d/dtheta in one fused launch",
                "points_per_gpu": N, "global_points": n_global, "streams": K,
                "collective": "none" if world == 1 else "1 all-reduce/step of [grad || loss] (41 477 floats)",
            },
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": pmc_traffic(),
                "kernel": "pinn::jet_kernel_wide<tanh, NT=1, NX=2, reverse>", "kernel_ms": kern_ms,
                "flops_per_point": flops_pt,
            },
            "residual_l2": math.sqrt(float(loss_sum) / n_global),
        }
        if rehearsal:
            out["config"]["rehearsal"] = "all ranks on cuda:0 over gloo - code-path check, not a measurement"
        if world == 1 and not args.no_cpu:
            cb, L_cpu, g_cpu = cpu_baseline(model, x_cpu, t_cpu, args.cpu_seconds)
            out["cpu_baseline"] = cb
            g_gpu = torch.cat([g.flatten().cpu() for g, tr in zip(E.split_flat_grad(prog, flat), prog.trainable) if tr])
            out["parity"] = {
                "loss_rel_err": abs(float(loss_sum) / N - L_cpu) / abs(L_cpu),
                "grad_rel_l2": float((g_gpu - g_cpu).norm() / g_cpu.norm()),
                "residual_l2_cpu": math.sqrt(L_cpu),
            }
            out["speedup_vs_cpu"] = out["value"] / cb["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
