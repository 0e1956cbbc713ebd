"""A/B of the coordinate edge kernel's MFMA shape (EGNN_XM16=1: v_mfma_f32_16x16x32_bf16, edge_x_m16.hip; 0: 32x32x16,
edge_bf16_v3.hip) on the C2 batch: outputs of one EGNN forward must agree to fp32-accumulation-order level (same bf16
operands), then interleaved timing rounds of the device sampler in ONE visit (each arm is its own process because the
switch is read once per process).  usage (GPU box): python tools/ab_xm16.py [rounds]"""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(out):
    import bench
    import diffusion_model_amd as dma
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    net = bench.build_net(dma, 4, 64).to(dev).eval()
    net.precision, net.norm_scope = "bf16", "graph"
    from tests.test_gpu_parity import c2_inputs
    h, x = c2_inputs(256)
    sizes = [64] * 256
    plan = dma.fully_connected_plan(sizes, dev)
    with torch.no_grad():
        ho, xo = net(plan, h.to(dev), x.to(dev))
    torch.save({"h": ho.cpu(), "x": xo.cpu()}, out)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    outs = {}
    for arm in ("0", "1"):
        f = f"/tmp/xm16_{arm}.pt"
        subprocess.run([sys.executable, __file__, "--child", f], env=dict(os.environ, EGNN_XM16=arm), check=True)
        outs[arm] = torch.load(f, weights_only=True)
    for k in ("h", "x"):
        a, b = outs["0"][k].double(), outs["1"][k].double()
        print(f"forward {k}: rel diff 16x16x32 vs 32x32x16 = {float((a - b).norm() / a.norm()):.3e}", flush=True)
    res = {"0": [], "1": []}
    for r in range(rounds):
        for arm in ("0", "1"):
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--reps", "3",
                                "--no-cpu-baseline", "--no-train-leg", "--no-slab-leg", "--no-latency-leg"],
                               env=dict(os.environ, EGNN_XM16=arm), capture_output=True, text=True)
            line = json.loads(p.stdout.strip().splitlines()[-1])
            res[arm].append((line["ms_per_step"], line["roofline"]["achieved"], line.get("edge_ms")))
            print(f"round {r} XM16={arm}: {line['ms_per_step']:.3f} ms/step, edge pass {line['roofline']}", flush=True)
    for arm in ("0", "1"):
        ms = sorted(v[0] for v in res[arm])
        print(f"XM16={arm}: ms/step median {ms[len(ms) // 2]:.3f} min {ms[0]:.3f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        main()
