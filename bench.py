"""bench.py -- training throughput of the V-Net hot path (BASELINE.json metric: training patches/sec,
128^3 x 1-channel fp32) on N GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic patches already resident in HBM:
forward (networks.VNet, batch-statistics BN) + softmax/Sorensen-Dice + backward + gradient all-reduce
(N>1, RCCL) + TF-form Adam.  Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events
around every launch of the dominant kernel family (the 16-channel 5x5x5 conv at 128^3: forward,
backward-data and filter-gradient launches); `cpu_baseline` times the CPU restatement (oracle/torch_ref.py,
PyTorch-CPU oneDNN fp32, same graph incl. backward + Adam) on a bounded sub-patch on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3     # MI355X fp32 matrix = vector peak (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak (no sparsity)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1, help="patches per GPU per step (weak scaling)")
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-patch", type=int, default=64)
    ap.add_argument("--compute", choices=("fp32", "bf16"), default="fp32",
                    help="arithmetic of the 5^3 convolutions: fp32 = the reference's (headline metric); bf16 = operands rounded "
                         "to bf16, fp32 accumulate (BASELINE config C5 with --channels 4 --classes 5)")
    return ap.parse_args()


def config(args):
    return {"TrainingSetting": {
        "Data": {"TrainingDataDirectory": "synthetic", "TestingDataDirectory": "synthetic",
                 "ImageFilenames": ["image%d.nii" % i for i in range(args.channels)], "LabelFilename": "label.nii",
                 "Synthetic": {"Cases": 1}},
        "SegmentationClasses": list(range(args.classes)), "BatchSize": args.batch, "PatchShape": [args.patch] * 3,
        "ComputeDtype": args.compute,
        "Networks": {"Name": "VNet", "Dropout": 0.0, "NumChannel": 16, "NumLevels": 4, "NumConvolutions": [1, 2, 3, 3],
                     "BottomConvolutions": 3},
        "Optimizer": {"Name": "Adam", "InitialLearningRate": 1e-2, "Decay": {"Factor": 0.99, "Steps": 100}},
        "Loss": {"Name": "sorensen", "Weights": [], "Alpha": 1}}}


def cpu_baseline(args):
    """CPU restatement ("port"): same graph, PyTorch-CPU ops, fp32, fwd+bwd+Adam, on a cpu_patch^3 sub-patch
    (the conv work scales with the voxel count, so patches/s at 128^3 = measured rate * (cpu_patch/128)^3)."""
    import numpy as np
    import torch
    from oracle import torch_ref as T
    from oracle import vnet_oracle as O
    cores = torch.get_num_threads()
    P = args.cpu_patch
    x, lab = O.synthetic_batch(1, P, args.channels, args.classes, seed=1000)
    torch.manual_seed(0)
    net = T.TorchVNet(args.classes, 16, 4, (1, 2, 3, 3), 3, "prelu", "networks", dtype=torch.float32)
    xt, lt = torch.from_numpy(x), torch.from_numpy(lab)
    with torch.no_grad():
        pass
    loss, _ = T.loss_head(net.forward(xt), lt, "sorensen")       # creates the parameters
    opt = torch.optim.Adam(list(net.p.values()), lr=1e-2, eps=1e-8)
    T.train_step_fp32(net, xt, lt, opt)                            # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < 3 and (time.perf_counter() - t0) < 20.0:
        T.train_step_fp32(net, xt, lt, opt)
        n += 1
    dt = (time.perf_counter() - t0) / n
    scale = (P / float(args.patch)) ** 3
    return {"value": scale / dt, "unit": "patches/s", "cores": cores, "kind": "port",
            "sample": "%d^3 sub-patch (%.4g of the %d^3 voxels), %d timed steps fwd+Dice+bwd+Adam at %.2f s/step, "
                      "rate scaled by the voxel ratio; PyTorch-CPU oneDNN fp32 restatement, not TF1" % (P, scale, args.patch, n, dt)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from vnet_tensorflow_amd import model as M
    from vnet_tensorflow_amd import ops, parallel
    from vnet_tensorflow_amd.data import synthetic_case
    import numpy as np

    rank, local, world = parallel.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    np.random.seed(42)                       # the reference's unseeded global-NumPy Xavier init, made repeatable
    m = M.image2label(None, config(args), device=dev, verbose=False)
    m.rank, m.local_rank, m.world = rank, local, world
    m.read_config()
    m.build_model_graph()
    m._setup_training()

    # synthetic batch, resident in HBM before the timed region (per-rank seed: SURVEY 8(d))
    imgs, labs = [], []
    for b in range(args.batch):
        im, lb = synthetic_case([args.patch] * 3, args.channels, args.classes, 1000 + rank * 64 + b)
        imgs.append(im)
        labs.append(lb[..., None])
    images = torch.from_numpy(np.stack(imgs)).to(dev)
    labels = torch.from_numpy(np.stack(labs).astype(np.int32)).to(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP events go around the launches of the dominant kernel family only (every event packet idles the GPU for
    # ~5.6 us; around all ~140 conv-family launches of a step that is 0.8 ms); BENCH_KERNEL_TABLE=1 times them all
    P = args.patch
    bf16 = args.compute == "bf16"
    fam = set("%s k5 s1 %d^3x%d %d->16" % (k, P, args.batch, c) for k in (("conv-bf16", "wgrad-bf16") if bf16 else ("conv", "wgrad"))
              for c in (16, 32))
    full_table = bool(os.environ.get("BENCH_KERNEL_TABLE"))
    loss = None
    ops.profile_start(None if full_table else fam)       # warm-up runs the same schedule as the timed steps
    for _ in range(args.warmup):
        loss = m.train_step(images, labels)
    ops.profile_stop()
    barrier()
    ops.profile_start(None if full_table else fam)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = m.train_step(images, labels)
    host_dt = time.perf_counter() - t0           # time for the host to ENQUEUE the steps (GPU runs behind)
    barrier()
    dt = time.perf_counter() - t0
    recs = ops.profile_stop()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.detach())

    if rank == 0:
        # dominant kernel family: every launch of the two kernels that carry the 16-output-channel 5^3 convs at
        # full resolution -- conv_kernel<5,1,4,8,8,4,4,1> (dec1/conv_1 fwd 32->16; 16->16 launches when the
        # single-modality input block is not fused) and wgrad_kernel<5,1,4,4,16,1,16> (their filter gradients):
        # 134.2 GF / 268.6 MB algorithmic per 16->16 launch, 268.4 GF / 402.9 MB per 32->16 launch at 128^3
        # 
        fl = by = ms = 0.0
        nl = 0
        per = {}
        for tag, f, b, t in recs:
            a = per.setdefault(tag, [0, 0.0, 0.0, 0.0])
            a[0] += 1; a[1] += f; a[2] += b; a[3] += t
            if tag in fam:
                fl += f; by += b; ms += t; nl += 1
        roof = None
        if nl:
            ach = fl / (ms * 1e-3) / 1e12
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r01_pmc.json")      # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
            if os.path.exists(pmc) and P == 128 and args.batch == 1 and not bf16:
                ks = json.load(open(pmc))["kernels"]
                sel = [v for k, v in ks.items() if k.startswith("conv_kernel<5, 1, 4, 8, 8, 4, 4, 1, false, 5> grid=2097152")
                       or k.startswith("wgrad_kernel<5, 1, 4, 4, 16, 1, 16, 5>")]
                if sel:
                    traffic = round(sum(v["launches"] * v["hbm_bytes_per_launch"] for v in sel) / sum(v["launches"] for v in sel))
            peak = PEAK_BF16_TFLOPS if bf16 else PEAK_FP32_TFLOPS
            kname = ("conv5_bf16_kernel<4,8,16,1> + wgrad5_bf16_kernel<4,4,16,1,16>" if bf16
                     else "conv_kernel<5,1,4,8,8,4,4,1> + wgrad_kernel<5,1,4,4,16,1,16>")
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic,
                    "kernel": kname + ": the 5^3 convs with 16 output channels @%d^3 (fwd, bwd-data, bwd-filter)" % P,
                    "launches": nl, "avg_ms": round(ms / nl, 4), "flops_per_launch": fl / nl, "algorithmic_bytes_per_launch": by / nl,
                    "hbm_GBps_algorithmic": round(by / (ms * 1e-3) / 1e9, 1), "hbm_frac": round(by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
        conv_ms = sum(v[3] for v in per.values()) / args.steps
        conv_tf = sum(v[1] for v in per.values()) / max(sum(v[3] for v in per.values()) * 1e-3, 1e-12) / 1e12
        metric = "training patches/sec (128^3x1ch fp32)" if not bf16 else \
            "training patches/sec (%d^3x%dch, bf16 compute / fp32 accumulate)" % (P, args.channels)
        out = {"metric": metric, "value": round(world * args.batch * args.steps / dt, 4),
               "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
               "config": {"workload": "V-Net (16ch,4 levels,(1,2,3,3),3) train step fwd+Dice+bwd+Adam, %d^3 patch, %d modality, %d classes, "
                                      "batch %d/GPU (BASELINE configs[%d])" % (P, args.channels, args.classes, args.batch, 4 if bf16 else (2 if world == 1 else 3)),
                          "global_batch": world * args.batch, "parallelism": "dp%d" % world, "bn": "per-replica"},
               "final_loss": round(final_loss, 6), "host_enqueue_ms_per_step": round(host_dt / args.steps * 1e3, 3),
               "roofline": roof}
        if full_table:
            out["conv_ms_per_step"], out["conv_tflops"] = round(conv_ms, 3), round(conv_tf, 2)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        if full_table:
            for tag, v in sorted(per.items(), key=lambda kv: -kv[1][3]):
                print("# %-40s n=%3d %8.3f ms/step %7.2f TF/s" % (tag, v[0] // args.steps, v[3] / args.steps, v[1] / (v[3] * 1e-3) / 1e12),
                      file=sys.stderr)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
