"""bench.py -- training throughput of the V-Net hot path (BASELINE.json metric: training patches/sec,
128^3 x 1-channel fp32) on N GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...            (no launcher: starts the N ranks itself as a child torch.distributed.run, see self_launch)

One "step" = one pass of the hot path over one batch of synthetic patches already resident in HBM:
forward (networks.VNet, batch-statistics BN) + softmax/Sorensen-Dice + backward + gradient all-reduce
(N>1, RCCL) + TF-form Adam + filter repack.  The step is enqueued the way image2label.train() enqueues it: as a
replayed hipGraph of the whole step (2 eager steps + 1 capture happen before the W warm-up steps; VNET_STEP_GRAPH=0
gives the kernel-by-kernel eager enqueue).  Rank 0 prints ONE JSON line.

`roofline`: HIP events around every launch of the dominant kernel family -- decoder level 1 / conv_1, the 5x5x5
convolution with 16 output channels at 128^3: forward (32->16), backward-data (16->32) and filter gradient, 268.4 GF
algorithmic each.  Events cannot be timed inside a replayed hipGraph on this runtime (profiles/probes/graph_event_probe.py),
so in graph mode the family is timed in R eager steps of the same model that follow the timed region.
`c5_bf16` (N=1 only): the same measurement for BASELINE configs[4]'s per-GPU workload (4 modalities, 5
classes; bf16 activations / gradients in HBM, bf16 operands into the matrix cores, fp32 accumulate, fp32 batch-norm statistics
and Dice sums -- SURVEY 8(d)), outside the headline's timed region.  `sustained`: >= 400 further replays after the timed
region (ms/step, shader clock and power from rocm-smi when readable).  `cpu_baseline`: the CPU
restatement (oracle/torch_ref.py: same graph on PyTorch-CPU oneDNN fp32 incl. backward + Adam) timed on the real
128^3 step on rank 0 at N=1.
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
PEAK_F32X3_TFLOPS = 2500.0 / 6.0   # fp32-equivalent peak of the six-bf16-product arithmetic (416.7)
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
    ap.add_argument("--no-c5", action="store_true", help="skip the bf16 / 4-modality / 5-class sub-measurement (N=1)")
    ap.add_argument("--no-c2", action="store_true", help="skip the 64^3 batch-2 fp32 sub-measurement (BASELINE configs[1], N=1)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 400 extra replays after the timed region (N=1)")
    ap.add_argument("--cpu-patch", type=int, default=0, help="CPU baseline patch edge (0 = the benchmarked patch itself)")
    ap.add_argument("--pin-core", type=int, default=-1,
                    help="pin this process to ONE host core before the GPU is initialised (host-overhead experiment)")
    ap.add_argument("--no-x3", action="store_true", help="skip the fp32_split3 sub-measurement (same network, 5^3 convolutions on the bf16 pipe, N=1)")
    ap.add_argument("--compute", choices=("fp32", "fp32_split3", "bf16"), default=None,
                    help="default: the headline arithmetic -- fp32_split3 if profiles/r06_promotion.json says every gate of VERDICT r5's ruling "
                         "passed, else fp32 (native v_mfma_f32_16x16x4_f32); the other fp32 mode is then measured beside it.  fp32_split3 = "
                         "fp32 tensors, 5^3 products from six bf16 MFMAs of exactly split operands; bf16 = bf16 activations in HBM and bf16 "
                         "operands into the matrix cores, fp32 accumulate (BASELINE config C5 with --channels 4 --classes 5)")
    return ap.parse_args()


def config(patch, batch, channels, classes, compute):
    return {"TrainingSetting": {
        "Data": {"TrainingDataDirectory": "synthetic", "TestingDataDirectory": "synthetic",
                 "ImageFilenames": ["image%d.nii" % i for i in range(channels)], "LabelFilename": "label.nii",
                 "Synthetic": {"Cases": 1}},
        "SegmentationClasses": list(range(classes)), "BatchSize": batch, "PatchShape": [patch] * 3,
        "ComputeDtype": compute,
        "Networks": {"Name": "VNet", "Dropout": 0.0, "NumChannel": 16, "NumLevels": 4, "NumConvolutions": [1, 2, 3, 3],
                     "BottomConvolutions": 3},
        "Optimizer": {"Name": "Adam", "InitialLearningRate": 1e-2, "Decay": {"Factor": 0.99, "Steps": 100}},
        "Loss": {"Name": "sorensen", "Weights": [], "Alpha": 1}}}


def host_cpu():
    """(model string, physical cores, logical cpus) of this host."""
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return model, len(phys) or (os.cpu_count() or 1), os.cpu_count() or 1


def cpu_baseline(args):
    """CPU restatement ("port", NOT TensorFlow 1.15): the same graph on PyTorch-CPU ops (oneDNN conv3d, fp32), fwd + Dice
    + bwd + Adam, on the benchmarked patch itself -- one warm-up step and one or two timed steps, no extrapolation."""
    import torch
    from oracle import torch_ref as T
    from oracle import vnet_oracle as O
    model, phys, logical = host_cpu()
    threads = torch.get_num_threads()

    def time_patch(P, max_steps, budget_s):
        x, lab = O.synthetic_batch(1, P, args.channels, args.classes, seed=1000)
        torch.manual_seed(0)
        net = T.TorchVNet(args.classes, 16, 4, (1, 2, 3, 3), 3, "prelu", "networks", dtype=torch.float32)
        xt, lt = torch.from_numpy(x), torch.from_numpy(lab)
        T.loss_head(net.forward(xt), lt, "sorensen")                    # creates the parameters
        opt = torch.optim.Adam(list(net.p.values()), lr=1e-2, eps=1e-8)
        T.train_step_fp32(net, xt, lt, opt)                             # warm-up
        t0 = time.perf_counter()
        n = 0
        while n < max_steps and (n == 0 or (time.perf_counter() - t0) * (n + 1) / n < budget_s):
            T.train_step_fp32(net, xt, lt, opt)
            n += 1
        return (time.perf_counter() - t0) / n, n

    P = args.cpu_patch or args.patch
    dt, n = time_patch(P, 2, 60.0)
    dt32, n32 = time_patch(32, 3, 10.0)                                  # BASELINE configs[0] (C1): 32^3, B=1
    scale = (P / float(args.patch)) ** 3
    sample = "%d^3 patch, batch 1: 1 warm-up + %d timed training steps (fwd+Dice+bwd+Adam) at %.2f s/step" % (P, n, dt)
    if scale != 1.0:
        sample += ", rate scaled by the voxel ratio %.4g" % scale
    sample += "; PyTorch-CPU oneDNN fp32 restatement of the reference graph (oracle/torch_ref.py), NOT TensorFlow 1.15; " \
              "CPU %s, %d physical cores / %d logical, %d threads" % (model, phys, logical, threads)
    return {"value": scale / dt, "unit": "patches/s", "cores": threads, "kind": "port", "sample": sample,
            "cpu_model": model, "physical_cores": phys, "threads": threads, "seconds_per_step": round(dt, 3),
            "c1_32cube": {"value": round(1.0 / dt32, 4), "unit": "patches/s", "seconds_per_step": round(dt32, 4), "steps": n32}}


def promotion():
    """profiles/r06_promotion.json: the gates of VERDICT r5's ruling on reporting fp32_split3 as `value` ((a) seed spread, (b) adversarial
    operands, (c) non-finite semantics), each with its evidence file, and the resulting decision -- written by profiles/promotion_gates.py
    from the committed records, read here so that the bench line and the records cannot disagree."""
    path = os.path.join(ROOT, "profiles", "r06_promotion.json")
    try:
        return json.load(open(path))
    except (OSError, ValueError):
        return None


X3_ARITHMETIC = ("fp32 tensors everywhere (activations, gradients, batch-norm, loss, optimiser); every product of the 5^3 convolutions "
                 "(forward, backward-data, filter gradient; every level, 128^3 .. 8^3) = 6 v_mfma_f32_16x16x32_bf16 products of exactly "
                 "split operands (x = h + m + l, 3 x 8 significant bits, no remainder); fp32 accumulate; the 1-channel input "
                 "block, the 2^3 stride-2 convolutions and the 1^3 head on v_mfma_f32_16x16x4_f32")


def latest_pmc():
    """The newest committed profiles/rNN_pmc.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")))
    return files[-1] if files else None


def smi_sample():
    """One reading of shader clock and power from rocm-smi while the GPU is busy (None if it cannot be read)."""
    import subprocess
    # (a clean environment for the child: under `rocprofv3 --pmc` LD_PRELOAD carries the profiler, which would initialise the GPU in
    #  rocm-smi's `#!/usr/bin/env python3` hop -- an exec from a GPU-initialised process, which the GPU boxes refuse)
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "ROCTX"))}
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=30, env=env).stdout
        card = json.loads(out)
        card = card.get("card0", next(iter(card.values())))
        sclk = power = None
        for k, v in card.items():
            kl = k.lower()
            if "sclk" in kl and "clock" in kl and sclk is None:
                sclk = "".join(ch for ch in str(v) if ch.isdigit() or ch == ".")
            if "power" in kl and "(w)" in kl and power is None:
                power = v
        return {"sclk_mhz": float(sclk) if sclk else None, "power_w": float(power) if power not in (None, "N/A") else None}
    except Exception:
        return None


def sustained_run(m, images, labels, ms_per_step):
    """>= 400 more replays of the step AFTER (outside) the timed region: a DVFS-settled figure next to the K-step one, long
    enough for the driver's GPU sampler to see the device busy.  rocm-smi is read once from a second thread mid-run."""
    import threading
    import torch
    n = int(min(1500, max(400, 8000.0 / max(ms_per_step, 1e-3))))
    box = []
    th = threading.Thread(target=lambda: (time.sleep(1.0), box.append(smi_sample())))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th.start()
    for _ in range(n):
        m.train_step(images, labels)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    th.join()
    smi = box[0] if box and box[0] else {}
    return {"steps": n, "ms_per_step": round(dt / n * 1e3, 3), "seconds": round(dt, 2),
            "sclk_mhz": smi.get("sclk_mhz"), "power_w": smi.get("power_w"),
            "note": "graph replays after the timed region (not part of `value`)"}


def hbm_kernel_table(dev, bf16, classes, iters=50):
    """The HBM-bound kernels of the step ("report both for every kernel"), one kernel at a time at the shapes of the
    128^3 network's level 1: `iters` back-to-back launches between two HIP events on the launch stream (the C ABI called as ops.py
    calls it), over the ALGORITHMIC bytes of the pass -- what it must read and write once.  Replaces, per the
    reference: networks.py:259,319 (batch-norm + PReLU), layers2.py:78-94 (2^3 stride-2 / transposed convolutions),
    model.py:447,60-83 (softmax + Dice) and model.py:649-662 (Adam)."""
    import torch
    from vnet_tensorflow_amd import _lib, ops
    L = _lib.lib()
    P_ = ops._ptr
    st = ops._stream()
    dt = torch.bfloat16 if bf16 else torch.float32
    esz = 2 if bf16 else 4
    out = {}

    def timed(name, nbytes, fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        gbps = nbytes / us / 1e3
        out[name] = {"us": round(us, 2), "algorithmic_bytes": int(nbytes), "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / PEAK_HBM_GBS, 4)}

    M, C = 128 ** 3, 16
    x = torch.randn(M, C, device=dev).to(dt)
    r = torch.randn(M, C, device=dev).to(dt)
    dy = torch.randn(M, C, device=dev).to(dt)
    y, ds = torch.empty_like(x), torch.empty_like(x)
    mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    gamma, beta, alpha = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.full((C,), 0.25, device=dev)
    dg, db, da = (torch.zeros(C, device=dev) for _ in range(3))
    mm, mv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    nb = L.vnet_bn_ws_bytes(C)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    if bf16:
        timed("bn_stats 128^3x16", M * C * esz, lambda: L.vnet_bn_stats_b16(P_(x), None, M, C, 1e-3, 0.99, P_(mean), P_(invstd), P_(mm), P_(mv), P_(ws), nb, st))
        timed("bn_act_fwd 128^3x16 +res", 3 * M * C * esz, lambda: L.vnet_bn_act_fwd_b16(P_(x), P_(r), 0, M, C, P_(mean), P_(invstd), P_(gamma), P_(beta), 2, P_(alpha), P_(y), st))
        timed("bn_act_bwd_reduce 128^3x16 +res", 3 * M * C * esz, lambda: L.vnet_bn_act_bwd_reduce_b16(P_(dy), P_(x), P_(r), 0, M, C, P_(mean), P_(invstd), P_(gamma), P_(beta), 2, P_(alpha), P_(dg), P_(db), P_(da), P_(ws), nb, st))
        timed("bn_act_bwd_apply 128^3x16 +res", 4 * M * C * esz, lambda: L.vnet_bn_act_bwd_apply_b16(P_(dy), P_(x), P_(r), 0, M, C, P_(mean), P_(invstd), P_(gamma), P_(beta), 2, P_(alpha), P_(db), P_(dg), float(M), None, P_(ds), st))
    else:
        timed("bn_stats 128^3x16", M * C * esz, lambda: L.vnet_bn_stats(P_(x), None, 0, M, C, 1e-3, 0.99, P_(mean), P_(invstd), P_(mm), P_(mv), P_(ws), nb, st))
        timed("bn_act_fwd 128^3x16 +res", 3 * M * C * esz, lambda: L.vnet_bn_act_fwd(P_(x), P_(r), 0, M, C, P_(mean), P_(invstd), P_(gamma), P_(beta), 2, P_(alpha), P_(y), st))
        timed("bn_act_bwd_reduce 128^3x16 +res", 3 * M * C * esz, lambda: L.vnet_bn_act_bwd_reduce(P_(dy), P_(x), P_(r), 0, M, C, P_(mean), P_(invstd), P_(gamma), P_(beta), 2, P_(alpha), P_(dg), P_(db), P_(da), P_(ws), nb, st))
        timed("bn_act_bwd_apply 128^3x16 +res", 4 * M * C * esz, lambda: L.vnet_bn_act_bwd_apply(P_(dy), P_(x), P_(r), 0, M, C, P_(mean), P_(invstd), P_(gamma), P_(beta), 2, P_(alpha), P_(db), P_(dg), float(M), None, P_(ds), st))
    del r, dy, y, ds
    # level-1 down convolution 16 -> 32 (128^3 -> 64^3) and transposed convolution 32 -> 16 (64^3 -> 128^3)
    x5 = x.view(1, 128, 128, 128, 16)
    wd = torch.randn(2, 2, 2, 16, 32, device=dev) * 0.1
    bd = torch.zeros(32, device=dev)
    xc = torch.randn(1, 64, 64, 64, 32, device=dev).to(dt)
    wu = torch.randn(2, 2, 2, 16, 32, device=dev) * 0.1
    bu = torch.zeros(16, device=dev)
    with torch.no_grad():
        timed("down_conv 2^3 s2 16->32 @128^3", (M * 16 + M // 8 * 32) * esz, lambda: ops.conv(x5, wd, bd, 2, 2))
        timed("up_conv 2^3 s2 32->16 @64^3", (M // 8 * 32 + M * 16) * esz, lambda: ops.conv_transpose2(xc, wu, bu, (128, 128, 128)))
    del x, x5, xc
    # softmax + Dice at 128^3 (fp32 logits in both modes), forward and backward
    K = classes
    logits = torch.randn(1, 128, 128, 128, K, device=dev)
    labels = torch.randint(0, K, (1, 128, 128, 128, 1), device=dev, dtype=torch.int32)
    kind = ops.parse_loss("sorensen")
    loss = torch.empty((), device=dev); dice = torch.empty((), device=dev)
    coef = torch.empty(2 * K + 1, device=dev)
    nbl = L.vnet_loss_ws_bytes(1, K)
    wsl = torch.empty(max(nbl, 16), dtype=torch.uint8, device=dev)
    g1 = torch.ones((), device=dev)
    dl = torch.empty_like(logits)
    timed("softmax_dice_fwd 128^3 K=%d" % K, M * K * 4 + M * 4,
          lambda: L.vnet_softmax_dice_fwd(P_(logits), P_(labels), 1, M, K, kind, None, 1.0, 1e-5, None, None, P_(loss), P_(dice), P_(coef), P_(wsl), nbl, st))
    timed("softmax_dice_bwd 128^3 K=%d" % K, 2 * M * K * 4 + M * 4,
          lambda: L.vnet_softmax_dice_bwd(P_(logits), P_(labels), 1, M, K, kind, None, 1.0, P_(coef), P_(g1), P_(dl), st))
    del logits, labels, dl
    # Adam over the whole flat parameter vector (43 940 486 floats): reads p, g, m, v, writes p, m, v
    n = 43940486
    p_ = torch.zeros(n, device=dev); g_ = torch.full((n,), 1e-3, device=dev); m_ = torch.zeros(n, device=dev); v_ = torch.zeros(n, device=dev)
    timed("adam 43.9M parameters", 7 * n * 4, lambda: L.vnet_adam_apply(P_(p_), P_(g_), P_(m_), P_(v_), n, 1e-3, 0.9, 0.999, 1e-8, 1.0, st))
    del p_, g_, m_, v_
    # the batched filter repack after the optimiser (model.py:649-662's variables -> the kernels' packed images): the three
    # bottom-level filters (5^3, 256 -> 256), both images; bf16: ONE read of the fp32 weights for the two bf16 images (round 4)
    with ops.context(ops.OpsContext()):
        fl = [torch.nn.Parameter(torch.randn(5, 5, 5, 256, 256, device=dev) * 0.01) for _ in range(3)]
        modes = (ops.PACK_FWD_BF16, ops.PACK_BWD_BF16) if bf16 else (ops.PACK_FWD, ops.PACK_BWD)
        for w_ in fl:
            for md in modes:
                ops.packed_weights(w_, md, 125, 256, 256)
        ops.invalidate_packed()
        ops.repack_registered()                       # (builds the descriptor table)
        descs, nrows = ops._PACK_REG["descs"], ops._PACK_REG["nrows"]
        npar = 3 * 125 * 256 * 256
        timed("repack 3 x (5^3, 256->256), forward + backward-data images", npar * (4 + (4 if bf16 else 8)),
              lambda: L.vnet_pack_weights_batched(P_(descs), nrows, st))
        ops.clear_pack_registry()
        del fl, descs
    torch.cuda.empty_cache()
    return out


def peak_for(bf16):
    return PEAK_BF16_TFLOPS if bf16 else PEAK_FP32_TFLOPS


def family_tags(P, B, bf16, x3=False):
    """Launch tags (ops._Timed) of decoder level 1 / conv_1: forward 32->16, backward-data 16->32, filter gradient."""
    c, w = ("conv-bf16", "wgrad-bf16") if bf16 else (("conv-x3", "wgrad-x3") if x3 else ("conv", "wgrad"))
    return {"%s k5 s1 %d^3x%d 32->16" % (c, P, B), "%s k5 s1 %d^3x%d 16->32" % (c, P, B), "%s k5 s1 %d^3x%d 32->16" % (w, P, B)}


def measure(args, patch, batch, channels, classes, compute, rank, local, world):
    """Build the model, run prepare + warm-up + the timed K steps + the roofline replays.  Returns the result dict."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from vnet_tensorflow_amd import model as M
    from vnet_tensorflow_amd import ops
    from vnet_tensorflow_amd.data import synthetic_case

    import gc
    gc.collect()                             # a previous measurement's model (graph memory pool, packed filters) is gone
    torch.cuda.empty_cache()                 # (compute dtype and pack registry are per-model state: nothing to reset by hand)
    dev = torch.device("cuda", local)
    np.random.seed(42)                       # the reference's unseeded global-NumPy Xavier init, made repeatable
    m = M.image2label(None, config(patch, batch, channels, classes, compute), device=dev, verbose=False)
    m.rank, m.local_rank, m.world = rank, local, world
    m.read_config()
    m.build_model_graph()
    m._setup_training()

    # synthetic batch, resident in HBM before the timed region (per-rank seed: SURVEY 8(d))
    imgs, labs = [], []
    for b in range(batch):
        im, lb = synthetic_case([patch] * 3, channels, classes, 1000 + rank * 64 + b)
        imgs.append(im)
        labs.append(lb[..., None])
    images = torch.from_numpy(np.stack(imgs)).to(dev)
    labels = torch.from_numpy(np.stack(labs).astype(np.int32)).to(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    bf16 = compute == "bf16"
    x3 = compute == "fp32_split3"
    fam = family_tags(patch, batch, bf16, x3)
    timed = set(fam) | {ops.WGRAD_GROUP_TAG}     # + the grouped launch of the other 5^3 filter gradients (bf16 storage)
    full_table = bool(os.environ.get("BENCH_KERNEL_TABLE"))
    graph = m._graph_mode() != "off"
    mode = m._graph_mode()
    # eager enqueue: HIP events go around the launches of the dominant kernel family only (an event packet idles the GPU
    # for ~5.6 us); graph replay: events cannot be timed inside a hipGraph on this runtime, the family is timed in eager
    # steps right after the timed region
    loss = None
    if not graph:
        ops.profile_start(None if full_table else timed)
    nprep = 3 if graph else 0                # prepare: 2 eager steps + the capture (and first replay) of the step graph
    if mode == "segmented":
        tn = m._dp_tuner()                   # + the data-parallel start-up autotune: 3 rounds x 5 steps each of segmented / serial / eager
        nprep += tn.total_steps() if tn is not None else 0
    for _ in range(nprep):
        loss = m.train_step(images, labels)
    mode = m.step_mode()                     # (the autotune may have picked the eager enqueue)
    for _ in range(args.warmup):
        loss = m.train_step(images, labels)
    if not graph:
        ops.profile_stop()                   # (synchronises; the warm-up records are dropped)
    barrier()
    if not graph:
        ops.profile_start(None if full_table else timed)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = m.train_step(images, labels)
    host_dt = time.perf_counter() - t0           # time for the host to ENQUEUE the steps (GPU runs behind)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.detach())
    nroof = min(args.steps, 10)
    fam_smi = None
    if graph:
        m.force_eager = True                     # same kernels, enqueued one by one so that events can bracket them
        m.train_step(images, labels)
        barrier()
        # shader clock / power WHILE the family is timed (VERDICT r5 #7d: the bf16-pipe kernels follow the clock a box sustains):
        # rocm-smi is read from a second thread; the eager steps go on (untimed) until it has answered
        import threading
        box = []
        th = threading.Thread(target=lambda: box.append(smi_sample())) if (world == 1 and rank == 0) else None
        if th is not None:
            th.start()
        ops.profile_start(None if full_table else timed)
        for _ in range(nroof):
            m.train_step(images, labels)
        recs = ops.profile_stop()
        if th is not None:
            t_end = time.perf_counter() + 4.0
            while th.is_alive() and time.perf_counter() < t_end:
                m.train_step(images, labels)
                torch.cuda.synchronize()
            th.join()
            fam_smi = box[0] if box else None
    else:
        recs = ops.profile_stop()
    barrier()
    sustained = None
    if world == 1 and rank == 0 and not args.no_sustained:
        m.force_eager = False
        sustained = sustained_run(m, images, labels, dt / args.steps * 1e3)

    res = {"value": round(world * batch * args.steps / dt, 4), "ms_per_step": round(dt / args.steps * 1e3, 3),
           "final_loss": round(final_loss, 6), "host_enqueue_ms_per_step": round(host_dt / args.steps * 1e3, 3),
           "dp_autotune_ms": ({c: round(t * 1e3, 3) for c, t in zip(m._tuner.candidates, m._tuner.times)}
                              if getattr(m, "_tuner", None) is not None else None),
           "step_enqueue": {"off": "eager (one ctypes launch per kernel)", "whole": "hipGraph replay of the whole step",
                            "segmented": "hipGraph(gradients) + eager RCCL bucket all-reduces + hipGraph(optimiser)",
                            "serial": "hipGraph(gradients) + eager RCCL bucket all-reduces after backward + hipGraph(optimiser)"}.get(mode, mode),
           "roofline": None, "sustained": sustained}
    if rank != 0:
        return res
    fl = by = ms = 0.0
    nl = 0
    per = {}
    for tag, f, b, t in recs:
        a = per.setdefault(tag, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += f; a[2] += b; a[3] += t
        if tag in fam:
            fl += f; by += b; ms += t; nl += 1
    if nl:
        ach = fl / (ms * 1e-3) / 1e12
        traffic = tsrc = tper = None
        pmc = latest_pmc()                                        # separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        if pmc and patch == 128 and batch == 1:
            fams = json.load(open(pmc)).get("families", {})
            ent = fams.get("bf16" if bf16 else ("f32x3" if x3 else "fp32"))
            if ent:
                traffic = ent["hbm_bytes_per_launch"]             # mean over the family's three launches
                tper = ent.get("per_kernel")                      # forward / backward-data / filter gradient, each per launch
                tsrc = "profiles/%s: rocprofv3 --pmc passes of this command (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), not this run" % os.path.basename(pmc)
        peak = PEAK_BF16_TFLOPS if bf16 else (PEAK_F32X3_TFLOPS if x3 else PEAK_FP32_TFLOPS)
        kname = ("conv5_bf16_c16pp_kernel (fwd 32->16: filter L2 -> VGPR, two workgroups per CU) + conv5_bf16_r32_kernel (bwd-data 16->32) + wgrad5_bf16_rr_kernel (row-reuse filter gradient), bf16 tensors in and out" if bf16
                 else "conv5_x3_kernel (fwd 32->16, bwd-data 16->32) + wgrad5_x3_kernel: fp32 tensors, six v_mfma_f32_16x16x32_bf16 per product of "
                      "exactly split operands; peak = 2500 / 6 TF/s fp32-equivalent" if x3
                 else "conv_kernel<5,1,4,8,8,4,4,{1,2}> (fwd 32->16, bwd-data 16->32) + wgrad_kernel<5,1,4,4,16,1,16>")
        res["roofline"] = {
            "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "traffic": traffic, "traffic_per_kernel": tper,
            "traffic_over_algorithmic": (round(traffic / (by / nl), 3) if traffic else None), "traffic_source": tsrc,
            "kernel": kname + ": decoder level 1 conv_1, the 5^3 conv with 16 output channels @%d^3 -- forward, backward-data "
                              "and filter-gradient launches" % patch,
            "launches": nl, "avg_ms": round(ms / nl, 4), "flops_per_launch": fl / nl, "algorithmic_bytes_per_launch": by / nl,
            "sclk_mhz": (fam_smi or {}).get("sclk_mhz"), "power_w": (fam_smi or {}).get("power_w"),
            "per_launch_ms": {tag: round(per[tag][3] / per[tag][0], 4) for tag in sorted(fam) if tag in per},
            "hbm_GBps_algorithmic": round(by / (ms * 1e-3) / 1e9, 1), "hbm_frac": round(by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            "measured": ("HIP events on the launch stream in %d eager steps that follow the timed region (same process, same "
                         "kernels and arguments; events cannot be timed inside a replayed hipGraph on this runtime)" % nroof)
            + ("; the filter-gradient launch is timed on its own here -- in the replayed step it is one of the layers of the grouped "
               "launch (`filter_gradient_group`)" if bf16 else "")
            if graph else "HIP events on the launch stream inside the timed region"}
    if ops.WGRAD_GROUP_TAG in per and per[ops.WGRAD_GROUP_TAG][3] > 0:
        v = per[ops.WGRAD_GROUP_TAG]
        res["filter_gradient_group"] = {
            "kernel": "wgrad5_b16_group_kernel: the 5^3 filter gradients of the backward pass in one launch (csrc/conv_b16.hip; the layer timed "
                      "on its own for `roofline` and the zero-padded network input are not in it here)",
            "avg_ms": round(v[3] / v[0], 4), "tflops": round(v[1] / (v[3] * 1e-3) / 1e12, 1),
            "frac_of_peak": round(v[1] / (v[3] * 1e-3) / 1e12 / peak_for(bf16), 4), "launches_timed": v[0],
            "note": "incl. its split-K slab writes; the slabs' one batched reduce is a separate launch"}
    if world == 1 and patch == 128 and not x3 and not os.environ.get("BENCH_NO_HBM_TABLE"):
        del m
        gc.collect()
        res["hbm_kernels"] = hbm_kernel_table(dev, bf16, classes)
        res["hbm_kernels_note"] = ("one kernel at a time after the timed region: 50 back-to-back launches between HIP events on the launch "
                                   "stream, GB/s over the algorithmic bytes of the pass, fraction of the 8 TB/s HBM3E peak "
                                   "(a float4 copy reaches 6.3 TB/s = 0.79 on this part)")
    if full_table:
        tot_ms = sum(v[3] for v in per.values())
        res["conv_ms_per_step"] = round(tot_ms / (nroof if graph else args.steps), 3)
        res["conv_tflops"] = round(sum(v[1] for v in per.values()) / max(tot_ms * 1e-3, 1e-12) / 1e12, 2)
        for tag, v in sorted(per.items(), key=lambda kv: -kv[1][3]):
            nst = nroof if graph else args.steps
            print("# %-40s n=%3d %8.3f ms/step %7.2f TF/s" % (tag, v[0] // nst, v[3] / nst, v[1] / (v[3] * 1e-3) / 1e12),
                  file=sys.stderr)
    return res


def summary(out):
    """The headline numbers of every leg in one compact object, emitted as the LAST key of the JSON line."""
    def leg(d):
        if not d:
            return None
        r = d.get("roofline") or {}
        su = d.get("sustained") or {}
        return {"patches_per_s": d.get("value"), "ms": d.get("ms_per_step"), "frac": r.get("frac"), "family_avg_ms": r.get("avg_ms"),
                "family_sclk_mhz": r.get("sclk_mhz"), "sustained_ms": su.get("ms_per_step"), "sustained_sclk_mhz": su.get("sclk_mhz"),
                "sustained_power_w": su.get("power_w")}
    x3_head = "v_mfma_f32_16x16x32_bf16" in (out.get("arithmetic") or "")
    s = {"value_is": "fp32_split3" if x3_head else out.get("dtype"),
         "fp32": leg(out.get("c3_f32_native") if x3_head else (out if out.get("dtype") == "f32" else None)),
         "f32x3": leg(out if x3_head else out.get("c3_f32x3")), "c5": leg(out.get("c5_bf16")),
         "c2": ({"patches_per_s": out["c2_64cube_b2"]["value"], "ms": out["c2_64cube_b2"]["ms_per_step"]} if out.get("c2_64cube_b2") else None)}
    g = (out.get("c5_bf16") or {}).get("filter_gradient_group")
    if g:
        s["c5"]["filter_gradient_group_frac"] = g.get("frac_of_peak")
    return s


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks ourselves.  The parent never touches
    the GPU (device_count() does not initialise it on this image; nothing is exec'ed): it starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <same arguments>` as a CHILD process, relays the
    child's output -- rank 0's single JSON line -- and exits with its code.  Fewer than N visible devices: refused, unless the
    gloo test hook (VNET_DIST_BACKEND=gloo: several ranks share one device) is set."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and os.environ.get("VNET_DIST_BACKEND") != "gloo":
        sys.stderr.write("bench.py: --gpus %d but only %d device(s) visible\n" % (args.gpus, ndev))
        return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode       # stdout / stderr inherited: the JSON line passes through


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    if args.pin_core >= 0:
        os.sched_setaffinity(0, {args.pin_core})     # before anything touches the GPU; no wrapper / launcher hop
    import torch
    import torch.distributed as dist
    from vnet_tensorflow_amd import ops, parallel

    rank, local, world = parallel.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1 and dist.get_world_size() != args.gpus:
        raise SystemExit("process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    torch.cuda.set_device(torch.device("cuda", local))

    promo = promotion()
    explicit = args.compute is not None
    if not explicit:
        args.compute = "fp32_split3" if (promo and promo.get("promote")) else "fp32"
    promoted = args.compute == "fp32_split3" and not explicit        # the headline IS fp32_split3: the native step goes beside it
    bf16 = args.compute == "bf16"
    r = measure(args, args.patch, args.batch, args.channels, args.classes, args.compute, rank, local, world)
    if rank == 0:
        P = args.patch
        metric = "training patches/sec (128^3x1ch fp32)" if not bf16 else \
            "training patches/sec (%d^3x%dch, bf16 compute / fp32 accumulate)" % (P, args.channels)
        out = {"metric": metric, "value": r["value"],
               "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16" if bf16 else "f32",
               "arithmetic": (X3_ARITHMETIC if args.compute == "fp32_split3" else
                              "bf16 storage and bf16 MFMA operands, fp32 accumulate / statistics / Dice / master weights" if bf16 else
                              "fp32 tensors, v_mfma_f32_16x16x4_f32 products, fp32 accumulate (the reference's arithmetic)"),
               "headline_rule": ({"promoted": bool(promo.get("promote")), "gates": promo.get("gates"), "source": "profiles/r06_promotion.json"}
                                 if promo else None),
               "data": "synthetic",
               "config": {"workload": "V-Net (16ch,4 levels,(1,2,3,3),3) train step fwd+Dice+bwd+Adam, %d^3 patch, %d modality, %d classes, "
                                      "batch %d/GPU (BASELINE configs[%d])" % (P, args.channels, args.classes, args.batch, 4 if bf16 else (2 if world == 1 else 3)),
                          "global_batch": world * args.batch, "parallelism": "dp%d" % world, "bn": "per-replica",
                          "ranks": world, "backend": (dist.get_backend() if world > 1 else None)},
               "final_loss": r["final_loss"], "host_enqueue_ms_per_step": r["host_enqueue_ms_per_step"],
               "step_enqueue": r["step_enqueue"], "dp_autotune_ms": r.get("dp_autotune_ms"), "pinned_to_core": args.pin_core if args.pin_core >= 0 else None,
               "roofline": r["roofline"], "hbm_kernels": r.get("hbm_kernels"), "hbm_kernels_note": r.get("hbm_kernels_note"),
               "sustained": r.get("sustained"),
               "parity": ("unpinned by the reference (TF 1.15 cannot run here, the reference has no tests): every number is checked against the "
                          "repo's fp64 oracle.  Full-size (this workload) gradients are held to 2.75e-2 rel-L2 per tensor / 1.4e-2 median / 1.5e-3 "
                          "whole vector in BOTH fp32 modes (one set of bounds = 1.25 x the largest value of a ten-draw seed spread, "
                          "profiles/r06_golden_seed_spread.txt: the per-tensor worst of a run is chaotic, 2.2e-2 in the reference's own arithmetic on one draw) "
                          "-- per tensor a 27x relaxation of BASELINE.md 2.1's 1e-3, which fp32 itself does not resolve here (stock "
                          "PyTorch-CPU fp32 on the same fixture: 5e-3); logits 1e-3, loss 1e-5, Dice sums 1e-5, argmax >= 99.99 % as "
                          "BASELINE states them (tests/test_hip_golden_full.py, DESIGN.md section 6)")}
        for k in ("conv_ms_per_step", "conv_tflops"):
            if k in r:
                out[k] = r[k]
    if world == 1 and args.compute in ("fp32", "fp32_split3") and not args.no_c5 and args.patch == 128 and args.channels == 1:
        # BASELINE configs[4] per-GPU workload on the same record (outside the headline's timed region)
        c5 = measure(args, args.patch, args.batch, 4, 5, "bf16", rank, local, world)
        c5["metric"] = "training patches/sec (128^3x4ch, 5 classes, bf16 storage + bf16 conv operands / fp32 accumulate, fp32 BN statistics and Dice sums), 1 GPU"
        c5["dtype"] = "bf16"
        c5["steps"], c5["warmup"] = args.steps, args.warmup
        out["c5_bf16"] = c5
    if world == 1 and promoted and not args.no_x3 and args.patch == 128 and args.channels == 1:
        # promoted headline: the SAME step on the native fp32 MFMA kernels beside it, with its own roofline (VERDICT r5's ruling)
        nat = measure(args, args.patch, args.batch, args.channels, args.classes, "fp32", rank, local, world)
        out["c3_f32_native"] = {"metric": "training patches/sec (128^3x1ch fp32, v_mfma_f32_16x16x4_f32 kernels), 1 GPU", "dtype": "f32",
                                "arithmetic": "fp32 tensors, v_mfma_f32_16x16x4_f32 products, fp32 accumulate", "value": nat["value"],
                                "unit": "patches/s", "ms_per_step": nat["ms_per_step"], "steps": args.steps, "warmup": args.warmup,
                                "final_loss": nat["final_loss"], "step_enqueue": nat["step_enqueue"], "roofline": nat["roofline"],
                                "sustained": nat.get("sustained")}
        out["hbm_kernels"], out["hbm_kernels_note"] = nat.get("hbm_kernels"), nat.get("hbm_kernels_note")
    if world == 1 and args.compute == "fp32" and not args.no_x3 and args.patch == 128 and args.channels == 1:
        # native headline: the same workload with ComputeDtype fp32_split3 NEXT TO it
        x3r = measure(args, args.patch, args.batch, args.channels, args.classes, "fp32_split3", rank, local, world)
        out["c3_f32x3"] = {"metric": "training patches/sec (128^3x1ch, fp32 tensors; 5^3 convolutions of the levels >= 32^3 as six bf16 MFMA "
                                     "products of exactly split operands, fp32 accumulate), 1 GPU",
                           "dtype": "f32 (3xbf16 split operands, fp32 accumulate)", "value": x3r["value"], "unit": "patches/s",
                           "ms_per_step": x3r["ms_per_step"], "steps": args.steps, "warmup": args.warmup, "final_loss": x3r["final_loss"],
                           "step_enqueue": x3r["step_enqueue"], "roofline": x3r["roofline"], "sustained": x3r.get("sustained"),
                           "arithmetic": X3_ARITHMETIC,
                           "parity": "tests/test_hip_x3.py (2e-6 vs the fp64 oracle per kernel; A/B vs the fp32 MFMA on adversarial operands: "
                                     "f32x3 error 0.29-1.11 x, profiles/r06_x3_adversarial.txt; +-Inf / |x| >= 3.39e38 -> NaN) and "
                                     "tests/test_hip_golden_full.py::test_full_size_network_fp32 (both fp32 modes at ONE set of bounds from a "
                                     "ten-draw seed spread, profiles/r06_golden_seed_spread.txt)"}
    if world == 1 and not bf16 and not args.no_c2 and args.patch == 128 and args.channels == 1:
        # BASELINE configs[1]: 64^3 patch, 1 modality, 2 classes, batch 2, fp32 -- the same measurement on a second model
        c2 = measure(args, 64, 2, 1, 2, "fp32", rank, local, world)
        out["c2_64cube_b2"] = {"metric": "training patches/sec (64^3x1ch fp32, batch 2), 1 GPU (BASELINE configs[1])", "value": c2["value"], "unit": "patches/s",
                               "ms_per_step": c2["ms_per_step"], "batch": 2, "patch": 64, "dtype": "f32", "steps": args.steps, "warmup": args.warmup,
                               "final_loss": c2["final_loss"], "step_enqueue": c2["step_enqueue"]}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        out["summary"] = summary(out)            # LAST key: survives a tail-truncated log (VERDICT r5 #7a)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
