"""-m gpu: the caller of the hot path (reference model.py:632-815 / 1131-1242) end to end on the HIP library:
config.json -> image2label.train() (synthetic volumes) -> checkpoints (checkpoint-<step>, checkpoint-latest)
-> restore -> evaluate() sliding window writing label / probability volumes; plus main.py's CLI."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(tmp, **train_over):
    cfg = {"TrainingSetting": {
        "Data": {"TrainingDataDirectory": "synthetic", "TestingDataDirectory": "synthetic", "ImageFilenames": ["image.npy"],
                 "LabelFilename": "label.npy", "Synthetic": {"Cases": 4, "Shape": [20, 20, 20]}},
        "Restore": False, "SegmentationClasses": [0, 1], "LogDir": str(tmp / "log"), "CheckpointDir": str(tmp / "ckpt"),
        "BatchSize": 2, "PatchShape": [16, 16, 16], "Testing": True, "TestStep": 2, "Epoches": 2, "MaxIterations": 100,
        "LogInterval": 3,
        "Networks": {"Name": "VNet", "Dropout": 0.05, "NumChannel": 4, "NumLevels": 2, "NumCovolutions": [1, 2], "BottomConvolutions": 1},
        "Loss": {"Name": "sorensen", "Weights": [], "Alpha": 1},
        "Optimizer": {"Name": "Adam", "InitialLearningRate": 1e-2, "Decay": {"Factor": 0.99, "Steps": 100}}},
        "EvaluationSetting": {"Data": {"EvaluateDataDirectory": str(tmp / "eval"), "ImageFilenames": ["image.npy"],
                                       "LabelFilename": "label_out.npy", "ProbabilityFilename": "prob_out.npy"},
                              "CheckpointPath": str(tmp / "ckpt" / "checkpoint-4"), "Stride": [8, 8, 8], "BatchSize": 2,
                              "ProbabilityOutput": True}}
    cfg["TrainingSetting"].update(train_over)
    return cfg


def test_train_checkpoint_restore_evaluate(tmp_path, dev):
    from vnet_tensorflow_amd.model import image2label
    from vnet_tensorflow_amd.data import synthetic_case
    np.random.seed(0)
    m = image2label(None, _cfg(tmp_path), device=dev, verbose=False)
    m.train()
    assert m.global_step == 4 and m.start_epoch == 2            # 2 epochs x (4 cases / batch 2)
    assert np.isfinite(m.last_loss) and 0.0 < m.last_loss < 1.0
    ck = sorted(os.listdir(tmp_path / "ckpt"))
    assert "checkpoint-latest" in ck and "checkpoint-4" in ck and "checkpoint-3" in ck   # LogInterval=3 and epoch ends
    sd = torch.load(tmp_path / "ckpt" / "checkpoint-4", weights_only=False)
    assert "vnet/encoder/level_1/conv_1/weights" in sd["variables"] and sd["global_step"] == 4
    assert "vnet/input_layer/batch_normalization/moving_mean" in sd["variables"]

    # resume: Restore=true picks up checkpoint-latest and continues the step / epoch counters
    m2 = image2label(None, _cfg(tmp_path, Restore=True, Epoches=3), device=dev, verbose=False)
    m2.train()
    assert m2.global_step == 6 and m2.start_epoch == 3
    w1 = sd["variables"]["vnet/output_layer/weights"]
    assert not torch.equal(w1, dict(m2.network.state_dict())["vnet/output_layer/weights"].cpu())

    # evaluate: sliding window over a 24^3 volume, writes label + per-class probability arrays
    case = tmp_path / "eval" / "case0"
    case.mkdir(parents=True)
    img, _ = synthetic_case((24, 24, 24), 1, 2, 5)
    np.save(case / "image.npy", img[..., 0])
    m3 = image2label(None, _cfg(tmp_path), device=dev, verbose=False)
    m3.evaluate()
    lab = np.load(case / "label_out.npy")
    p0, p1 = np.load(case / "prob_out_0.npy"), np.load(case / "prob_out_1.npy")
    assert lab.shape == (24, 24, 24) and set(np.unique(lab)) <= {0, 1}
    assert np.allclose(p0 + p1, 1.0, atol=1e-5) and ((p1 > p0) == (lab == 1)).mean() > 0.999


@pytest.mark.parametrize("compute,cin,K", [("fp32", 1, 2), ("bf16", 4, 5)])
def test_loss_decreases_on_fixed_batch(dev, compute, cin, K):
    """Sanity of the whole fwd/bwd/Adam loop: 12 steps on one synthetic batch reduce the Dice loss -- in the
    reference's fp32 arithmetic and in the bf16-compute mode of BASELINE config C5 (4 modalities, 5 classes; this also
    runs the batched bf16 filter repack after every optimiser step)."""
    from vnet_tensorflow_amd import ops
    from vnet_tensorflow_amd.model import image2label
    from oracle.vnet_oracle import synthetic_batch
    import pathlib
    np.random.seed(1)
    cfg = _cfg(pathlib.Path("/tmp"), ComputeDtype=compute, SegmentationClasses=list(range(K)))
    cfg["TrainingSetting"]["Data"]["ImageFilenames"] = ["image%d.npy" % i for i in range(cin)]
    if compute == "bf16":
        cfg["TrainingSetting"]["Networks"]["NumChannel"] = 8          # bf16 storage: 16-byte channel units
    m = image2label(None, cfg, device=dev, verbose=False)
    try:
        m.read_config()
        m.build_model_graph()
        assert m.ctx.compute["name"] == compute and ops.get_compute_dtype() == "fp32"      # per-model state (round 4), not a process global
        m._setup_training()
        x, lab = synthetic_batch(2, 16, cin, K, seed=11)
        xt, lt = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev)
        losses = [float(m.train_step(xt, lt, dropout=0.0)) for _ in range(12)]
    finally:
        ops.set_compute_dtype("fp32")
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.05, losses


@pytest.mark.parametrize("P", [16, 32])
def test_param_grad_stream_is_bit_identical(dev, P, monkeypatch):
    """Filter/bias gradients on their own HIP stream (the default in image2label) vs everything on one stream: the same
    kernels on the same data, so after 4 optimiser steps every parameter and moving statistic must match bit for bit
    (a missing cross-stream dependency would show up as a stale or half-written gradient)."""
    from vnet_tensorflow_amd import ops
    from vnet_tensorflow_amd.model import image2label
    from oracle.vnet_oracle import synthetic_batch
    import pathlib
    x, lab = synthetic_batch(2, P, 1, 2, seed=21)
    xt, lt = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev)
    res = []
    monkeypatch.setenv("VNET_STEP_GRAPH", "0")      # the replayed step graph is captured single-stream; this is the eager two-stream path
    for on in ("1", "0"):
        monkeypatch.setenv("VNET_PARAM_GRAD_STREAM", on)
        np.random.seed(3)
        cfg = _cfg(pathlib.Path("/tmp"), PatchShape=[P] * 3)
        cfg["TrainingSetting"]["Networks"].update(NumChannel=8, NumLevels=3, NumConvolutions=[1, 2, 2])
        m = image2label(None, cfg, device=dev, verbose=False)
        m.read_config()
        m.build_model_graph()
        m._setup_training()
        assert m.ctx.pg["on"] == (on == "1") and not ops._PG["on"]      # per-model state (round 4): the default context is untouched
        # the side stream is held back ~0.2 ms per layer: anything the main stream does to a tensor the filter gradient still
        # needs (recycling it, accumulating into it in place) now lands BEFORE the filter gradient runs
        monkeypatch.setitem(ops._PG, "test_delay", 400000 if on == "1" else 0)
        losses = [float(m.train_step(xt, lt, dropout=0.0)) for _ in range(4)]
        torch.cuda.synchronize()
        res.append((losses, m.flat.data.clone(), {k: v.clone() for k, v in m.network.state_dict().items()}))
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1])
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k


def test_main_cli(tmp_path):
    cfg = _cfg(tmp_path, Epoches=1)
    path = tmp_path / "config.json"
    path.write_text(json.dumps(cfg))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "-p", "train", "--config_json", str(path), "--gpu", "0"],
                         capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Segmentation training loss" in out.stdout and os.path.exists(tmp_path / "ckpt" / "checkpoint-latest")


def test_fp32_and_bf16_models_coexist_and_interleave(dev):
    """ComputeDtype, the parameter-gradient stream and the packed-filter registry are per-model state (an
    ops.OpsContext owned by image2label), not process globals.  A fp32 and a bf16-storage model are built in ONE process and their
    training steps interleaved (eager steps, the capture of each step graph and graph replays all fall between the other model's
    steps); each must produce, bit for bit, the losses and parameters of the same model trained alone."""
    from vnet_tensorflow_amd import ops
    from vnet_tensorflow_amd.model import image2label
    from oracle.vnet_oracle import synthetic_batch
    import pathlib
    P, NSTEP = 16, 6
    x, lab = synthetic_batch(2, P, 1, 2, seed=33)
    xt, lt = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev)

    def build(compute):
        np.random.seed(5)
        cfg = _cfg(pathlib.Path("/tmp"), PatchShape=[P] * 3)
        cfg["TrainingSetting"]["Networks"].update(NumChannel=8, NumLevels=3, NumConvolutions=[1, 2, 2])
        cfg["TrainingSetting"]["ComputeDtype"] = compute
        m = image2label(None, cfg, device=dev, verbose=False)
        m.read_config(); m.build_model_graph(); m._setup_training()
        return m

    alone = {}
    for compute in ("fp32", "bf16"):
        m = build(compute)
        alone[compute] = ([float(m.train_step(xt, lt, dropout=0.0)) for _ in range(NSTEP)], m.flat.data.clone())
        del m
    a, b = build("fp32"), build("bf16")
    assert a.ctx is not b.ctx and ops.get_compute_dtype() == "fp32"
    la, lb = [], []
    for _ in range(NSTEP):
        la.append(float(a.train_step(xt, lt, dropout=0.0)))
        lb.append(float(b.train_step(xt, lt, dropout=0.0)))
    torch.cuda.synchronize()
    assert a.ctx.compute["name"] == "fp32" and b.ctx.compute["name"] == "bf16" and ops.get_compute_dtype() == "fp32"
    assert la == alone["fp32"][0] and torch.equal(a.flat.data, alone["fp32"][1])
    assert lb == alone["bf16"][0] and torch.equal(b.flat.data, alone["bf16"][1])
    assert alone["fp32"][0] != alone["bf16"][0]                    # (the two really compute differently)
