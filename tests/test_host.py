"""CPU: host-side logic of the drop-in surface -- C-ABI library loads and exports every declared symbol,
TF variable catalogue, config contract, data contract, optimiser schedule, CLI.  No kernel is launched."""
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from vnet_tensorflow_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "vnet_hip.h")).read()
    declared = set(re.findall(r"\b(vnet_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.lib()                       # binds every symbol; AttributeError if one is missing
    for name in declared:
        assert getattr(L, name) is not None
    assert b"gfx950" in L.vnet_version()
    # pure host queries work without a GPU
    assert L.vnet_packed_weight_floats(0, 125, 16, 16) == 125 * 16 * 16
    assert L.vnet_packed_weight_floats(0, 125, 3, 5) == 125 * 16 * 16
    assert L.vnet_packed_weight_floats(2, 8, 32, 16) == 32 * 128
    assert L.vnet_conv_ws_bytes(5, 0, 1, 0, 16, 16, 1, 128, 128, 128) == 0          # no split-K at full resolution
    assert L.vnet_conv_ws_bytes(5, 0, 1, 0, 256, 256, 1, 8, 8, 8) > 0               # split-K at the bottom level
    assert L.vnet_wgrad_ws_bytes(5, 0, 1, 16, 16, 1, 128, 128, 128) > 0
    # the one struct of the C ABI (vnet_wgrad_job, the grouped filter gradients): the binding's layout is the library's
    import ctypes
    assert L.vnet_wgrad_job_bytes() == ctypes.sizeof(_lib.WgradJob) == 88
    assert L.vnet_conv_wgrad_b16_group(None, 0, None) == 0      # an empty group: no launch
    # tuning switches: read once from the environment, changed only through the ABI
    assert L.vnet_get_option(b"WGRAD_RR") == 1.0 and L.vnet_set_option(b"WGRAD_RR", 2.0) == 1.0 and L.vnet_set_option(b"WGRAD_RR", 1.0) == 2.0
    assert L.vnet_get_option(b"NO_SUCH_OPTION") != L.vnet_get_option(b"NO_SUCH_OPTION")          # NaN
    assert L.vnet_conv_wgrad_b16_group(None, 3, None) == -1


def test_no_cpu_fallback_and_argument_errors():
    from vnet_tensorflow_amd import ops, VnetHipError, _lib
    x = torch.zeros(1, 4, 4, 4, 16)
    with pytest.raises(VnetHipError):
        ops.conv(x, torch.zeros(5, 5, 5, 16, 16), torch.zeros(16), 5, 1)
    with pytest.raises(VnetHipError):
        ops.head_conv(x, torch.zeros(1, 1, 1, 16, 2), torch.zeros(2))
    with pytest.raises(VnetHipError):
        ops.softmax_loss(torch.zeros(1, 4, 4, 4, 2), torch.zeros(1, 4, 4, 4, 1, dtype=torch.int32))
    L = _lib.lib()
    assert L.vnet_pack_weights(0, None, None, 125, 16, 16, None) == -1           # VNET_E_BADARG, no launch
    assert L.vnet_conv_fwd(3, 0, 1, 0, None, 16, None, 0, None, None, None, 16, None, 0, 1, 4, 4, 4, 4, 4, 4, None, 0, None) == -1
    with pytest.raises(SystemExit):
        ops.parse_loss("dice")


@pytest.mark.parametrize("variant,cin,K", [("networks", 1, 2), ("networks", 4, 5), ("legacy", 1, 2), ("legacy", 2, 3)])
def test_variable_names_match_oracle(variant, cin, K):
    """The mirrored networks create exactly the TF variables (names, shapes, creation order) of the restatement."""
    from oracle import vnet_oracle as O
    from vnet_tensorflow_amd import networks, VNet
    ps = O.ParamStore(rng=np.random.default_rng(0))
    ref = O.VNetOracle(K, 0.0, 8, 3, (1, 2, 3), 2, "prelu", variant, ps)
    ref.GetNetwork(np.zeros((1, 8, 8, 8, cin)))
    if variant == "networks":
        net = networks.VNet(K, 0.0, 8, 3, (1, 2, 3), 2, True, "prelu", device="cpu").build((1, 8, 8, 8, cin))
    else:
        net = VNet.VNet(K, 1.0, 8, 3, (1, 2, 3), 2, True, "prelu", device="cpu").build((1, 8, 8, 8, cin))
    got = [(n, tuple(p.shape)) for n, p in net.named_parameters()]
    want = [(n, tuple(ps.vars[n].v.shape)) for n in ps.order]
    assert got == want
    assert set(net.variables.buffers) == set(ps.state)


def test_full_width_catalogue_counts():
    from vnet_tensorflow_amd import networks
    net = networks.VNet(2, 0.01, 16, 4, (1, 2, 3, 3), 3, True, "prelu", device="cpu").build((1, 32, 32, 32, 1))
    assert sum(p.numel() for p in net.parameters()) == 43940486
    assert sum(b.numel() for b in net.variables.buffers.values()) == 6532
    assert len(net.parameters()) + len(net.variables.buffers) == 241
    w = dict(net.named_parameters())["vnet/decoder/level_1/up_convolution/weights"]
    assert tuple(w.shape) == (2, 2, 2, 16, 32)
    # xavier-uniform limit of layers2.py:19: sqrt(6 / (prod(k) * (Cin + Cout)))
    w5 = dict(net.named_parameters())["vnet/encoder/level_1/conv_1/weights"]
    lim = np.sqrt(6.0 / (125 * 32))
    assert float(w5.abs().max()) <= lim and float(w5.abs().max()) > 0.9 * lim
    with pytest.raises(AssertionError):
        networks.VNet(2, num_levels=3, num_convolutions=(1, 2))


def _config(tmp, **over):
    cfg = {"TrainingSetting": {"Data": {"TrainingDataDirectory": str(tmp), "TestingDataDirectory": str(tmp),
                                        "ImageFilenames": ["image.npy"], "LabelFilename": "label.npy"},
                               "SegmentationClasses": [0, 2], "BatchSize": 2, "PatchShape": [8, 8, 8],
                               "Networks": {"Name": "VNet", "Dropout": 0.01, "NumChannel": 4, "NumLevels": 2,
                                            "NumCovolutions": [1, 2], "BottomConvolutions": 1},
                               "Optimizer": {"Name": "Adam", "InitialLearningRate": 1e-2, "Momentum": 0.8,
                                             "Decay": {"Factor": 0.99, "Steps": 100}},
                               "Loss": {"Name": "weighted_sorensen", "Weights": [0.1, 1.0], "Alpha": 1}},
           "EvaluationSetting": {"Stride": [4, 4, 4], "BatchSize": 3}}
    cfg["TrainingSetting"].update(over)
    return cfg


def test_read_config_contract(tmp_path):
    from vnet_tensorflow_amd.model import image2label
    m = image2label(None, _config(tmp_path), device="cpu", verbose=False)
    m.read_config()
    assert m.num_convolutions == [1, 2]            # shipped typo "NumCovolutions" accepted
    assert m.input_channel_num == 1 and m.output_channel_num == 2 and m.dimension == 3
    assert m.momentum == 0.8 and m.test_step == 100 and m.max_itr == 10 ** 12
    assert m.evaluate_stride == [4, 4, 4] and m.evaluate_batch == 3
    m2 = image2label(None, _config(tmp_path, PatchShape=[8, 8]), device="cpu", verbose=False)
    m2.read_config()
    with pytest.raises(SystemExit):
        m2.build_model_graph()                       # 2-D is out of scope, fails loudly
    cfg3 = _config(tmp_path)
    cfg3["TrainingSetting"]["Networks"]["Name"] = "UNet"
    m3 = image2label(None, cfg3, device="cpu", verbose=False)
    m3.read_config()
    with pytest.raises(SystemExit):
        m3.build_model_graph()
    # ComputeDtype 'bf16' (bf16 storage) needs NumChannel = 8 * 2^k: refused when the config is read, not at the first forward
    cfg4 = _config(tmp_path)
    cfg4["TrainingSetting"]["ComputeDtype"] = "bf16"
    cfg4["TrainingSetting"]["Networks"]["NumChannel"] = 12
    with pytest.raises(SystemExit, match="NumChannel"):
        image2label(None, cfg4, device="cpu", verbose=False).read_config()
    cfg4["TrainingSetting"]["ComputeDtype"] = "fp32_split3"                 # (fp32 tensors: any width)
    image2label(None, cfg4, device="cpu", verbose=False).read_config()
    cfg4["TrainingSetting"]["ComputeDtype"] = "bf16_operands"               # round 2's mode, retired in round 5
    with pytest.raises(SystemExit, match="retired"):
        image2label(None, cfg4, device="cpu", verbose=False).read_config()
    cfg4["TrainingSetting"]["ComputeDtype"] = "fp16"
    with pytest.raises(SystemExit, match="ComputeDtype"):
        image2label(None, cfg4, device="cpu", verbose=False).read_config()


def test_shipped_reference_style_config_parses(tmp_path):
    """A config written like the reference's configs/config_sample.json (typo included) is accepted."""
    from vnet_tensorflow_amd.model import image2label
    cfg = {"ProjectName": "x", "TrainingSetting": {
        "Data": {"TrainingDataDirectory": "./data/training", "TestingDataDirectory": "./data/testing",
                 "ImageFilenames": ["image.nii"], "LabelFilename": "label.nii"},
        "Restore": True, "SegmentationClasses": [0, 1, 2], "LogDir": "./tmp/log", "CheckpointDir": "./tmp/ckpt",
        "BatchSize": 1, "PatchShape": [64, 64, 64], "ImageLog": False, "Testing": True, "TestStep": 30, "Epoches": 99999,
        "MaxIterations": 15000, "LogInterval": 50,
        "Networks": {"Name": "VNet", "Dropout": 0.01, "NumChannel": 16, "NumLevels": 4, "NumCovolutions": [1, 2, 3, 3],
                     "BottomConvolutions": 3},
        "Loss": {"Name": "weighted_sorensen", "Weights": [0.01, 0.1, 1], "Alpha": 1},
        "Optimizer": {"Name": "Adam", "InitialLearningRate": 1e-2, "Momentum": 0.9, "Decay": {"Factor": 0.99, "Steps": 100}},
        "Spacing": [0.75, 0.75, 0.75], "DropRatio": 0.01, "MinPixel": 30, "Pipeline": "./pipeline/pipeline3D.yaml"},
        "EvaluationSetting": {"Data": {"EvaluateDataDirectory": "./data/evaluate", "ImageFilenames": ["image.nii"],
                                       "LabelFilename": "label_tf.nii.gz", "ProbabilityFilename": "probability_tf.nii.gz"},
                              "CheckpointPath": "./tmp/ckpt/checkpoint-23125", "Stride": [64, 64, 64], "BatchSize": 10,
                              "ProbabilityOutput": True, "Pipeline": "./pipeline/pipeline3D.yaml"}}
    m = image2label(None, json.loads(json.dumps(cfg)), device="cpu", verbose=False)
    m.read_config()
    assert m.loss_weights == [0.01, 0.1, 1] and m.max_itr == 15000 and m.evaluate_probability_output is True


def test_data_contract(tmp_path):
    from vnet_tensorflow_amd import data
    vol = np.random.default_rng(0).integers(0, 5, (6, 7, 8)).astype(np.int16)
    p = str(tmp_path / "v.nii")
    data.write_nifti(p, vol, (0.5, 0.75, 1.0))
    back, hdr = data.read_nifti(p)
    assert (back == vol).all() and hdr["dim"] == (6, 7, 8) and np.allclose(hdr["pixdim"], (0.5, 0.75, 1.0))
    lfs = tmp_path / "stub.nii"
    lfs.write_text("version https://git-lfs.github.com/spec/v1\noid sha256:0\nsize 1\n")
    with pytest.raises(ValueError):
        data.read_nifti(str(lfs))                    # the reference's sample volumes are LFS pointers
    assert (data.remap_labels(np.array([0, 2, 7, 2]), [0, 2]) == [0, 1, 0, 1]).all()
    img, lab = data.synthetic_case((16, 16, 16), 2, 3, 5)
    assert img.shape == (16, 16, 16, 2) and img.dtype == np.float32 and img.min() >= 0 and img.max() <= 255
    assert lab.dtype == np.int32 and set(np.unique(lab)) <= {0, 1, 2}
    ci, cl = data.random_crop(img, lab, (8, 20, 8), np.random.default_rng(0))
    assert ci.shape == (8, 20, 8, 2) and cl.shape == (8, 20, 8)
    for c in range(3):
        d = tmp_path / ("case%d" % c)
        d.mkdir()
        a, b = data.synthetic_case((12, 12, 12), 1, 2, c)
        np.save(d / "image.npy", a[..., 0])
        np.save(d / "label.npy", b * 2)
    ds = data.VolumeDataset(str(tmp_path), ["image.npy"], "label.npy", [0, 2], (8, 8, 8), 2, train=True)
    batches = list(ds)
    assert len(batches) == 1                          # drop_remainder=True (model.py:293)
    x, y = batches[0]
    assert x.shape == (2, 8, 8, 8, 1) and x.dtype == np.float32 and y.shape == (2, 8, 8, 8, 1) and y.dtype == np.int32
    assert set(np.unique(y)) <= {0, 1}


def test_lr_schedule_and_flat_layout():
    from vnet_tensorflow_amd import optim
    assert np.isclose(optim.exponential_decay(1e-2, 250, 100, 0.99), 1e-2 * 0.99 ** 2.5)
    ps = [("a", torch.nn.Parameter(torch.arange(5.0))), ("b", torch.nn.Parameter(torch.ones(2, 3))),
          ("c", torch.nn.Parameter(torch.full((7,), 2.0)))]
    flat = optim.FlatParams(ps)
    assert flat.names == ["c", "b", "a"]              # reverse creation order = gradient production order
    assert flat.offsets == [0, 8, 16] and flat.numel == 24
    assert all(o % 4 == 0 for o in flat.offsets)
    ps[0][1].grad.add_(1.0)
    assert float(flat.grad[16:21].sum()) == 5.0       # .grad is a view of the flat gradient buffer
    (ps[1][1] * 3).sum().backward()
    assert float(flat.grad[8:14].sum()) == 18.0       # autograd accumulates in place into the flat buffer
    flat.zero_grad()
    assert float(flat.grad.abs().sum()) == 0.0
    bk = flat.buckets(bucket_bytes=32)
    assert bk[0][0] == 0 and bk[-1][1] == flat.numel and [b[2] for b in bk] == sorted(b[2] for b in bk)
    with pytest.raises(SystemExit):
        optim.make_optimizer("RMSProp", flat)


def test_cli_parser():
    from vnet_tensorflow_amd.main import get_parser
    a = get_parser(["-p", "evaluate", "--config_json", "c.json", "--gpu", "0,1", "-v"])
    assert a.phase == "evaluate" and a.config_json == "c.json" and a.gpu == "0,1" and a.verbose
    assert get_parser([]).phase == "train"


def test_plain_xent_exits_like_the_reference():
    """model.py:495-560: `if name == "xent"` is followed by a separate if/elif chain whose else is sys.exit -> the reference
    exits with "Invalid loss function" for Loss.Name == "xent".  Kept; Loss.AllowPlainXent (extension) opts out."""
    import pytest
    from vnet_tensorflow_amd.model import image2label
    cfg = {"TrainingSetting": {"Data": {"TrainingDataDirectory": "", "TestingDataDirectory": "", "ImageFilenames": ["a"], "LabelFilename": "l"},
                               "SegmentationClasses": [0, 1], "BatchSize": 1, "PatchShape": [16, 16, 16],
                               "Networks": {"Name": "VNet", "Dropout": 0.0, "NumChannel": 4, "NumLevels": 2, "NumConvolutions": [1, 1],
                                            "BottomConvolutions": 1},
                               "Optimizer": {"Name": "Adam", "InitialLearningRate": 1e-3, "Decay": {"Factor": 0.99, "Steps": 100}},
                               "Loss": {"Name": "xent"}}}
    m = image2label(None, cfg, verbose=False)
    m.read_config()
    with pytest.raises(SystemExit, match="Invalid loss function"):
        m.build_model_graph()
    cfg["TrainingSetting"]["Loss"]["Name"] = "no_such_loss"
    m = image2label(None, cfg, verbose=False)
    m.read_config()
    with pytest.raises(SystemExit, match="Invalid loss function"):
        m.build_model_graph()
    cfg["TrainingSetting"]["Loss"] = {"Name": "xent", "AllowPlainXent": True}
    m = image2label(None, cfg, verbose=False)
    m.read_config()
    m._validate_loss()                       # no exit


def test_tf_metrics_auc_restatement_known_answers():
    """oracle.tf_metrics_auc against hand-derivable cases of TF's thresholded ROC (200 thresholds, trapezoid)."""
    from oracle import vnet_oracle as O
    lab = np.array([0, 0, 1, 1])
    assert abs(O.tf_metrics_auc(lab, np.array([0.1, 0.2, 0.8, 0.9])) - 1.0) < 1e-6            # perfectly separated
    assert abs(O.tf_metrics_auc(lab, np.array([0.9, 0.8, 0.2, 0.1])) - 0.0) < 1e-6            # perfectly wrong
    assert abs(O.tf_metrics_auc(lab, np.array([0.5, 0.5, 0.5, 0.5])) - 0.5) < 1e-6            # one ROC step: area 1/2
    # TF's doc example: labels [0,0,1,1], predictions [0.1,0.4,0.35,0.8] -> 0.75
    assert abs(O.tf_metrics_auc(lab, np.array([0.1, 0.4, 0.35, 0.8])) - 0.75) < 1e-6


def test_make_batch_into_caller_buffers_equals_stacked_batches():
    """The prefetch threads crop straight into pinned host tensors (VolumeDataset.make_batch(out=...)): same batch, same
    dtypes as the stacking path, with and without a transform pipeline."""
    import numpy as np
    from vnet_tensorflow_amd import data as vd
    ds = vd.VolumeDataset("synthetic", ["a.npy", "b.npy"], "l.npy", [0, 1, 2], [12, 16, 8], 3, train=True,
                          synthetic={"Cases": 6, "Shape": [20, 18, 12]})
    (cases, seeds), = ds.epoch_plan()[:1]
    ref_i, ref_l = ds.make_batch(cases, seeds)
    si, sl = ds.batch_shapes()
    assert tuple(ref_i.shape) == si and tuple(ref_l.shape) == sl
    oi, ol = np.full(si, np.nan, np.float32), np.full(sl, -7, np.int32)
    got = ds.make_batch(cases, seeds, out=(oi, ol))
    assert got[0] is oi and got[1] is ol
    assert np.array_equal(oi, ref_i) and np.array_equal(ol, ref_l) and ol.dtype == np.int32


def test_bench_refuses_more_ranks_than_devices():
    """`python bench.py --gpus N` without a launcher starts its own ranks (tests/test_hip_dp.py on the GPU); with fewer visible
    devices than ranks it must refuse with a non-zero code and print no JSON line -- never measure one GPU and call it N."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this host has the devices")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "VNET_DIST_BACKEND")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "device(s) visible" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
