"""CPU: the array restatements of the reference's index / intensity sample transforms (pipeline/NiftiDataset3D.py) --
known answers and invariants; the YAML pipeline loader (model.py:340-372 schema)."""
import numpy as np
import pytest

from vnet_tensorflow_amd import transforms as T
from vnet_tensorflow_amd import data


def _sample(shape=(20, 18, 16), C=2, seed=0, blob=True):
    rng = np.random.default_rng(seed)
    img = rng.normal(100.0, 30.0, size=shape + (C,)).astype(np.float32)
    lab = np.zeros(shape, dtype=np.int32)
    if blob:
        lab[5:9, 6:10, 3:8] = 1
        lab[14:17, 2:5, 10:13] = 2
    return {'image': img, 'label': lab}


def test_statistical_normalization_known_answer():
    s = _sample()
    out = T.StatisticalNormalization(2.5)(s)['image']
    ch = s['image'][..., 0].astype(np.float64)
    mu, sd = ch.mean(), ch.std(ddof=1)                       # ITK StatisticsImageFilter: unbiased sigma
    ref = np.clip((ch - (mu - 2.5 * sd)) / (5.0 * sd) * 255.0, 0, 255)
    assert np.allclose(out[..., 0], ref, atol=1e-3) and out.min() >= 0 and out.max() <= 255
    assert abs(out[..., 0].mean() - 127.5) < 1.0               # the window is centred on the mean


def test_manual_and_extremum_normalization():
    s = {'image': np.array([-10., 0., 50., 100., 300.], dtype=np.float32).reshape(5, 1, 1, 1), 'label': np.zeros((5, 1, 1), np.int32)}
    out = T.ManualNormalization(0, 100)(s)['image'].ravel()
    assert np.allclose(out, [0, 0, 127.5, 255, 255])
    out = T.ExtremumNormalization(0.0)(s)['image'].ravel()
    assert np.allclose(out, (np.array([-10, 0, 50, 100, 300.]) + 10) / 310 * 255)
    out = T.Normalization()(s)['image'].ravel()
    assert out.min() == 0 and out.max() == 255


def test_padding_keeps_voxels_and_pads_high_side():
    s = _sample((10, 18, 7))
    out = T.Padding([16, 16, 16])(s)
    assert out['label'].shape == (16, 18, 16) and out['image'].shape == (16, 18, 16, 2)        # an axis already larger is kept
    assert np.array_equal(out['image'][:10, :, :7], s['image']) and out['image'][10:].sum() == 0 and out['label'][:, :, 7:].sum() == 0
    assert T.Padding(7)(s) is s


def test_random_flip_is_all_or_nothing():
    s = _sample()
    seen = set()
    for seed in range(8):
        out = T.RandomFlip([True, False, True])(s, np.random.default_rng(seed))
        flipped = np.array_equal(out['image'], s['image'][::-1, :, ::-1]) and np.array_equal(out['label'], s['label'][::-1, :, ::-1])
        same = np.array_equal(out['image'], s['image'])
        assert flipped or same
        seen.add(flipped)
    assert seen == {True, False}


def test_random_noise_statistics():
    s = _sample((32, 32, 32), C=1)
    out = T.RandomNoise(5)(s, np.random.default_rng(0))
    d = out['image'] - s['image']
    assert abs(d.mean()) < 0.1 and abs(d.std() - 5.0) < 0.1 and np.array_equal(out['label'], s['label'])


def test_random_crop_respects_min_pixel():
    s = _sample()
    t = T.RandomCrop([8, 8, 8], drop_ratio=0.0, min_pixel=1)
    for seed in range(10):
        out = t(s, np.random.default_rng(seed))
        assert out['label'].shape == (8, 8, 8) and out['image'].shape == (8, 8, 8, 2)
        assert (out['label'] > 0).sum() >= 1                     # never an empty window when drop_ratio = 0
    empty = _sample(blob=False)
    out = T.RandomCrop(8, drop_ratio=1.0, min_pixel=1)(empty, np.random.default_rng(0))          # kept with probability 1
    assert out['label'].sum() == 0
    with pytest.raises(RuntimeError):
        T.RandomCrop(8, drop_ratio=1.5)


def test_confidence_crop2_centres_on_a_component():
    s = _sample((40, 40, 40))
    s['label'][:] = 0
    s['label'][20:24, 10:14, 30:34] = 3                          # one component, bounding box centre (22, 12, 32)
    t = T.ConfidenceCrop2([16, 16, 16], rand_range=0, probability=1.0)
    out = t(s, np.random.default_rng(1))
    # index = box start + ext/2 - out/2 = (14, 4, 24); the last axis hits the border rule: 40 - 24 - 1 < 16 -> 40 - 16 - 1 = 23
    assert np.array_equal(out['label'], s['label'][14:30, 4:20, 23:39])
    assert (out['label'] == 3).sum() == 64
    # probability 0: a random region; random_empty_region: one without label
    out = T.ConfidenceCrop2(16, probability=0.0, random_empty_region=True)(s, np.random.default_rng(2))
    assert out['label'].shape == (16, 16, 16) and out['label'].sum() == 0
    # no label at all -> falls back to a random region
    e = _sample((40, 40, 40), blob=False)
    assert T.ConfidenceCrop2(16, probability=1.0)(e, np.random.default_rng(3))['label'].shape == (16, 16, 16)


def test_pipeline_yaml_and_dataset(tmp_path):
    y = tmp_path / "pipeline3D.yaml"
    y.write_text("""
description: test
preprocess:
  train:
    3D:
      - name: "StatisticalNormalization"
        variables:
          sigma: 2.5
      - name: "Padding"
        variables:
          output_size: [16, 16, 16]
      - name: "ConfidenceCrop2"
        variables:
          output_size: [16, 16, 16]
          rand_range: 2
          probability: 0.8
      - name: "RandomNoise"
  test:
    3D:
      - name: "Padding"
        variables:
          output_size: [16, 16, 16]
      - name: "RandomCrop"
        variables:
          output_size: [16, 16, 16]
""")
    tf = T.build_pipeline(str(y), "train")
    assert [t.name for t in tf] == ['StatisticalNormalization', 'Padding', 'Confidence Crop 2', 'Random Noise']
    ds = data.VolumeDataset("synthetic", ["a.npy"], "l.npy", [0, 1, 2], (16, 16, 16), 2, train=True, seed=1,
                            synthetic={"Cases": 4, "Shape": [24, 20, 18]}, transforms=tf)
    batches = list(ds)
    assert len(batches) == 2
    for img, lab in batches:
        assert img.shape == (2, 16, 16, 16, 1) and lab.shape == (2, 16, 16, 16, 1) and lab.dtype == np.int32
        assert -40 < img.min() and img.max() < 300
    bad = tmp_path / "bad.yaml"
    bad.write_text("preprocess:\n  train:\n    3D:\n      - name: Resample\n        variables: {voxel_size: [1, 1, 1]}\n")
    with pytest.raises(NotImplementedError):
        T.build_pipeline(str(bad), "train")


def test_largest_component_and_volume_threshold():
    """model.py:117-167 post-processing of the predicted label map."""
    from vnet_tensorflow_amd.model import ExtractLargestConnectedComponents, volume_threshold
    lab = np.zeros((12, 12, 12), dtype=np.int64)
    lab[1:3, 1:3, 1:3] = 2            # 8 voxels
    lab[5:9, 5:9, 5:9] = 1            # 64 voxels
    lab[8, 8, 9] = 3                  # face-connected to the big block: 65 voxels, one component across label values
    lab[11, 0, 0] = 1                 # 1 voxel
    lcc = ExtractLargestConnectedComponents(lab)
    assert lcc.dtype == np.uint8 and lcc.sum() == 65 and lcc[6, 6, 6] == 1 and lcc[1, 1, 1] == 0
    vt = volume_threshold(lab, 7.5)
    assert vt.sum() == 65 + 8 and vt[11, 0, 0] == 0
    assert volume_threshold(lab, 7.5, spacing=(0.5, 0.5, 0.5)).sum() == 65           # physical size: 8 voxels = 1.0 < 7.5
    assert ExtractLargestConnectedComponents(np.zeros((4, 4, 4))).sum() == 0
