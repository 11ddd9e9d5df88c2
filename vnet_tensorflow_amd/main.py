"""CLI mirror of the reference's main.py (main.py:13-83): -p {train,evaluate} --config_json F --gpu IDS [-v]."""
import argparse
import json
import os
import sys


def str2bool(v):
    return v.lower() in ("yes", "true", "t", "1")


def get_parser(argv=None):
    parser = argparse.ArgumentParser(description='MI355X-native V-Net segmentation (drop-in for vnet-tensorflow main.py).')
    parser.register('type', 'bool', str2bool)
    parser.add_argument('-v', '--verbose', dest='verbose', help='Show verbose output', action='store_true')
    parser.add_argument('-p', '--phase', dest='phase', help='Training phase (default= train)', choices=['train', 'evaluate'],
                        default='train', metavar='[train evaluate]')
    parser.add_argument('--config_json', dest='config_json', help='JSON file for model configuration', type=str,
                        default='config.json', metavar='FILENAME')
    parser.add_argument('--gpu', dest='gpu', default='0', type=str, help='Select GPU device(s) (default = 0)', metavar='GPU_IDs')
    args = parser.parse_args(argv)
    if args.verbose:
        for key, val in sorted(vars(args).items()):
            print("{} = {}".format(str(key), str(val)))
    return args


def main(args):
    # select gpu (reference main.py:62 sets CUDA_VISIBLE_DEVICES; HIP honours HIP_VISIBLE_DEVICES too).
    # Under torchrun every rank sees all GPUs and picks LOCAL_RANK, so only restrict single-process runs.
    if "LOCAL_RANK" not in os.environ:
        os.environ["HIP_VISIBLE_DEVICES"] = str(args.gpu)
    with open(args.config_json) as config_json:
        config = json.load(config_json)
    from .model import image2label
    model = image2label(None, config)
    if args.phase == "train":
        model.train()
    elif args.phase == "evaluate":
        model.evaluate()
    else:
        sys.exit("Invalid training phase")


if __name__ == "__main__":
    main(get_parser())
