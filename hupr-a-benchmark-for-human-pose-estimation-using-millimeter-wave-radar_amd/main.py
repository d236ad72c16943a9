#!/usr/bin/env python
"""Entry point with the reference's CLI (main.py:17-41):

    python main.py --config mscsa_prgcn.yaml --dir <name> [--eval] [--visDir d] [--gpuIDs "[0]"]
                   [--seed 0] [-sr N] [--keypoints]

Extra opt-in flags (defaults reproduce the reference): --synthetic_length, --max_steps, --max_epochs.
Under ``python -m torch.distributed.run --nproc-per-node N main.py ...`` it trains data-parallel
(one process per GPU, RCCL all-reduce).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--seed', type=int, default=0, metavar='S', help='random seed (default: 0)')
    parser.add_argument('--dir', type=str, default='test', metavar='B', help='directory of saving/loading')
    parser.add_argument('--visDir', type=str, default='none', metavar='B', help='directory of visualization')
    parser.add_argument('--config', type=str, default='mscsa_prgcn.yaml', metavar='B', help='config file under ./config')
    parser.add_argument('--gpuIDs', default=[0], type=eval, help='IDs of GPUs to use')
    parser.add_argument('--eval', action="store_true")
    parser.add_argument('-sr', '--sampling_ratio', type=int, default=1, help='sampling ratio for training/test (default: 1)')
    parser.add_argument('--keypoints', action='store_true', help='print out the APs of all keypoints')
    parser.add_argument('--pretrained', action='store_true', help='load weights only, start a fresh optimizer')
    parser.add_argument('--synthetic_length', type=int, default=64, help='items per synthetic dataset split')
    parser.add_argument('--max_steps', type=int, default=0, help='stop each epoch after N steps (0 = full epoch)')
    parser.add_argument('--max_epochs', type=int, default=0, help='stop after N epochs (0 = cfg.TRAINING.epochs)')
    return parser.parse_args(argv)


def main(argv=None):
    import torch
    import torch.distributed as dist
    from hupr_amd.config_tree import load_config
    from hupr_amd.tools.run import Runner

    args = parse(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        # control plane (rendezvous, communicator id, gathers of evaluation records) on gloo; the gradient exchange is the
        # C ABI's own RCCL communicator (tools/distributed.py); the nccl half is only used if that has to fall back
        dist.init_process_group("cpu:gloo,cuda:nccl")
    cfg_dir = "./config" if os.path.isdir("./config") else None
    cfg = load_config(args.config, cfg_dir)
    trigger = Runner(args, cfg)
    vis = False if args.visDir == 'none' else True
    if args.eval:
        trigger.loadModelWeight('model_best')
        trigger.eval(visualization=vis)
    else:
        trigger.loadModelWeight('checkpoint')
        trigger.train()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
