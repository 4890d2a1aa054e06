#!/usr/bin/env python3
"""Evaluation driver: the hot loop of `/root/reference/test.py:162-265` on synthetic clips.

  python -m motif_amd.test [-opt options/test.yml] [--clips 4] [--lr 180 320] [--times 7]

Reproduces what the reference's driver does around the model: zero-pad LQ frames to a multiple of 4
(test.py:168-175), `scale` handling (176-182), `feed_data` -> `test()`, crop `fake_H[..., :H, :W]`,
Y-channel PSNR per frame (212-238) and SSIM (245-249), log line.  Clips come from `motif_amd.data.synthetic`, or with
`--frames GT_ROOT [--lq LQ_ROOT]` from PNG folders through `motif_amd.data.folder_dataset` (the sample-dict contract of
`data/Adobe_test_3.py`, pixels decoded on the GPU); weights from `path.pretrain_model_G` when given, else the seeded
key-hashed generator.  With `--launcher pytorch` clips are sharded over the
ranks (one process per GPU) and the per-frame PSNR vectors are gathered to rank 0.
"""
import argparse
import logging
import os

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-opt", type=str, default=None, help="YAML options file (test.yml format)")
    ap.add_argument("--launcher", choices=["none", "pytorch"], default="none")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend of --launcher pytorch: nccl = RCCL (one GPU per rank, as the reference's init_dist); gloo = "
                         "plumbing runs with fewer GPUs than ranks (ranks share GPUs, the PSNR gather is staged through the host)")
    ap.add_argument("--local_rank", type=int, default=0)
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--lr", type=int, nargs=2, default=[180, 320])
    ap.add_argument("--times", type=int, default=7)
    ap.add_argument("--ssim", action="store_true")
    ap.add_argument("--frames", type=str, default=None, help="folder of HR PNG frame folders (<root>/<video>/<NNN>.png)")
    ap.add_argument("--lq", type=str, default=None, help="folder of the LR frames (same layout); default: --frames")
    ap.add_argument("--dataset-mode", choices=["mid", "arbitrary"], default="mid")
    args = ap.parse_args()

    from . import dist as mdist
    from . import option
    from .data.synthetic import synthetic_sample
    from .models import create_model
    from .utils import util
    from .utils.synth_weights import fill_state_dict

    opt = option.parse(args.opt, is_train=True) if args.opt else option.default_opt()
    opt = option.dict_to_nonedict(opt)
    opt["dist"] = args.launcher == "pytorch"
    if opt["dist"]:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", args.local_rank))
        if args.backend == "gloo":
            local %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        dist.init_process_group(backend=args.backend)
    rank, world = mdist.world()
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, format="%(asctime)s %(message)s")
    logger = logging.getLogger("base")
    torch.manual_seed(0)

    model = create_model(opt)
    if not opt["path"]["pretrain_model_G"]:
        fill_state_dict(model.netG)
    scale = opt["scale"]
    psnrs = []
    dataset = None
    if args.frames:
        from .data.folder_dataset import FolderClipDataset, collate_u8, decode_batch
        dataset = FolderClipDataset({"dataroot_GT": args.frames, "dataroot_LQ": args.lq, "mode": args.dataset_mode})
        args.clips = min(args.clips, len(dataset)) if args.clips else len(dataset)
    mine = mdist.shard_indices(args.clips)
    for clip in mine:
        if dataset is not None:
            data = decode_batch(collate_u8([dataset[clip]]), "cuda")
            args.times = len(data["time"])
        else:
            data = synthetic_sample(args.lr[0], args.lr[1], scale, args.times, seed=clip)
        imgs_in = data["LQs"]
        b, n, c, h, w = imgs_in.size()
        h_n, w_n = int(4 * np.ceil(h / 4)), int(4 * np.ceil(w / 4))
        padded = imgs_in.new_zeros(b, n, c, h_n, w_n)
        padded[:, :, :, 0:h, 0:w] = imgs_in
        data["LQs"] = padded
        H, W = data["GT"].shape[3], data["GT"].shape[4]
        data["scale"] = [[h_n * scale], [w_n * scale]]
        model.feed_data(data)
        model.test()
        nfr = model.real_H.shape[1] - 2
        real_H = model.real_H[:, 1:-1].reshape(b * nfr, 3, H, W)
        model.ensure_finite()                # f16x2's range guard (VideoSRBaseModel.ensure_finite): a clip beyond fp16's range is rendered again with bf16x3
        fake_H = model.fake_H[:, :, :, 0:H, 0:W].permute(1, 0, 2, 3, 4).reshape(b * nfr, 3, H, W)
        p = util.y_psnr_per_frame(real_H, fake_H)
        psnrs.append(p)
        msg = "clip %d: Y-PSNR anchor %.3f inter %.3f" % (clip, p[0], float(np.mean(p[1:-1])) if len(p) > 2 else p[-1])
        if args.ssim:
            ry, fy = util.rgb_to_y(real_H).cpu().numpy() * 255.0, util.rgb_to_y(fake_H).cpu().numpy() * 255.0
            ss = [util.calculate_ssim(ry[i], fy[i]) for i in range(len(ry))]
            msg += " ssim %.4f" % float(np.mean(ss[:-1]))      # test.py:248: the clip's figure leaves the last frame out (sic)
        logger.info(msg + " lr %s" % model.get_current_learning_rate())
    local = torch.tensor(np.stack(psnrs) if psnrs else np.zeros((0, args.times)), dtype=torch.float32,
                         device="cuda" if opt["dist"] else "cpu")
    allp = mdist.gather_to_rank0(local, args.clips)
    if rank == 0:
        logger.info("mean Y-PSNR over %d clips: %.4f dB" % (args.clips, float(allp.mean())))
        os.makedirs("psnrs", exist_ok=True)
        np.save(os.path.join("psnrs", opt["name"] + ".npy"), allp.cpu().numpy())


if __name__ == "__main__":
    main()
