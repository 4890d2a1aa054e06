"""YAML options (`/root/reference/option.py:9-68`), reduced to what the inference path reads.

Differences that the build adds on purpose (SURVEY.md §5 "Config"): `gpu_ids: ~` is accepted (the
reference crashes at option.py:13), and `path.root` is not overwritten with a hard-coded home directory
(option.py:45).  `default_opt` builds the same dict without a YAML file (synthetic benchmark / tests).
"""
import os
import os.path as osp
from collections import OrderedDict

import yaml


class NoneDict(dict):
    def __missing__(self, key):
        return None


def dict_to_nonedict(opt):
    if isinstance(opt, dict):
        return NoneDict(**{k: dict_to_nonedict(v) for k, v in opt.items()})
    if isinstance(opt, list):
        return [dict_to_nonedict(v) for v in opt]
    return opt


def parse(opt_path, is_train=True):
    with open(opt_path, mode="r") as f:
        opt = yaml.safe_load(f)
    if opt.get("gpu_ids") is not None:
        os.environ["CUDA_VISIBLE_DEVICES"] = ",".join(str(x) for x in opt["gpu_ids"])
    opt["is_train"] = is_train
    scale = opt.get("scale", 4)
    for phase, dataset in (opt.get("datasets") or {}).items():
        dataset["phase"] = phase.split("_")[0]
        dataset["scale"] = scale
        dataset["data_type"] = "img"
    for key, path in (opt.get("path") or {}).items():
        if path and key != "strict_load":
            opt["path"][key] = osp.expanduser(path)
    opt.setdefault("path", {})
    opt["path"].setdefault("root", os.getcwd())
    root = opt["path"]["root"]
    sub = "experiments" if is_train else "results"
    opt["path"]["experiments_root" if is_train else "results_root"] = osp.join(root, sub, opt["name"])
    opt["path"]["log"] = osp.join(root, sub, opt["name"])
    opt["network_G"]["scale"] = scale
    return opt


def default_opt(scale=4, gpu_ids=(0,), pretrain_model_G=None, name="synthetic", mma=None, which_model_G="Ours", hip_graph=False):
    """The option dict `test.yml` yields (test.yml:1-83) for the `Ours` generator, setting 5."""
    return dict_to_nonedict(OrderedDict(
        name=name, use_tb_logger=False, model="VideoSR_base", distortion="sr", scale=scale,
        gpu_ids=list(gpu_ids) if gpu_ids is not None else None, dist=False, is_train=True,
        network_G=OrderedDict(which_model_G=which_model_G, nf=64, nframes=7, groups=8, front_RBs=5, back_RBs=40, setting=5, scale=scale, mma=mma),
        path=OrderedDict(pretrain_model_G=pretrain_model_G, strict_load=True, models="./saved_checkpoints/", root="./"),
        train=OrderedDict(lr_G=0.0, lr_scheme="CosineAnnealingLR_Restart", beta1=0.9, beta2=0.99, pixel_criterion="cb",
                          pixel_weight=1.0, manual_seed=0),
        logger=OrderedDict(print_freq=1),
        hip_graph=bool(hip_graph),        # MI355X addition: record a clip's launches into a HIP graph and replay it (VideoSR_base_model.py)
    ))
