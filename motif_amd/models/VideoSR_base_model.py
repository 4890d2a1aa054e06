"""`VideoSRBaseModel` -- the shell `test.py` drives (`/root/reference/models/VideoSR_base_model.py`).

Reproduced for the inference path: `feed_data` keys and `scale` default (95-121), `test` time-chunking
into <=3 timestamps with `cat` on dim 0 (169-200, leaves the net in train() mode afterwards, :198),
`load`/`save` (224-231), the Adam + scheduler objects `test.py:272` reads the learning rate from
(`is_train=True` at test.py:311).  Training (`optimize_parameters`) is out of scope.
"""
import logging
from collections import OrderedDict

import torch

from . import networks
from .base_model import BaseModel

logger = logging.getLogger("base")


class VideoSRBaseModel(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        self.rank = -1
        if opt.get("dist"):
            self.rank = torch.distributed.get_rank()
        # gpu_ids: ~ builds the parameter tree on the host (checkpoint plumbing); the forward itself has
        # no CPU route and raises -- the CPU restatement lives in oracle/ as test infrastructure only.
        self.netG = networks.define_G(opt).to(self.device)
        self.net_opt = opt["network_G"]
        self.net_base = self.net_opt["which_model_G"]
        self.load()
        self.log_dict = OrderedDict()
        if self.is_train:
            self.netG.train()
            train_opt = opt["train"]
            params = [v for v in self.netG.parameters() if v.requires_grad]
            self.optimizer_G = torch.optim.Adam(params, lr=train_opt["lr_G"], weight_decay=train_opt.get("weight_decay_G") or 0,
                                                betas=(train_opt["beta1"], train_opt["beta2"]))
            self.optimizers.append(self.optimizer_G)

    def feed_data(self, data, need_GT=True):
        self.var_L = data["LQs"].to(self.device)
        if "time" in data.keys() and "Ours" in self.net_base:
            self.times = [t_.to(self.device) for t_ in data["time"]]
        else:
            self.times = None
        self.scale = data["scale"] if "scale" in data.keys() else 4
        self.testmode = data["test"] if "test" in data.keys() else False
        if need_GT:
            self.real_H = data["GT"].to(self.device)
        self.flows = None
        if hasattr(self.netG, "clear_cache"):
            self.netG.clear_cache()

    def test(self, output=False):
        self.netG.eval()
        with torch.no_grad():
            if self.times is None or "Ours" not in self.net_base:
                raise NotImplementedError("only the 'Ours' generator is on the hot path")
            if self.net_base == "Ours_44":              # one timestamp per call (VideoSR_base_model.py:182-187)
                self.fake_H, flow, flow_GT = self.netG(self.var_L, getattr(self, "real_H", None), self.times[:1], self.scale,
                                                       use_GT=False, iter=4)
                for l in range(1, len(self.times), 1):
                    tmp, flow, flow_GT = self.netG(self.var_L, None, self.times[l:l + 1], self.scale, use_GT=False, iter=4)
                    self.fake_H = torch.cat((self.fake_H, tmp), 0)
            else:
                self.fake_H, flow, flow_GT = self.netG(self.var_L, getattr(self, "real_H", None), self.times[:3], self.scale,
                                                       use_GT=False, iter=4)
                if len(self.times) != 3:
                    for l in range(3, len(self.times), 3):
                        tmp, flow, flow_GT = self.netG(self.var_L, None, self.times[l:l + 3], self.scale, use_GT=False, iter=4)
                        self.fake_H = torch.cat((self.fake_H, tmp), 0)
            self.flow = flow
            self.flow_GT = flow_GT
        self.netG.train()
        if output:
            return self.fake_H

    def get_current_log(self):
        return self.log_dict

    def get_current_visuals(self, need_GT=True):
        out = OrderedDict()
        out["LQ"] = self.var_L.detach()[0].float().cpu()
        out["restore"] = self.fake_H.detach()[0].float().cpu()
        if need_GT:
            out["GT"] = self.real_H.detach()[0].float().cpu()
        return out

    def load(self):
        load_path_G = self.opt["path"]["pretrain_model_G"]
        if load_path_G is not None:
            logger.info("Loading model for G [{:s}] ...".format(load_path_G))
            self.load_network(load_path_G, self.netG, self.opt["path"]["strict_load"])

    def save(self, iter_label):
        self.save_network(self.netG, "G", iter_label)
