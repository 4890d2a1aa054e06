"""`BaseModel` (`/root/reference/models/base_model.py`): device choice, checkpoint load/save."""
import os
from collections import OrderedDict

import torch
import torch.nn as nn


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        self.device = torch.device("cuda" if opt["gpu_ids"] is not None else "cpu")
        self.is_train = opt["is_train"]
        self.schedulers = []
        self.optimizers = []

    def get_current_learning_rate(self):
        return [g["lr"] for g in self.optimizers[0].param_groups]

    def get_network_description(self, network):
        if isinstance(network, nn.DataParallel):
            network = network.module
        return str(network), sum(p.numel() for p in network.parameters())

    def save_network(self, network, network_label, iter_label):
        save_path = os.path.join(self.opt["path"]["models"], "{}_{}.pth".format(iter_label, network_label))
        if isinstance(network, nn.DataParallel):
            network = network.module
        torch.save({k: v.cpu() for k, v in network.state_dict().items()}, save_path)

    def load_network(self, load_path, network, strict=True):
        """base_model.py:89-101: accepts {'params': ...}, strips 'module.' prefixes."""
        if isinstance(network, nn.DataParallel):
            network = network.module
        load_net = torch.load(load_path, map_location="cpu")
        if "params" in load_net.keys():
            load_net = load_net["params"]
        clean = OrderedDict()
        for k, v in load_net.items():
            clean[k[7:] if k.startswith("module.") else k] = v
        network.load_state_dict(clean, strict=strict)
