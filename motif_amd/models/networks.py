"""`models.networks.define_G` (`/root/reference/models/networks.py:17-61`).  Only the generator the
reference's test.yml selects (`which_model_G: Ours`, test.yml:41) is on the hot path; the competitor
and ablation models are out of scope."""
from .modules import Ours


def define_G(opt):
    opt_net = opt["network_G"]
    which_model = opt_net["which_model_G"]
    if which_model == "Ours":
        if opt_net.get("mma"):                          # "bf16x3" (default) | "fp32" | "bf16x2" | "bf16", see ops.set_mma
            from .. import ops
            ops.set_mma(opt_net["mma"])
        if "setting" in opt_net and opt_net["setting"] is not None:
            return Ours.LunaTokis(setting=opt_net["setting"])
        return Ours.LunaTokis()
    raise NotImplementedError("Generator model [{:s}] not recognized".format(which_model))
