"""`models.networks.define_G` (`/root/reference/models/networks.py:17-61`).  The generator the reference's test.yml
selects (`which_model_G: Ours`, test.yml:41) and its 4-frame forms `Ours_4` / `Ours_44` (networks.py:40-43, SURVEY.md
8(f)4) are on the hot path; the competitor and other ablation models are out of scope."""
from .modules import Ours, Ours_4, Ours_44


def define_G(opt):
    opt_net = opt["network_G"]
    which_model = opt_net["which_model_G"]
    if which_model in ("Ours", "Ours_4", "Ours_44") and opt_net.get("mma"):
        from .. import ops                              # "bf16x3" (default) | "fp32" | "bf16x2" | "bf16", see ops.set_mma
        ops.set_mma(opt_net["mma"])
    if which_model == "Ours":
        if "setting" in opt_net and opt_net["setting"] is not None:
            return Ours.LunaTokis(setting=opt_net["setting"])
        return Ours.LunaTokis()
    if which_model == "Ours_4":
        return Ours_4.LunaTokis()
    if which_model == "Ours_44":
        return Ours_44.LunaTokis()
    raise NotImplementedError("Generator model [{:s}] not recognized".format(which_model))
