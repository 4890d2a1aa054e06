"""PWC-Net cost volume operator (`/root/reference/OpticalFlow/correlation.py:294-348,415-429`)."""
import torch.nn as nn

from .. import ops


def FunctionCorrelation(tensorFirst, tensorSecond):
    assert tensorFirst.is_contiguous() and tensorSecond.is_contiguous()
    return ops.corr81(tensorFirst, tensorSecond)


class ModuleCorrelation(nn.Module):
    def forward(self, tensorFirst, tensorSecond):
        return FunctionCorrelation(tensorFirst, tensorSecond)
