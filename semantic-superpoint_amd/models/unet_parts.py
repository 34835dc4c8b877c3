"""Parameter containers with the module tree (and therefore the state_dict keys) of the reference's
models/unet_parts.py:10-48.  They are never *called*: the arithmetic runs in libssp_hip.so; torch's own
Conv2d / BatchNorm2d constructors are used so that a fresh model gets PyTorch's default initialisation,
exactly like the reference."""
import torch.nn as nn


class double_conv(nn.Module):
    """(conv3x3 => BN => ReLU) * 2; children 0,1,3,4 hold parameters (2 and 5 are the ReLUs)."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_ch, out_ch, 3, padding=1), nn.BatchNorm2d(out_ch), nn.ReLU(inplace=True),
                                  nn.Conv2d(out_ch, out_ch, 3, padding=1), nn.BatchNorm2d(out_ch), nn.ReLU(inplace=True))


class inconv(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = double_conv(in_ch, out_ch)


class down(nn.Module):
    """MaxPool2d(2) then double_conv (the pool is child 0, so the convs live under mpconv.1)."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.mpconv = nn.Sequential(nn.MaxPool2d(2), double_conv(in_ch, out_ch))
