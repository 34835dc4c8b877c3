"""Drop-in for the reference's models/SuperPointNet_gauss2.py (class SuperPointNet_gauss2, :12-69)."""
from ._base import SspNetBase


class SuperPointNet_gauss2(SspNetBase):
    """SuperPoint network: forward(x[N,1,H,W]) -> {"semi": [N,65,H/8,W/8], "desc": [N,256,H/8,W/8]}."""
    ARCH = "SuperPointNet_gauss2"

    def __init__(self, subpixel_channel=1):
        super().__init__()
        self._build(n_classes=None)

    def forward(self, x):
        output = self._run(x, want_sem=False)
        return {"semi": output["semi"], "desc": output["desc"]}
