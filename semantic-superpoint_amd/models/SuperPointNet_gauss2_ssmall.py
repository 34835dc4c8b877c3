"""Drop-in for the reference's models/SuperPointNet_gauss2_ssmall.py (class SuperPointNet_gauss2_ssmall, :14-99)."""
from ._base import SspNetBase


class SuperPointNet_gauss2_ssmall(SspNetBase):
    """Semantic-SuperPoint: adds "sem": [N,n_classes,H,W] (bilinear upsample of the segmentation head)."""
    ARCH = "SuperPointNet_gauss2_ssmall"

    def __init__(self, n_classes=133, subpixel_channel=1):
        super().__init__()
        self._build(n_classes=n_classes)

    def forward(self, x, subpixel=False):
        output = self._run(x, want_sem=True)
        return {"semi": output["semi"], "desc": output["desc"], "sem": output["sem"]}

    def removeSem(self):
        """Reference :101-104 deletes the seg-head modules; the engine keeps computing that head, so this only
        drops the Python-side parameter containers (export-time helper, off the hot path)."""
        raise NotImplementedError("removeSem() is an export-time helper outside the accelerated path")
