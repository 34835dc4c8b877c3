"""Drop-in replacements for the reference's `models` package (only the two networks on the hot path).
`modelLoader(model=name, **params)` of the reference (utils/loader.py:167-177) resolves
`models.<name>.<name>`; see INTEGRATION.md for the two-line change that points it here."""
from .SuperPointNet_gauss2 import SuperPointNet_gauss2  # noqa: F401
from .SuperPointNet_gauss2_ssmall import SuperPointNet_gauss2_ssmall  # noqa: F401
